# same-box A/B of kernel-library variants: bash tools/probes/ab_libs.sh "<configs>" "<variants ('' = default)>" [steps]
for cfg in $1; do for v in $2; do
  if [ "$v" != default ]; then export PQ3D_LIB_PATH=pq3d_amd/libpq3d_hip_$v.so; else unset PQ3D_LIB_PATH; fi
  r=$(timeout 400 python bench.py --config $cfg --headline-only --cpu-steps 0 --steps ${3:-20} --warmup 5 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.readlines()[-1])['ms_per_step'],4))")
  echo "RESULT $cfg $v $r"
done; done
