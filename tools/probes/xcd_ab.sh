set -x

timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "gemm" 2>&1 | tail -5
for cfg in s2 s1 c2 c4 c5; do
  for v in xoff ""; do
    if [ -n "$v" ]; then export PQ3D_LIB_PATH=pq3d_amd/libpq3d_hip_$v.so; else unset PQ3D_LIB_PATH; fi
    r=$(timeout 400 python bench.py --config $cfg --headline-only --cpu-steps 0 --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.readlines()[-1])['ms_per_step'],4))")
    echo "RESULT $cfg ${v:-default1024} $r"
  done
done
