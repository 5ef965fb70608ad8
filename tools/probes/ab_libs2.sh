# bash tools/probes/ab_libs2.sh <libA.so> <libB.so> <configs...>: interleaved same-box A/B of two builds of the library
A=$1; B=$2; shift 2
for cfg in "$@"; do for rep in 1 2; do for lib in $A $B; do
  r=$(PQ3D_LIB_PATH=$lib timeout 400 python bench.py --config $cfg --headline-only --cpu-steps 0 --steps 30 --warmup 10 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.readlines()[-1])['ms_per_step'],4))")
  echo "RESULT $cfg $(basename $lib) $r"
done; done; done
