#!/bin/bash
# A/B of the whole-K small-M GEMM kernels on one box: bench c2 (and c4) with PQ3D_WK=0 / 1, + the replayed step's dispatch sequence
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/ab_wk
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for WK in 0 1; do
  for CFG in ${CFGS:-c2}; do
    PQ3D_WK=$WK timeout 600 python $R/bench.py --config $CFG --steps 50 --warmup 10 --cpu-steps 0 --profile-steps 1 --headline-only > $OUT/bench_${CFG}_wk${WK}_$rep.json 2> $OUT/bench_${CFG}_wk${WK}_$rep.err
    python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_${CFG}_wk${WK}_$rep.json").read().strip().splitlines()[-1])
    print("$CFG WK=$WK rep $rep ms_per_step", d["ms_per_step"])
except Exception as e:
    print("$CFG WK=$WK failed", e)
PY
  done
done
done
for WK in 0 1; do
  rm -rf /tmp/ks$WK
  PQ3D_WK=$WK timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks$WK -o s -- python $R/bench.py --config c2 --steps 20 --warmup 5 --cpu-steps 0 --profile-steps 0 --headline-only > /dev/null 2>&1
  DB=$(find /tmp/ks$WK -name "*.db" | head -1)
  python $R/tools/rocprof_step_sequence.py $DB > $OUT/step_sequence_c2_wk$WK.txt 2>&1
  tail -1 $OUT/step_sequence_c2_wk$WK.txt
done
