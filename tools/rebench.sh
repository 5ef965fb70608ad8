#!/bin/bash
# Re-runs only the bench legs of tools/refresh_profiles.sh (bench JSON + kernel table) against the evidence already under
# profiles/ (same build): bash tools/rebench.sh <tag> [configs...]
set -u
TAG=${1:-r04}; shift
CONFIGS=${@:-c2 c4 c5 s1 s2}
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
for CFG in $CONFIGS; do
  STEPS=50; [ $CFG != c2 ] && STEPS=20
  CPUS=32; case $CFG in s1|s2) CPUS=3;; esac
  timeout 1500 python $R/bench.py --config $CFG --steps $STEPS --warmup 10 --cpu-steps $CPUS --dump-kernels $OUT/kernel_table_${TAG}_fused_$CFG.txt > $OUT/bench_${TAG}_${CFG}_1gpu.json 2> $OUT/bench_$CFG.err
done
timeout 600 python $R/bench.py --config c2 --compute fp32 --headline-only --cpu-steps 0 --steps 30 > $OUT/bench_${TAG}_c2_fp32.json 2>/dev/null
timeout 900 python $R/bench.py --config c5p --cpu-steps 0 --steps 20 > $OUT/bench_${TAG}_c5p_1gpu.json 2>/dev/null
echo done
