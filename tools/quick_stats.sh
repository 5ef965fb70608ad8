#!/bin/bash
# rocprofv3 kernel stats + step sequence of one config's replayed step into gpurun_out/quick/ (no PMC, no bench JSON):
#   bash tools/quick_stats.sh <config> [tag]
CFG=${1:-c2}; TAG=${2:-quick}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_$CFG
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks_$CFG -o s -- python $R/bench.py --config $CFG ${BENCH_EXTRA} --no-live-trace --steps 20 --warmup 5 --cpu-steps 0 --profile-steps 1 --headline-only --min-time 0 > $OUT/bench_under_rocprof_$CFG.json 2>/dev/null
DB=$(find /tmp/ks_$CFG -name "*.db" | head -1)
python $R/tools/rocprof_summary.py $DB 29 > $OUT/kernel_stats_$CFG.txt
python $R/tools/rocprof_step_sequence.py $DB > $OUT/step_sequence_$CFG.txt 2>&1
head -3 $OUT/kernel_stats_$CFG.txt | cut -c1-200
