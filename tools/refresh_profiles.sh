#!/bin/bash
# Regenerates the judged evidence under gpurun_out/ on the GPU box (copy into profiles/ afterwards):
#   bench JSON (with cpu_baseline), per-entry kernel table, rocprofv3 kernel stats, PMC FETCH/WRITE passes.
# usage (from the repo root on the GPU box): bash tools/refresh_profiles.sh <tag>
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "${SKIP_BENCH:-0}" != "1" ]; then
python $R/bench.py --dump-kernels $OUT/kernel_table_c2.txt > $OUT/bench_c2_1gpu.json 2> $OUT/bench_c2.err
python $R/bench.py --config c4 --steps 20 --warmup 5 --cpu-steps 0 --dump-kernels $OUT/kernel_table_c4.txt > $OUT/bench_c4_1gpu.json 2> $OUT/bench_c4.err
fi
rm -rf /tmp/ks; rocprofv3 --kernel-trace --stats -d /tmp/ks -o s -- python $R/bench.py --steps 20 --warmup 5 --cpu-steps 0 --profile-steps 1 --headline-only > $OUT/bench_under_rocprof.json 2>/dev/null
python $R/tools/rocprof_summary.py $(find /tmp/ks -name "*.db" | head -1) 26 > $OUT/rocprofv3_kernel_stats_c2.txt
for cnt in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$cnt
  rocprofv3 --kernel-trace --pmc $cnt -d /tmp/pmc_$cnt -o p -- python $R/bench.py --no-graph --steps 3 --warmup 1 --cpu-steps 0 --profile-steps 1 --headline-only > /dev/null 2>&1
done
python $R/tools/rocprof_pmc_summary.py $(find /tmp/pmc_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/pmc_WRITE_SIZE -name "*.db" | head -1) > $OUT/rocprofv3_pmc_hbm_traffic_c2.txt
echo done
