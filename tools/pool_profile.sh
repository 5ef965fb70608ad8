#!/bin/bash
# Evidence of the segment-pooling workload (bench.py --config pool) under gpurun_out/<tag>/ and profiles/:
#   rocprofv3 --kernel-trace --stats summary of the bench command, separate --pmc FETCH_SIZE / WRITE_SIZE passes of the
#   headline launch (with the calibration copy), then the bench JSON itself (which reads the PMC file for roofline.traffic).
# usage (repo root, GPU box): bash tools/pool_profile.sh <tag>
set -u
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cnt in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$cnt
  timeout 300 rocprofv3 --kernel-trace --pmc $cnt -d /tmp/pmc_$cnt -o p -- python $R/bench.py --config pool --pool-pmc > /dev/null 2>&1
done
F=$(find /tmp/pmc_FETCH_SIZE -name "*.db" | head -1); W=$(find /tmp/pmc_WRITE_SIZE -name "*.db" | head -1)
python $R/tools/pmc_traffic_json.py $F $W > $OUT/pmc_traffic_${TAG}_pool.json && cp $OUT/pmc_traffic_${TAG}_pool.json $R/profiles/
rm -rf /tmp/ks
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks -o s -- python $R/bench.py --config pool --steps 10 --warmup 3 --cpu-steps 0 > $OUT/bench_under_rocprof_pool.json 2>/dev/null
DB=$(find /tmp/ks -name "*.db" | head -1)
python $R/tools/rocprof_summary.py $DB 1 > $OUT/rocprofv3_kernel_stats_${TAG}_pool.txt
cp $OUT/rocprofv3_kernel_stats_${TAG}_pool.txt $R/profiles/
timeout 600 python $R/bench.py --config pool --steps 20 --warmup 5 > $OUT/bench_${TAG}_pool_1gpu.json 2> $OUT/bench_pool.err
cp $OUT/bench_${TAG}_pool_1gpu.json $R/profiles/
echo done
