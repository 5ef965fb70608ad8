# kernel stats + FETCH_SIZE traffic of one config with the current build (quick look; not judged evidence)
R=$PWD; CFG=${1:-s2}; OUT=$R/gpurun_out/xcd; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks -o s -- python $R/bench.py --config $CFG --steps 20 --warmup 5 --cpu-steps 0 --profile-steps 1 --headline-only > /dev/null 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/ks -name "*.db" | head -1) 29 > $OUT/stats_$CFG.txt
for cnt in FETCH_SIZE WRITE_SIZE; do rm -rf /tmp/pmc_$cnt
 timeout 600 rocprofv3 --kernel-trace --pmc $cnt -d /tmp/pmc_$cnt -o p -- python $R/bench.py --config $CFG --no-graph --steps 3 --warmup 1 --cpu-steps 0 --profile-steps 1 --headline-only --pmc-calibration > /dev/null 2>&1
done
python $R/tools/pmc_traffic_json.py $(find /tmp/pmc_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/pmc_WRITE_SIZE -name "*.db" | head -1) > $OUT/traffic_$CFG.json
