#!/bin/bash
# Regenerates the judged evidence under gpurun_out/<tag>/ on the GPU box (copy what should be judged into profiles/):
#   per config: bench JSON (roofline + cpu_baseline), per-entry kernel table, rocprofv3 kernel stats of the replayed
#   step, the step's dispatch sequence, and HBM traffic per kernel from separate --pmc passes (FETCH_SIZE, WRITE_SIZE).
# usage (from the repo root on the GPU box): bash tools/refresh_profiles.sh <tag> [configs...]
# Order matters: the PMC pass of a config runs BEFORE its bench so that bench.py finds profiles/pmc_traffic_<tag>_<cfg>.json
# (roofline.traffic of the same build).
set -u
TAG=${1:-r02}; shift
CONFIGS=${@:-c2 c4 c5}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
# COMPUTE=bf16x3 (or fp32): profile that compute mode; its files carry the mode as a suffix (bench.py committed_profiles)
XC=""; SFX=""; if [ -n "${COMPUTE:-}" ]; then XC="--compute $COMPUTE"; SFX="_$COMPUTE"; fi
cd /tmp && export TMPDIR=/tmp
for CFG in $CONFIGS; do
 if [ -z "${SKIP_PMC:-}" ]; then     # SKIP_PMC=1: only the kernel stats + bench legs (the committed counter files stay)
  for cnt in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$cnt
    timeout 600 rocprofv3 --kernel-trace --pmc $cnt -d /tmp/pmc_$cnt -o p -- python $R/bench.py --config $CFG $XC --no-live-trace --no-graph --steps 3 --warmup 1 --cpu-steps 0 --profile-steps 1 --headline-only --pmc-calibration > /dev/null 2>&1
  done
  F=$(find /tmp/pmc_FETCH_SIZE -name "*.db" | head -1); W=$(find /tmp/pmc_WRITE_SIZE -name "*.db" | head -1)
  python $R/tools/rocprof_pmc_summary.py $F $W > $OUT/rocprofv3_pmc_hbm_traffic_${TAG}_$CFG$SFX.txt
  python $R/tools/pmc_traffic_json.py $F $W > $OUT/pmc_traffic_${TAG}_$CFG$SFX.json
  cp $OUT/pmc_traffic_${TAG}_$CFG$SFX.json $R/profiles/pmc_traffic_${TAG}_$CFG$SFX.json
  # MFMA utilisation (north_star: "MFMA utilisation reported"): its own counter pass
  rm -rf /tmp/pmc_mfma
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d /tmp/pmc_mfma -o p -- python $R/bench.py --config $CFG $XC --no-live-trace --no-graph --steps 3 --warmup 1 --cpu-steps 0 --profile-steps 1 --headline-only > /dev/null 2>&1
  MF=$(find /tmp/pmc_mfma -name "*.db" | head -1)
  python $R/tools/pmc_mfma_json.py $MF > $OUT/pmc_mfma_${TAG}_$CFG$SFX.json && cp $OUT/pmc_mfma_${TAG}_$CFG$SFX.json $R/profiles/
 fi
  rm -rf /tmp/ks
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks -o s -- python $R/bench.py --config $CFG $XC --no-live-trace --steps 20 --warmup 5 --cpu-steps 0 --profile-steps 1 --headline-only --min-time 0 > $OUT/bench_under_rocprof_$CFG$SFX.json 2>/dev/null
  DB=$(find /tmp/ks -name "*.db" | head -1)
  # steps in this trace: 3 eager warm-up steps + 5 warm-up replays + 20 timed replays + 1 eager profiled step
  python $R/tools/rocprof_summary.py $DB 29 > $OUT/rocprofv3_kernel_stats_${TAG}_fused_graph_$CFG$SFX.txt
  python $R/tools/rocprof_step_sequence.py $DB > $OUT/rocprofv3_step_sequence_${TAG}_$CFG$SFX.txt 2>&1
  # the bench reads the in-graph kernel averages of THIS build from profiles/ (roofline.in_graph)
  cp $OUT/rocprofv3_kernel_stats_${TAG}_fused_graph_$CFG$SFX.txt $R/profiles/
  STEPS=50; [ $CFG != c2 ] && STEPS=20
  CPUS=32; case $CFG in s1|s2) CPUS=3;; esac     # the shipped-size oracle step takes ~10 s on the host: a 3-step sample
  timeout 1500 python $R/bench.py --config $CFG $XC --steps $STEPS --warmup 10 --cpu-steps $CPUS --dump-kernels $OUT/kernel_table_${TAG}_fused_$CFG$SFX.txt > $OUT/bench_${TAG}_${CFG}${SFX}_1gpu.json 2> $OUT/bench_$CFG.err
  cp $OUT/bench_${TAG}_${CFG}${SFX}_1gpu.json $OUT/kernel_table_${TAG}_fused_$CFG$SFX.txt $OUT/rocprofv3_step_sequence_${TAG}_$CFG$SFX.txt $OUT/rocprofv3_pmc_hbm_traffic_${TAG}_$CFG$SFX.txt $R/profiles/ 2>/dev/null
done
echo done
