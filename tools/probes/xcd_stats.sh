# per-kernel in-graph durations, XCD-aware order on (default build) vs off (variant xoff), same box
R=$PWD; OUT=$R/gpurun_out/xcd; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for CFG in "$@"; do for v in xoff on; do
  if [ $v = xoff ]; then export PQ3D_LIB_PATH=$R/pq3d_amd/libpq3d_hip_xoff.so; else unset PQ3D_LIB_PATH; fi
  rm -rf /tmp/ks; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks -o s -- python $R/bench.py --config $CFG --steps 20 --warmup 5 --cpu-steps 0 --profile-steps 1 --headline-only > /dev/null 2>&1
  python $R/tools/rocprof_summary.py $(find /tmp/ks -name "*.db" | head -1) 29 > $OUT/stats_${CFG}_$v.txt
done; done
