#!/bin/bash
# A/B of the "unit" XCD order of gemm_tt128 (K/V weight gradients): step time at c2 / c4 / c5 and FETCH_SIZE of the kernel at c5,
# default library vs -DPQ3D_TT_UNIT_OFF (tools/build_variant.py nottunit -DPQ3D_TT_UNIT_OFF).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/probes/ab_libs.sh "c2 c4 c5" "default nottunit default nottunit" 20 2>&1 | grep RESULT
cd /tmp && export TMPDIR=/tmp
for v in default nottunit; do
  if [ $v != default ]; then export PQ3D_LIB_PATH=$R/pq3d_amd/libpq3d_hip_$v.so; else unset PQ3D_LIB_PATH; fi
  rm -rf /tmp/pmc_$v
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_$v -o p -- python $R/bench.py --config c5 --no-graph --steps 2 --warmup 1 --cpu-steps 0 --profile-steps 1 --headline-only --min-time 0 > /dev/null 2>&1
  F=$(find /tmp/pmc_$v -name "*.db" | head -1)
  python $R/tools/rocprof_pmc_generic.py $F gemm_tt128 | sed "s/^/$v /"
done
