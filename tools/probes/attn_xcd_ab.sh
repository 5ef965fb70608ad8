#!/bin/bash
# A/B of the XCD-aware attention workgroup order (attn_common.h attn_wg_xyz): timing + FETCH_SIZE per kernel, default library
# vs the variant built with -DPQ3D_ATTN_XCD=0 (tools/build_variant.py noxcd -DPQ3D_ATTN_XCD=0).  Repo root, GPU box.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/attn_xcd; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in default noxcd; do
  if [ $v != default ]; then export PQ3D_LIB_PATH=$R/pq3d_amd/libpq3d_hip_$v.so; else unset PQ3D_LIB_PATH; fi
  python $R/tools/probes/attn_xcd_probe.py 30 2>&1 | grep RESULT | sed "s/^/$v /" | tee -a $OUT/timing.txt
  rm -rf /tmp/pmc_$v
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_$v -o p -- python $R/tools/probes/attn_xcd_probe.py 3 > /dev/null 2>&1
  F=$(find /tmp/pmc_$v -name "*.db" | head -1)
  python $R/tools/rocprof_pmc_generic.py $F > $OUT/fetch_$v.txt
  grep -i "attn\|copy_many" $OUT/fetch_$v.txt | sed "s/^/$v /"
done
