cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in 0 3; do
for cnt in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  rm -rf /tmp/p1
  timeout 200 rocprofv3 --kernel-trace --pmc $cnt -d /tmp/p1 -o p -- python $R/tools/probes/ws_prof.py $mode 32 32768 > /dev/null 2>&1
  DB=$(find /tmp/p1 -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocprof_pmc_generic.py $DB gemm_ | sed "s/^/mode$mode /"
done; done
