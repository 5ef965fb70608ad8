# bash tools/probes/runtime_knob_sweep.sh [config]: ms/step of the replayed step under HIP-runtime (ROCclr) environment knobs
# that govern how a captured graph is submitted (AQL packet capture, batch size, fence scope, queue count).  One bench.py
# process per setting; "base" lines interleaved to expose box drift.
CFG=${1:-c2}
run() {  # name=value ...
  r=$(env "$@" timeout 300 python bench.py --config $CFG --headline-only --cpu-steps 0 --steps 50 --warmup 10 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.readlines()[-1])['ms_per_step'],4))" 2>/dev/null)
  echo "RESULT $CFG $* -> ${r:-FAILED}"
}
run PQ3D_BASE=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=8
run DEBUG_HIP_GRAPH_BATCH_SIZE=64
run DEBUG_HIP_GRAPH_BATCH_SIZE=1024
run PQ3D_BASE=2
run AMD_OPT_FLUSH=0
run AMD_OPT_FLUSH=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run ROC_SYSTEM_SCOPE_SIGNAL=0
run GPU_MAX_HW_QUEUES=1
run GPU_MAX_HW_QUEUES=2
run GPU_MAX_HW_QUEUES=8
run PQ3D_BASE=3
run DEBUG_HIP_DYNAMIC_QUEUES=0
run DEBUG_HIP_DYNAMIC_QUEUES=1
run ROC_USE_FGS_KERNARG=0
run ROC_SKIP_KERNEL_ARG_COPY=1
run DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0
run DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1
run DEBUG_HIP_KERNARG_COPY_OPT=0
run AMD_DIRECT_DISPATCH=0
run GPU_FLUSH_ON_EXECUTION=1
run PQ3D_BASE=4
