# One-rank RCCL runs of the data-parallel step flow (PQ3D_BENCH_FORCE_DIST=1) in every step mode, next to the plain single-GPU
# step: what the collectives + their launch / join machinery cost with ZERO wire time.  usage: bash tools/probes/rccl_one_rank_modes.sh "<configs>" [buckets]
export MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0
CFGS=${1:-c2}
[ -n "${2:-}" ] && export PQ3D_BENCH_BUCKETS=$2
for cfg in $CFGS; do
  unset PQ3D_BENCH_FORCE_DIST PQ3D_BENCH_STEP_MODE
  r=$(timeout 400 python bench.py --config $cfg --headline-only --cpu-steps 0 --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.readlines()[-1])['ms_per_step'],4))")
  echo "RESULT $cfg single-GPU (no process group) $r"
  export PQ3D_BENCH_FORCE_DIST=1
  for mode in graph_then_allreduce two_graph one_graph; do
    PQ3D_BENCH_STEP_MODE=$mode timeout 400 python -X faulthandler bench.py --config $cfg --gpus 1 --steps 20 --warmup 5 --headline-only --cpu-steps 0 > gpurun_out/rccl_${cfg}_$mode.out 2> gpurun_out/rccl_${cfg}_$mode.err
    rc=$?
    python - <<PY
import json
try:
    l=[x for x in open("gpurun_out/rccl_${cfg}_$mode.out") if x.startswith("{")]
    r=json.loads(l[0]); print("RESULT $cfg $mode", round(r["ms_per_step"],4), "|", r["config"]["step_mode"], "| rccl_ranks", r.get("rccl_ranks"), "| identical", r.get("grads_identical_across_ranks"), "| fp", [round(x[1],3) for x in r.get("grad_fingerprint_per_bucket", [])])
except Exception as e: print("RESULT $cfg $mode no json (rc=$rc)", e)
PY
    grep -v "amdgpu.ids\|hostname of the client\|^$" gpurun_out/rccl_${cfg}_$mode.err | tail -5 | cut -c1-250
  done
done
