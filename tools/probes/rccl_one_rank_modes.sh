export PQ3D_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0
for mode in eager graph_then_allreduce two_graph one_graph; do
  echo "=== mode $mode"
  PQ3D_BENCH_STEP_MODE=$mode timeout 300 python -X faulthandler bench.py --gpus 1 --steps 6 --warmup 2 --headline-only > gpurun_out/rccl_$mode.out 2> gpurun_out/rccl_$mode.err
  echo "rc=$?"
  python - <<PY
import json
try:
    l=[x for x in open("gpurun_out/rccl_$mode.out") if x.startswith("{")]
    r=json.loads(l[0]); print(r["ms_per_step"], r["config"]["step_mode"], r.get("rccl_ranks"), r.get("grads_identical_across_ranks"))
except Exception as e: print("no json", e)
PY
  grep -v "amdgpu.ids\|hostname of the client" gpurun_out/rccl_$mode.err | tail -40 | cut -c1-250
done
