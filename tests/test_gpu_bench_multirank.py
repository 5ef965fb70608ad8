"""GPU: the N > 1 flow of bench.py end to end on ONE GPU -- two ranks launched exactly as the driver launches them
(python -m torch.distributed.run ... bench.py --gpus 2), sharing the device through the gloo test hook
(PQ3D_BENCH_BACKEND, see bench.py; RCCL refuses two ranks on one device).  Checks the contract fields and that the
all-reduced gradient is identical on both ranks."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("launcher", ["torchrun", "plain"])
@pytest.mark.parametrize("config", ["c1", "c2"])
def test_bench_two_ranks_one_gpu(config, launcher):
    """launcher 'torchrun': the driver's N > 1 form; 'plain': `python bench.py --gpus 2` with no launcher -- bench.py starts
    the ranks itself (a 1-rank process must never print a line for a 2-GPU request)."""
    env = dict(os.environ, PQ3D_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--config", config, "--headline-only"]
    if launcher == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(_free_port())] + tail
    else:
        cmd = [sys.executable] + tail
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]            # rank 0 prints ONE JSON line
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 4 and r["warmup"] == 2 and r["scaling"] == "weak"
    assert r["config"]["parallelism"] == "dp2" and r["config"]["global_batch"] % 2 == 0
    assert r["value"] > 0 and abs(r["value"] - r["config"]["global_batch"] / (r["ms_per_step"] * 1e-3)) < 1e-6 * r["value"]
    assert r["grads_identical_across_ranks"] is True
    assert r["collective_backend"] == "gloo" and r["rccl_ranks"] == 0     # the test hook, not RCCL
    assert "cpu_baseline" not in r                       # rank-0 / N=1 only
    if config == "c2":
        # the fused decoder path: collectives cannot be captured with gloo, so the RCCL-independent overlap mode must have
        # run -- graphs split at decoder-gradients-final, the coalesced decoder bucket all-reduced in between
        mode = r["config"]["step_mode"]
        assert mode.startswith("2 graphs split at decoder") and "buckets=coalesced" in mode, (mode, p.stderr[-1500:])


@pytest.mark.parametrize("mode", ["", "one_graph", "graph_then_allreduce", "eager"])
def test_bench_one_rank_rccl(mode):
    """RCCL on the box we have: a communicator of ONE rank (PQ3D_BENCH_FORCE_DIST=1) runs the whole data-parallel step
    flow over backend 'nccl' -- the default (two graphs, the decoder buckets' ReduceOp.AVG all-reduces launched eagerly on
    the side stream between them), the collectives captured INSIDE one HIP graph (on request; on the capturing stream: a
    forked side stream crashes hipStreamEndCapture on this stack, tools/probes/rccl_capture_probe.py), and both
    fallbacks.  A mean over one rank is the identity, so the gradients must come out finite and non-zero, and the step
    mode must be the requested one."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0",
               PQ3D_BENCH_FORCE_DIST="1", PQ3D_BENCH_STEP_MODE=mode)
    for k in ("PQ3D_BENCH_BACKEND", "RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--headline-only"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert r["n_gpus"] == 1 and r["rccl_ranks"] == 1 and r["collective_backend"] == "nccl"
    assert r["grads_identical_across_ranks"] is True and r["value"] > 0
    sm = r["config"]["step_mode"]
    want = {"": "2 graphs split", "one_graph": "graph(step+allreduce", "graph_then_allreduce": "graph(fwd+bwd) then allreduce",
            "eager": "eager"}[mode]
    assert sm.startswith(want), (sm, p.stderr[-2000:])


@pytest.mark.parametrize("mode", ["", "one_graph"])
def test_bench_one_rank_rccl_bf16_wire(mode):
    """the bf16 wire format (fp32 accumulation: all-to-all + local fp32 sum + all-gather, pq3d_amd/parallel.py)
    through RCCL with one rank -- the collectives, the casts and their capture inside the step's graphs all execute; with one rank
    the result is the bf16 rounding of the rank's own gradients (finite, non-zero, identical fingerprints)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0",
               PQ3D_BENCH_FORCE_DIST="1", PQ3D_BENCH_STEP_MODE=mode, PQ3D_BENCH_WIRE="bf16")
    for k in ("PQ3D_BENCH_BACKEND", "RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--headline-only"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert r["rccl_ranks"] == 1 and r["config"]["gradient_wire_dtype"] == "bf16"
    assert r["grads_identical_across_ranks"] is True and r["value"] > 0 and r["chain_error"] is False


@pytest.mark.parametrize("buckets", ["coalesced", "per_layer"])
def test_bench_one_rank_rccl_caption_config_splits_at_the_heads(buckets):
    """Config 5 (caption head) over a one-rank RCCL communicator: the heads' bucket is launched when the decoder backward
    starts (a 3-piece graph split: heads / decoder / rest) in both bucket layouts; gradients finite, non-zero in every bucket."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0",
               PQ3D_BENCH_FORCE_DIST="1", PQ3D_BENCH_BUCKETS=buckets)
    for k in ("PQ3D_BENCH_BACKEND", "RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "PQ3D_BENCH_STEP_MODE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--headline-only",
           "--config", "c5", "--cpu-steps", "0"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert r["rccl_ranks"] == 1 and r["grads_identical_across_ranks"] is True
    sm = r["config"]["step_mode"]
    assert sm.startswith("3 graphs split at heads / decoder") and f"buckets={buckets}" in sm, (sm, p.stderr[-2000:])
    assert all(fp[1] > 0 for fp in r["grad_fingerprint_per_bucket"])


def test_bench_two_ranks_rccl():
    """The real thing where the box has it: two ranks on two GPUs over RCCL (backend 'nccl'), so that the driver's
    multi-GPU run is not RCCL's first execution of this path.  Exercises graph capture with the collectives inside (or
    its fallback), ReduceOp.AVG and the early bucket-0 launch.  Skipped on single-GPU boxes."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL refuses two ranks on one device)")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("PQ3D_BENCH_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--headline-only"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert r["n_gpus"] == 2 and r["grads_identical_across_ranks"] is True and r["value"] > 0 and r["rccl_ranks"] == 2


def test_world_size_must_match_the_request():
    """WORLD_SIZE=1 with --gpus 2 (a launcher that started one rank) must fail instead of printing a 1-GPU line."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
