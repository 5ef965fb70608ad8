"""GPU: the C-ABI gradient exchange (pq3d_comm_init / pq3d_allreduce_grads / pq3d_allreduce_grads_wire, csrc/comm.hip; SURVEY 8b's
export list, 8e) on the box we have: a ONE-rank RCCL communicator.  RCCL refuses two ranks on one device, so N > 1 stays covered by
the arithmetic tests here (the wire kernels against their definition with W simulated pieces), the 2-rank gloo tests and the
>= 2-GPU test at the end (skipped on one-GPU boxes).  Reference behaviour: DDP's all-reduce(mean), trainer/build.py:66-75."""
import ctypes as C
import json
import os
import subprocess
import sys

import pytest
import torch

from pq3d_amd import _lib
from pq3d_amd.parallel import FlatGradAllReducer, NativeComm

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda"


@pytest.fixture(scope="module")
def comm():
    c = NativeComm(0, 1, NativeComm.unique_id())
    yield c
    c.close()


def test_one_rank_communicator_reports_itself(comm):
    r, w, v = C.c_int32(-1), C.c_int32(-1), C.c_int32(0)
    _lib.check(_lib.lib().pq3d_comm_info(comm._h, C.byref(r), C.byref(w), C.byref(v)), "pq3d_comm_info")
    assert (r.value, w.value) == (0, 1) and v.value >= 20000      # an RCCL 2.x runtime was bound


@pytest.mark.parametrize("n", [1, 7, 8, 1000, 9_199_873])
@pytest.mark.parametrize("mean", [True, False])
def test_allreduce_fp32_one_rank_is_the_identity(comm, n, mean):
    g = torch.randn(n, device=DEV)
    ref = g.clone()
    comm.all_reduce(g, mean=mean)
    torch.cuda.synchronize()
    assert torch.equal(g, ref)


def test_allreduce_bf16_buffer(comm):
    g = torch.randn(4096, device=DEV).to(torch.bfloat16)
    ref = g.clone()
    comm.all_reduce(g)
    torch.cuda.synchronize()
    assert torch.equal(g, ref)


@pytest.mark.parametrize("n", [1, 7, 8, 9, 1000, 4097, 9_199_873])
def test_wire_form_one_rank_is_one_bf16_rounding(comm, n):
    """bf16 on the links, fp32 accumulation: with one rank the result is float(bf16(g)) -- and exactly that for every tail length
    (the 16-byte padding travels as zeros and never comes back)."""
    g = torch.randn(n + 5, device=DEV) * 3.0
    guard = g[n:].clone()
    view = g[:n]
    ref = view.to(torch.bfloat16).float()
    comm.all_reduce(view, wire_bf16=True)
    torch.cuda.synchronize()
    assert torch.equal(view, ref)
    assert torch.equal(g[n:], guard)          # nothing written past the bucket


def test_scratch_size_and_argument_errors(comm):
    L = _lib.lib()
    for world, n in [(1, 1), (2, 17), (8, 9_199_873)]:
        per = -(-n // world)
        per += (-per) % 8
        assert L.pq3d_allreduce_wire_scratch_bytes(world, n) == (3 * world + 1) * per * 2
    assert L.pq3d_allreduce_wire_scratch_bytes(0, 8) == -1
    g = torch.zeros(64, device=DEV)
    small = torch.zeros(16, dtype=torch.uint8, device=DEV)
    s = _lib.stream()
    assert L.pq3d_allreduce_grads_wire(comm._h, g.data_ptr(), 64, small.data_ptr(), small.numel(), 1, s) == -1
    assert b"scratch" in L.pq3d_last_error()
    assert L.pq3d_allreduce_grads(comm._h, g.data_ptr(), 64, 7, 1, s) == -1            # no such dtype
    bogus = torch.zeros(64, dtype=torch.uint8)                                          # host memory that is not a handle
    assert L.pq3d_allreduce_grads(bogus.data_ptr(), g.data_ptr(), 64, 0, 1, s) == -1
    assert b"not a communicator" in L.pq3d_last_error()
    h = C.c_void_p()
    assert L.pq3d_comm_init(3, 2, C.c_char_p(NativeComm.unique_id()), C.byref(h)) == -1  # rank outside the world
    torch.cuda.synchronize()


@pytest.mark.parametrize("W", [2, 3, 8])
@pytest.mark.parametrize("mean", [True, False])
def test_wire_reduce_sums_in_fp32_in_rank_order(W, mean):
    """The W > 1 arithmetic of the wire form on one GPU (pq3d_test_wire_reduce = the kernel pq3d_allreduce_grads_wire launches
    between its all-to-all and its all-gather): shard = bf16((((p_0 + p_1) + p_2) ...) / W) in fp32 -- bit for bit, and NOT what a
    bf16 ring sum would give."""
    torch.manual_seed(W)
    per = 8 * 1031
    pieces = (torch.randn(W, per, device=DEV) * 10).to(torch.bfloat16)
    acc = torch.zeros(per, device=DEV)
    for r in range(W):
        acc = acc + pieces[r].float()
    want = (acc / float(W) if mean else acc).to(torch.bfloat16)
    shard = torch.empty(per, dtype=torch.bfloat16, device=DEV)
    _lib.check(_lib.lib().pq3d_test_wire_reduce(pieces.data_ptr(), shard.data_ptr(), per, W, int(mean), _lib.stream()), "wire_reduce")
    torch.cuda.synchronize()
    assert torch.equal(shard, want)
    ring = pieces[0].clone()
    for r in range(1, W):
        ring = ring + pieces[r]                 # bf16 accumulation: what ncclAllReduce on a bf16 buffer does
    exact = pieces.double().sum(0)
    got = shard.double() * (W if mean else 1)
    assert float((got - exact).abs().max()) <= float((ring.double() - exact).abs().max())


def test_flat_reducer_through_the_native_communicator(comm):
    """FlatGradAllReducer(comm=...) with the one-rank probe switch: pack -> C-ABI exchange on the side stream -> views; fp32 wire
    leaves the gradients untouched, bf16 wire leaves their bf16 rounding -- identical to the torch.distributed-free definitions."""
    torch.manual_seed(1)
    ps = [torch.nn.Parameter(torch.randn(s, device=DEV)) for s in [(256, 256), (256,), (1000, 3), (7,)]]
    grads = [torch.randn_like(p) for p in ps]
    for wire, expect in [(None, lambda g: g), (torch.bfloat16, lambda g: g.to(torch.bfloat16).float())]:
        for p, g in zip(ps, grads):
            p.grad = g.clone()
        red = FlatGradAllReducer(ps, groups=[ps[:2], ps[2:]], wire_dtype=wire, comm=comm)
        red.force_collectives = True
        red.pack()
        red.launch(0)            # early launch of bucket 0 on the side stream
        red.finish()
        red.unpack_views()
        torch.cuda.synchronize()
        for p, g in zip(ps, grads):
            assert torch.equal(p.grad, expect(g))


def test_native_exchange_inside_a_hip_graph(comm):
    """The exchange is stream-ordered (no host synchronisation), so a step's graph can hold it: capture pack-less buckets' wire
    exchange on the capturing stream, replay twice with new contents."""
    g = torch.randn(100_003, device=DEV)
    comm.all_reduce(g, wire_bf16=True)        # warm-up outside the capture (scratch allocation, RCCL's lazy setup)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        comm.all_reduce(g, wire_bf16=True)
    for seed in (3, 4):
        torch.manual_seed(seed)
        src = torch.randn(100_003, device=DEV)
        g.copy_(src)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(g, src.to(torch.bfloat16).float())


@pytest.mark.parametrize("wire", ["fp32", "bf16"])
def test_bench_one_rank_native_comm(wire):
    """bench.py's data-parallel step flow with the buckets routed through the C-ABI communicator (PQ3D_BENCH_COMM=native), one rank."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0",
               PQ3D_BENCH_FORCE_DIST="1", PQ3D_BENCH_COMM="native", PQ3D_BENCH_WIRE=wire)
    for k in ("PQ3D_BENCH_BACKEND", "RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "PQ3D_BENCH_STEP_MODE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--headline-only"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert r["rccl_ranks"] == 1 and r["config"]["gradient_exchange"].startswith("pq3d_comm")
    assert r["config"]["gradient_wire_dtype"] == wire
    assert r["grads_identical_across_ranks"] is True and r["value"] > 0 and r["chain_error"] is False


def test_two_ranks_native_comm(tmp_path):
    """Two ranks on two GPUs through the C-ABI communicator: mean of rank-dependent buckets, fp32 and bf16 wire.  Skipped on
    one-GPU boxes (RCCL refuses two ranks on one device)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    code = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from pq3d_amd.parallel import NativeComm
rank = int(os.environ["RANK"]); torch.cuda.set_device(rank)
dist.init_process_group("gloo")
c = NativeComm.from_process_group()
n = 1_000_003
base = torch.arange(n, device="cuda") * 1e-3
g = base + float(rank + 1)
a = g.clone(); c.all_reduce(a)
b = g.clone(); c.all_reduce(b, wire_bf16=True)
torch.cuda.synchronize()
want = ((base + 1.0) + (base + 2.0)) / 2
assert torch.allclose(a, want, rtol=1e-6), "fp32 mean"
p0, p1 = (base + 1.0).to(torch.bfloat16).float(), (base + 2.0).to(torch.bfloat16).float()
assert torch.equal(b, ((p0 + p1) / 2).to(torch.bfloat16).float()), "wire form"
c.close(); dist.destroy_process_group()
print("ok", rank)
""" % ROOT
    script = tmp_path / "two_ranks.py"
    script.write_text(code)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.count("ok") == 2, p.stderr[-3000:]
