"""CPU, world_size 2, gloo: the data-parallel gradient exchange used by bench.py / a standalone harness
(pq3d_amd.parallel.FlatGradAllReducer): pack -> all-reduce(mean) -> re-point .grad at the reduced flat views."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pq3d_amd.parallel import FlatGradAllReducer
    torch.manual_seed(0)
    lin = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.LayerNorm(5), torch.nn.Linear(5, 3))
    unused = torch.nn.Parameter(torch.ones(4))           # a parameter that gets no gradient this step
    params = list(lin.parameters()) + [unused]
    red = FlatGradAllReducer(params, bucket_bytes=64)    # tiny buckets: exercises multi-bucket packing
    x = torch.full((4, 7), float(rank + 1))
    lin(x).square().mean().backward()
    local = [p.grad.clone() if p.grad is not None else torch.zeros_like(p) for p in params]
    red.step()
    gathered = [None] * world
    dist.all_gather_object(gathered, [g.numpy() for g in local])
    for i, p in enumerate(params):
        mean = sum(torch.from_numpy(gathered[r][i]) for r in range(world)) / world
        assert torch.allclose(p.grad, mean, atol=1e-6), (rank, i)
        assert p.grad.data_ptr() >= red.flat[0].data_ptr() or len(red.flat) > 1   # a view of a flat bucket
    assert len(red.flat) > 1
    # explicit buckets in readiness order + early launch of the first one (what bench.py does from inside the backward)
    for p in params:
        p.grad = None
    red2 = FlatGradAllReducer(params, groups=[list(lin[2].parameters()), list(lin[0].parameters()) + list(lin[1].parameters()) + [unused]])
    lin(x).square().mean().backward()
    local2 = [p.grad.clone() if p.grad is not None else torch.zeros_like(p) for p in params]
    red2.pack(buckets=[0])
    red2.launch(0)              # bucket 0 in flight ...
    red2.pack(buckets=[1])      # ... while the rest is still being packed
    red2.finish()
    red2.unpack_views()
    dist.all_gather_object(gathered, [g.numpy() for g in local2])
    for i, p in enumerate(params):
        mean = sum(torch.from_numpy(gathered[r][i]) for r in range(world)) / world
        assert torch.allclose(p.grad, mean, atol=1e-6), (rank, i, "groups")
    # bf16 wire format with fp32 accumulation (wire_dtype): every rank ends with bit-identical buffers = bf16(mean in fp32 of the
    # bf16-rounded contributions)
    for p in params:
        p.grad = None
    red3 = FlatGradAllReducer(params, groups=[list(lin[2].parameters()), list(lin[0].parameters()) + list(lin[1].parameters()) + [unused]],
                              wire_dtype=torch.bfloat16)
    (lin(x) * 0.37).square().mean().backward()
    local3 = [p.grad.clone() if p.grad is not None else torch.zeros_like(p) for p in params]
    red3.pack(buckets=[0])
    red3.launch(0)
    red3.pack(buckets=[1])
    red3.finish()
    red3.unpack_views()
    dist.all_gather_object(gathered, [g.numpy() for g in local3])
    for i, p in enumerate(params):
        contrib = [torch.from_numpy(gathered[r][i]).bfloat16().float() for r in range(world)]
        want = (sum(contrib) / world).bfloat16().float()
        assert torch.equal(p.grad, want), (rank, i, "bf16 wire", (p.grad - want).abs().max())
    flat3 = [f.clone() for f in red3.flat]
    other = [None] * world
    dist.all_gather_object(other, [f.numpy() for f in flat3])
    assert all((other[0][k] == other[1][k]).all() for k in range(len(flat3))), "ranks must hold identical reduced buffers"
    out.put((rank, float(sum(f.abs().sum() for f in red.flat))))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_grad_allreduce_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get() for _ in range(2))
    assert abs(res[0] - res[1]) < 1e-6   # identical averaged gradients on both ranks
