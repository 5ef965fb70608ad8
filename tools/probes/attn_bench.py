"""Microbenchmark of the attention launches at the c2 / c4 shapes (eager, HIP events over many iterations).
   python tools/probes/attn_bench.py [iters]"""
import sys
sys.path.insert(0, '/root/repo')
import torch
from pq3d_amd import _lib as L
from pq3d_amd import fused as F
dev = 'cuda'
it = int(sys.argv[1]) if len(sys.argv) > 1 else 50
H, d = 8, 256


def case(name, B, Lq, Lk, kpm=False, bias=False):
    q = torch.randn(B, Lq, d, device=dev).bfloat16()
    k = torch.randn(B, Lk, d, device=dev).bfloat16()
    v = torch.randn(B, Lk, d, device=dev).bfloat16()
    o = torch.empty_like(q); lse = torch.empty(B, H, Lq, device=dev)
    do = torch.randn_like(o); dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v)
    delta = torch.empty_like(lse)
    kw = {}
    if kpm:
        m = torch.zeros(B, Lk, dtype=torch.bool, device=dev)
        m[:, Lk - Lk // 8:] = True
        m[0] = False
        kw["kpm"] = m
    if bias:
        kw["bias"] = torch.randn(B, H, Lq, Lk, device=dev)
    fwd = lambda: F._attn(q, k, v, o, lse, H, L.BF16, kpm, **kw)
    bwd = lambda: F._attn(q, k, v, o, lse, H, L.BF16, kpm, bwd=(do, dq, dk, dv, delta, None), **kw)
    res = []
    for fn in (fwd, bwd):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(it):
            fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / it * 1e3)
    print(f"{name:28s} fwd {res[0]:7.1f} us   bwd {res[1]:7.1f} us")


case("c2 cross B24 Lq100 Lk1024", 24, 100, 1024, kpm=True)
case("c2 self  B8  Lq100 Lk100", 8, 100, 100, bias=True)
case("c4 cross B12 Lq200 Lk4096", 12, 200, 4096, kpm=True)
