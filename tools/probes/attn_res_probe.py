"""Resident attention backward: timing at the c2 / c4 cross-attention shapes, on/off, key splits, ablations."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pq3d_amd import _lib as L
from pq3d_amd import fused as F
from pq3d_amd import ops
dev = 'cuda'
H, d = 8, 256
lib = L.lib()


def timeit(fn, it=40):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


def case(name, B, Lq, Lk, mask3=False):
    q = torch.randn(B, Lq, d, device=dev).bfloat16(); k = torch.randn(B, Lk, d, device=dev).bfloat16()
    v = torch.randn(B, Lk, d, device=dev).bfloat16()
    o = torch.empty_like(q); lse = torch.empty(B, H, Lq, device=dev)
    do = torch.randn_like(o); dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v)
    delta = torch.empty_like(lse)
    kw = {}
    vl = torch.randint(Lk // 2, Lk + 1, (B,)); vl[0] = Lk
    if mask3:
        m = torch.rand(B // 3, Lq, Lk, device=dev) < 0.6
        kw = dict(mask=m, row_open=m.all(-1), mask_bmod=B // 3)
    else:
        kw = dict(kpm=(torch.arange(Lk)[None] >= vl[:, None]).to(dev))
    F._attn(q, k, v, o, lse, H, L.BF16, True, **kw)
    bwd = lambda: F._attn(q, k, v, o, lse, H, L.BF16, True, bwd=(do, dq, dk, dv, delta, None), **kw)
    for ks in (0, 1, 2, 4, 8):
        ops._ATTN_KSPLIT = ks
        row = []
        for mode in (0, 1):
            lib.pq3d_attn_resident(mode)
            row.append(timeit(bwd))
        lib.pq3d_attn_resident(1)
        print(f"{name:26s} ks={ks}: two-kernel {row[0]:6.1f} | resident {row[1]:6.1f} us")
    ops._ATTN_KSPLIT = 0


case("c2 cross B24 Lq100 Lk1024", 24, 100, 1024)
case("c4 cross B12 Lq200 Lk4096", 12, 200, 4096, mask3=True)
case("c5 cross B48 Lq100 Lk2048", 48, 100, 2048)
