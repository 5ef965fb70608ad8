"""Target for rocprofv3 runs: the c2 cross-attention (B24 H8 Lq100 Lk1024 dh32, kpm, zero_attn) and spatial self-attention
(B8 H8 Lq100 Lk100, bias) forward + backward launches alone.  argv[1] = iterations (default 5)."""
import sys
sys.path.insert(0, '/root/repo')
import torch
from pq3d_amd import _lib as L
from pq3d_amd import fused as F
dev = 'cuda'
it = int(sys.argv[1]) if len(sys.argv) > 1 else 5
H, d = 8, 256


def case(B, Lq, Lk, kpm=False, bias=False):
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
    for _ in range(it):
        F._attn(q, k, v, o, lse, H, L.BF16, kpm, **kw)
        F._attn(q, k, v, o, lse, H, L.BF16, kpm, bwd=(do, dq, dk, dv, delta, None), **kw)


case(24, 100, 1024, kpm=True)
case(8, 100, 100, bias=True)
torch.cuda.synchronize()
