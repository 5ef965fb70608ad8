"""Probe: what does the 3-D self-mask cost the attention kernels at config-4 shapes (B12 H8 Lq200 Lk4096 dh32)?"""
import sys, time
sys.path.insert(0, '/root/repo')
import torch
from pq3d_amd import ops, _lib as L
dev = 'cuda'
B, H, Lq, Lk, d = 12, 8, 200, 4096, 256
q = torch.randn(B, Lq, d, device=dev).bfloat16().requires_grad_(True)
k = torch.randn(B, Lk, d, device=dev).bfloat16().requires_grad_(True)
v = torch.randn(B, Lk, d, device=dev).bfloat16().requires_grad_(True)
mask = torch.rand(4, Lq, Lk, device=dev) < 0.5
row_open = torch.zeros(4, Lq, dtype=torch.bool, device=dev)
kpm = torch.zeros(B, Lk, dtype=torch.bool, device=dev)
def run(use_mask):
    from pq3d_amd import fused as F
    o = torch.empty_like(q); lse = torch.empty(B, H, Lq, device=dev)
    do = torch.randn_like(o); dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v); delta = torch.empty_like(lse)
    kw = dict(mask=mask, row_open=row_open, mask_bmod=4) if use_mask else dict(kpm=kpm)
    def fwd(): F._attn(q.detach(), k.detach(), v.detach(), o, lse, H, L.BF16, True, **kw)
    def bwd(): F._attn(q.detach(), k.detach(), v.detach(), o, lse, H, L.BF16, True, bwd=(do, dq, dk, dv, delta, None), **kw)
    res = []
    for fn in (fwd, bwd):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): fn()
        torch.cuda.synchronize(); res.append((time.perf_counter() - t) / 20 * 1e6)
    return res
print("3-D mask   fwd/bwd us:", run(True))
print("kpm only   fwd/bwd us:", run(False))
