"""One configuration of the resident attention backward, a few launches (target of rocprofv3 runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pq3d_amd import _lib as L
from pq3d_amd import fused as F
dev = 'cuda'
H, d = 8, 256
B, Lq, Lk = 24, 100, 1024
q = torch.randn(B, Lq, d, device=dev).bfloat16(); k = torch.randn(B, Lk, d, device=dev).bfloat16()
v = torch.randn(B, Lk, d, device=dev).bfloat16()
o = torch.empty_like(q); lse = torch.empty(B, H, Lq, device=dev)
do = torch.randn_like(o); dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v)
delta = torch.empty_like(lse)
vl = torch.randint(Lk // 2, Lk + 1, (B,)); vl[0] = Lk
kw = dict(kpm=(torch.arange(Lk)[None] >= vl[:, None]).to(dev))
F._attn(q, k, v, o, lse, H, L.BF16, True, **kw)
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
L.lib().pq3d_attn_resident(mode)
for _ in range(10):
    F._attn(q, k, v, o, lse, H, L.BF16, True, bwd=(do, dq, dk, dv, delta, None), **kw)
torch.cuda.synchronize()
