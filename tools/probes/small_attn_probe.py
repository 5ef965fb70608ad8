import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pq3d_amd import _lib as L
from pq3d_amd import fused as F
dev = 'cuda'
H, d, B, Lq = 8, 256, 8, 100
q, k, v = (torch.randn(B, Lq, d, device=dev) for _ in range(3))
o = torch.empty_like(q); lse = torch.empty(B, H, Lq, device=dev)
bias = torch.randn(B, H, Lq, Lq, device=dev)
kpm = torch.zeros(B, Lq, dtype=torch.bool, device=dev)
fn = lambda: F._attn(q, k, v, o, lse, H, L.F32, False, kpm=kpm, bias=bias)
for _ in range(5): fn()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(50): fn()
e1.record(); torch.cuda.synchronize()
print("fwd us", e0.elapsed_time(e1) / 50 * 1e3)
do = torch.randn_like(o); dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v)
delta = torch.empty_like(lse); dsb = torch.empty_like(bias)
bw = lambda: F._attn(q, k, v, o, lse, H, L.F32, False, kpm=kpm, bias=bias, bwd=(do, dq, dk, dv, delta, dsb))
for _ in range(5): bw()
torch.cuda.synchronize(); e0.record()
for _ in range(50): bw()
e1.record(); torch.cuda.synchronize()
print("bwd us", e0.elapsed_time(e1) / 50 * 1e3)
