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
buf = (C.c_longlong * 16)()
lib = C.CDLL(L.LIB_PATH)
lib.pq3d_small_debug_read(buf)
t = list(buf)[:6]
print("stamps (cycles):", [t[i + 1] - t[i] for i in range(5)], "total", t[5] - t[0])
