"""Probe: cross-attention forward at the config-2/5 shapes, resident vs streaming kernel, in a HIP graph."""
import sys, time, math
sys.path.insert(0, '/root/repo')
import torch
from pq3d_amd import ops, _lib as L
dev = 'cuda'


def graph_time(fn, n=20, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps / n * 1e6


for (B, Lq, Lk) in ((24, 100, 1024), (48, 100, 2048), (24, 100, 512)):
    H, d = 8, 256
    q, k, v = [torch.randn(B, n, d, device=dev).bfloat16() for n in (Lq, Lk, Lk)]
    kpm = torch.zeros(B, Lk, dtype=torch.uint8, device=dev)
    for mode in (7, 3):
        old = L.lib().pq3d_attn_resident(mode)
        with torch.no_grad():
            t = graph_time(lambda: ops.attention(q, k, v, H=H, ct=L.BF16, zero_attn=True, kpm=kpm))
        L.lib().pq3d_attn_resident(old)
        print(f"B{B} Lq{Lq} Lk{Lk} mode {mode}: fwd {t:6.1f} us")
