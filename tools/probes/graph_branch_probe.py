"""Probe: do forked branches of a captured HIP graph run concurrently on the GPU?  Two independent chains of small
(latency-bound) GEMM launches, captured (a) on one stream, (b) forked onto two streams; replay times compared."""
import sys, time
sys.path.insert(0, '/root/repo')
import torch
from pq3d_amd import _lib as L
dev = 'cuda'
M, N, K = 800, 256, 256
def mk():
    return (torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.empty(M, N, device=dev))
a1, b1, c1 = mk(); a2, b2, c2 = mk()
big_a, big_b, big_c = torch.randn(16384, 256, device=dev).bfloat16(), torch.randn(2048, 256, device=dev).bfloat16(), torch.empty(16384, 2048, device=dev, dtype=torch.bfloat16)
def chain(a, b, c, n=40):
    for _ in range(n):
        L.gemm(M=M, N=N, K=K, A=[a], B=[b], Cs=[c], ct=L.BF16, lda=K, ldb=K, ldc=N)
def big(n=6):
    for _ in range(n):
        L.gemm(M=16384, N=2048, K=256, A=[big_a], B=[big_b], Cs=[big_c], ct=L.BF16, lda=256, ldb=256, ldc=2048)
def timeit(g, n=20):
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
def capture(fn):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fn()
    return g
side = torch.cuda.Stream()
def serial_small(): chain(a1, b1, c1); chain(a2, b2, c2)
def forked_small():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side): chain(a2, b2, c2)
    chain(a1, b1, c1)
    cur.wait_stream(side)
def serial_mixed(): chain(a1, b1, c1); big()
def forked_mixed():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side): big()
    chain(a1, b1, c1)
    cur.wait_stream(side)
for name, fn in (("one chain of 40 small", lambda: chain(a1, b1, c1)), ("big x6", big), ("serial small+small", serial_small), ("forked small|small", forked_small),
                 ("serial small+big", serial_mixed), ("forked small|big", forked_mixed)):
    print(f"{name:24s} graph replay {timeit(capture(fn)):.3f} ms")
# eager two-stream
def eager(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
print("eager serial small+big", eager(serial_mixed), "forked", eager(forked_mixed))
