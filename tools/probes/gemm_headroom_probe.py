"""Probe: how far is the grouped 64x64-tile GEMM from what the shapes allow?  Times the hoisted K/V projection shape
(24 groups of [8192x256] x [256x256]^T -> bf16) against torch.matmul (hipBLASLt) on equivalent single-GEMM shapes."""
import sys, time
sys.path.insert(0, '/root/repo')
import torch
from pq3d_amd import _lib as L
dev = 'cuda'
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
G, M, N, K = 24, 8192, 256, 256
A = [torch.randn(M, K, device=dev).bfloat16() for _ in range(3)]
Wf = [torch.randn(N, K, device=dev) for _ in range(G)]
Wb = [w.bfloat16() for w in Wf]
C = torch.empty(G, M, N, device=dev, dtype=torch.bfloat16)
As = [A[g % 3] for g in range(G)]
print("ours  A bf16, W fp32 :", timeit(lambda: L.gemm(M=M, N=N, K=K, A=As, B=Wf, Cs=[C[g] for g in range(G)], ct=L.BF16, lda=K, ldb=K, ldc=N)), "us")
print("ours  A bf16, W bf16 :", timeit(lambda: L.gemm(M=M, N=N, K=K, A=As, B=Wb, Cs=[C[g] for g in range(G)], ct=L.BF16, lda=K, ldb=K, ldc=N)), "us")
# hipBLASLt equivalents: 3 GEMMs [8192,256] x [256, 2048] (8 weight matrices concatenated along N per memory)
Wcat = [torch.cat([Wb[g] for g in range(m, G, 3)], 0) for m in range(3)]   # [2048, 256]
out = [torch.empty(M, 2048, device=dev, dtype=torch.bfloat16) for _ in range(3)]
def blas():
    for m in range(3): torch.matmul(A[m], Wcat[m].t(), out=out[m])
print("torch 3 x [8192x256]x[256x2048] bf16:", timeit(blas), "us")
big = torch.randn(24576, 256, device=dev).bfloat16(); wbig = torch.randn(2048, 256, device=dev).bfloat16(); obig = torch.empty(24576, 2048, device=dev, dtype=torch.bfloat16)
print("torch 1 x [24576x256]x[256x2048] bf16:", timeit(lambda: torch.matmul(big, wbig.t(), out=obig)), "us")
# weight-gradient shape: [256 x 8192] x [8192 x 256] x 24
g = torch.randn(8192, 256, device=dev).bfloat16(); x = torch.randn(8192, 256, device=dev).bfloat16(); dw = torch.empty(256, 256, device=dev, dtype=torch.float32)
print("torch dW [256x8192]x[8192x256] bf16 x24:", timeit(lambda: [torch.matmul(g.t(), x) for _ in range(24)]), "us")
