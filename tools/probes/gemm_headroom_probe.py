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
# ---- restructured hoisted projection: K and V inputs differ -> 6 batched GEMMs [8192x256] x [256x1024]
A6 = torch.randn(6, M, K, device=dev).bfloat16(); W6 = torch.randn(6, 1024, K, device=dev).bfloat16()
o6 = torch.empty(6, M, 1024, device=dev, dtype=torch.bfloat16)
print("torch bmm 6 x [8192x256]x[256x1024] bf16:", timeit(lambda: torch.bmm(A6, W6.transpose(1, 2), out=o6)), "us")
b6 = torch.randn(6, 1, 1024, device=dev).bfloat16()
print("torch baddbmm (bias) same shape        :", timeit(lambda: torch.baddbmm(b6, A6, W6.transpose(1, 2), out=o6)), "us")
Cs6 = torch.empty(6, M, 1024, device=dev, dtype=torch.bfloat16)
Wf6 = [W6[i].float() for i in range(6)]
print("ours  6 groups N=1024 W fp32            :", timeit(lambda: L.gemm(M=M, N=1024, K=K, A=[A6[i] for i in range(6)], B=Wf6, Cs=[Cs6[i] for i in range(6)], ct=L.BF16, lda=K, ldb=K, ldc=1024)), "us")
print("ours  6 groups N=1024 W bf16            :", timeit(lambda: L.gemm(M=M, N=1024, K=K, A=[A6[i] for i in range(6)], B=[W6[i] for i in range(6)], Cs=[Cs6[i] for i in range(6)], ct=L.BF16, lda=K, ldb=K, ldc=1024)), "us")
# hoisted dX: dfeat = dKVcat [8192 x 1024] x Wcat [1024 x 256], 6 of them summed pairwise (K and V) -> 3 x K=2048
dk = torch.randn(3, M, 2048, device=dev).bfloat16(); w3 = torch.randn(3, 2048, K, device=dev).bfloat16()
dx3 = torch.empty(3, M, K, device=dev, dtype=torch.bfloat16)
print("torch bmm dX 3 x [8192x2048]x[2048x256] :", timeit(lambda: torch.bmm(dk, w3, out=dx3)), "us")
# hoisted dW: 6 x [1024 x 8192] x [8192 x 256]
dk6 = torch.randn(6, M, 1024, device=dev).bfloat16()
dw6 = torch.empty(6, 1024, K, device=dev, dtype=torch.bfloat16)
print("torch bmm dW 6 x [1024x8192]x[8192x256] :", timeit(lambda: torch.bmm(dk6.transpose(1, 2), A6, out=dw6)), "us")
try:
    print("torch bmm dW fp32 out                   :", timeit(lambda: torch.bmm(dk6.transpose(1, 2), A6, out_dtype=torch.float32)), "us")
except Exception as e:
    print("bmm out_dtype unsupported:", type(e).__name__, str(e)[:80])
