"""Target for rocprofv3 --pmc runs: the hoisted K/V projection shape (24 groups of [8192x256] x [256x256]^T -> bf16) on
the 64x64-tile kernel (fp32 weights converted in flight) and on the 128x128-tile kernel (weights pre-cast to bf16),
plus the weight-gradient shape on the TT 128-tile kernel."""
import sys
sys.path.insert(0, '/root/repo')
import torch
from pq3d_amd import _lib as L
dev = 'cuda'
G, M, N, K = 24, 8192, 256, 256
A = [torch.randn(M, K, device=dev).bfloat16() for _ in range(3)]
Wf = [torch.randn(N, K, device=dev) for _ in range(G)]
Wb = [w.bfloat16() for w in Wf]
C = torch.empty(G, M, N, device=dev, dtype=torch.bfloat16)
As = [A[g % 3] for g in range(G)]
gs = [torch.randn(M, N, device=dev).bfloat16() for _ in range(G)]
dW = torch.zeros(G, N, K, device=dev)
for _ in range(5):
    L.gemm(M=M, N=N, K=K, A=As, B=Wf, Cs=[C[g] for g in range(G)], ct=L.BF16, lda=K, ldb=K, ldc=N)
    L.gemm(M=M, N=N, K=K, A=As, B=Wb, Cs=[C[g] for g in range(G)], ct=L.BF16, lda=K, ldb=K, ldc=N)
    L.gemm(M=N, N=K, K=M, A=gs, B=As, Cs=[dW[g] for g in range(G)], ct=L.BF16, lda=N, ldb=K, ldc=K, transA=True, transB=True,
           splitk=2, accumulate=True)
torch.cuda.synchronize()
