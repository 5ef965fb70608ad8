"""Target for rocprofv3 --pmc runs: the hoisted K/V projection GEMM shape (24 groups of [8192x256] x [256x256]^T) alone."""
import sys
sys.path.insert(0, '/root/repo')
import torch
from pq3d_amd import _lib as L
dev = 'cuda'
G, M, N, K = 24, 8192, 256, 256
A = [torch.randn(M, K, device=dev).bfloat16() for _ in range(3)]
Wf = [torch.randn(N, K, device=dev) for _ in range(G)]
C = torch.empty(G, M, N, device=dev, dtype=torch.bfloat16)
As = [A[g % 3] for g in range(G)]
for _ in range(5):
    L.gemm(M=M, N=N, K=K, A=As, B=Wf, Cs=[C[g] for g in range(G)], ct=L.BF16, lda=K, ldb=K, ldc=N)
torch.cuda.synchronize()
