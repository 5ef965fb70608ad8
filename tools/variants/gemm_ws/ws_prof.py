"""One launch shape of the weight-stationary projection for rocprofv3 --pmc passes.  usage: ws_prof.py <mode> <G> <R>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pq3d_amd import _lib as L
mode, G, R = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = "cuda"
xs = [torch.randn(R, 256, device=dev).bfloat16() for _ in range(3)]
ws = [(torch.randn(256, 256, device=dev) * 0.06).bfloat16() for _ in range(G)]
bs = [torch.randn(256, device=dev) for _ in range(G)]
C = torch.empty(G, R, 256, device=dev, dtype=torch.bfloat16)
L.lib().pq3d_gemm_ws(mode)
for _ in range(5):
    L.gemm(M=R, N=256, K=256, A=[xs[g % 3] for g in range(G)], B=ws, bias=bs, Cs=[C[g] for g in range(G)], ct=L.BF16, lda=256, ldb=256, ldc=256)
torch.cuda.synchronize()
