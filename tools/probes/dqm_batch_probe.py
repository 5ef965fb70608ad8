import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pq3d_amd import _lib as L
dev = torch.device("cuda")
B, Ns, Nq, d, Mm, NC = 4, 4096, 200, 256, 3, 5
g = [torch.randn(B, Ns, Nq, device=dev).bfloat16() for _ in range(NC)]
keys = [torch.randn(B, Ns, d, device=dev).bfloat16() for _ in range(Mm)]
out = torch.zeros(NC, Mm, B, Nq, d, device=dev)
def sep():
    for c in range(NC):
        L.gemm(M=Nq, N=d, K=Ns, A=[g[c]] * Mm, B=keys, Cs=[out[c, m] for m in range(Mm)], ct=L.BF16, lda=Nq, ldb=d, ldc=d, transA=True,
               transB=True, batch=B, strideA=Ns * Nq, strideB=Ns * d, strideC=Nq * d, splitk=8, accumulate=True)
def one():
    L.gemm(M=Nq, N=d, K=Ns, A=[g[c] for c in range(NC) for m in range(Mm)], B=keys * NC, Cs=[out[c, m] for c in range(NC) for m in range(Mm)],
           ct=L.BF16, lda=Nq, ldb=d, ldc=d, transA=True, transB=True, batch=B, strideA=Ns * Nq, strideB=Ns * d, strideC=Nq * d, splitk=8, accumulate=True)
def time(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(10): fn()
        for _ in range(3): gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): gr.replay()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 200 * 1e3
print(f"five launches {time(sep):.1f} us, one launch of 15 groups {time(one):.1f} us")
