"""Probe: are hipMemsetAsync nodes of a captured HIP graph re-executed on every replay?  (They were not on ROCm 7.x /
torch 2.10: a split-K GEMM whose outputs the library zeroed with hipMemsetAsync was only correct on the first replay.)
The library now zero-fills with its own kernel (csrc/common.h ZeroList); with that build this probe prints zeros."""
import sys; sys.path.insert(0,'/root/repo')
import torch
from pq3d_amd import _lib as L
dev='cuda'
torch.manual_seed(0)
G=2; N=64; K=64; R=192
dlin=[torch.randn(R,N,device=dev) for _ in range(G)]
xs=[torch.randn(R,K,device=dev) for _ in range(G)]
dWs=[torch.empty(N,K,device=dev) for _ in range(G)]
dbl=[torch.empty(N,device=dev) for _ in range(G)]
def run():
    L.gemm(M=N, N=K, K=R, A=dlin, B=xs, Cs=dWs, ct=L.F32, lda=N, ldb=K, ldc=K, transA=True, transB=True, splitk=3, colsum=dbl)
run(); torch.cuda.synchronize()
ref=[d.clone() for d in dbl]; refw=[w.clone() for w in dWs]
s=torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s): run()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g=torch.cuda.CUDAGraph()
with torch.cuda.graph(g): run()
for i in range(4):
    for t in dbl+dWs: t.fill_(float('inf') if i%2 else 7.0)
    g.replay(); torch.cuda.synchronize()
    print(i, [float((a-b).abs().max()) for a,b in zip(dbl,ref)], [float((a-b).abs().max()) for a,b in zip(dWs,refw)])
