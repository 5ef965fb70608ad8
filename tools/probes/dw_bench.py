import sys, time
sys.path.insert(0, '/root/repo')
import torch
from pq3d_amd import _lib as L
dev='cuda'
G,R,M,N=24,8192,256,256
gs=[torch.randn(R,M,device=dev).bfloat16() for _ in range(G)]
xs=[torch.randn(R,N,device=dev).bfloat16() for _ in range(3)]
dW=torch.zeros(G,M,N,device=dev); cb=torch.zeros(G,M,device=dev)
def f(): L.gemm(M=M,N=N,K=R,A=gs,B=[xs[g%3] for g in range(G)],Cs=[dW[g] for g in range(G)],ct=L.BF16,lda=M,ldb=N,ldc=N,transA=True,transB=True,splitk=2,accumulate=True,colsum=[cb[g] for g in range(G)])
for _ in range(5): f()
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(50): f()
torch.cuda.synchronize(); print((time.perf_counter()-t)/50*1e6,"us")
