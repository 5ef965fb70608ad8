import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pq3d_amd import _lib as L
from pq3d_amd._lib import BF16
dev="cuda"
def t(fn, rep=20, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(rep): fn()
    g.replay(); torch.cuda.synchronize()
    e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e3/(rep*it)
print("empty-ish torch kernel:", t(lambda: torch.empty(1,device=dev).fill_(1.0)))
for (M,N,K,dta,dtb,dtc,bias) in [(64,64,64,'f','f','b',0),(64,64,256,'f','f','b',0),(800,256,64,'f','f','b',0),(800,256,256,'f','f','b',0),(800,256,256,'f','f','b',1),(800,256,256,'b','f','f',1),(800,256,256,'b','b','b',0),(800,256,512,'f','f','b',0),(800,256,1024,'f','f','b',0),(800,256,2048,'b','f','f',1),(3200,256,256,'f','f','b',0),(8192,256,256,'f','f','b',1),(8192,256,256,'b','b','b',0),(8192,1024,256,'b','b','b',0)]:
    dt = lambda c: torch.float32 if c=='f' else torch.bfloat16
    A = torch.randn(M,K,device=dev).to(dt(dta)); B = (torch.randn(N,K,device=dev)*0.05).to(dt(dtb)); Cc = torch.empty(M,N,device=dev,dtype=dt(dtc)); bi = torch.zeros(N,device=dev) if bias else None
    f = lambda: L.gemm(M=M,N=N,K=K,A=[A],B=[B],bias=[bi],Cs=[Cc],ct=BF16,lda=K,ldb=K,ldc=N)
    us = t(f)
    print(f"M{M:5d} N{N:5d} K{K:5d} A:{dta} B:{dtb} C:{dtc} bias:{bias}  {us:8.2f} us  {2*M*N*K/us/1e6:8.2f} TF")
