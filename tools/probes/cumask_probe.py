"""Probe (round 4): a CU-masked stream as the background lane.  (a) does the mask restrict a plain launch and a replayed
graph (bg time vs. number of CUs)?  (b) chain of small launches on the full chip || big GEMMs on the masked stream."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pq3d_amd import _lib as L

P = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "liboverlap_probe.so"))
P.op_stream_masked.restype = C.c_void_p
dev = "cuda"
vp = lambda t: C.c_void_p(t.data_ptr())
cs = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
out = torch.zeros(16, device=dev)


def masked_stream(ncu, pattern="low"):
    """pattern 'low': the first ncu bits; 'spread': every (256/ncu)-th bit."""
    bits = [0] * 256
    if pattern == "low":
        for i in range(ncu):
            bits[i] = 1
    else:
        step = 256 / ncu
        for i in range(ncu):
            bits[int(i * step)] = 1
    words = (C.c_uint32 * 8)(*[sum(bits[32 * w + b] << b for b in range(32)) for w in range(8)])
    p = P.op_stream_masked(words, 8)
    assert p, "hipExtStreamCreateWithCUMask failed"
    return torch.cuda.ExternalStream(p)


def capture(fn, s):
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        fn()
    return g


def time_ms(step, n=30):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


M, N, K = 800, 256, 256
a1, b1, c1 = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.empty(M, N, device=dev)
big_a = torch.randn(16384, 256, device=dev).bfloat16()
big_b = torch.randn(2048, 256, device=dev).bfloat16()
big_c = torch.empty(16384, 2048, device=dev, dtype=torch.bfloat16)
ga, gb_ = torch.randn(800, 2048, device=dev), torch.randn(800, 256, device=dev)
gw = torch.zeros(2048, 256, device=dev)
dkv = torch.randn(8192 * 4, 256, device=dev).bfloat16()
kin = torch.randn(8192 * 4, 256, device=dev).bfloat16()
gkv = torch.zeros(256, 256, device=dev)


def chain():
    for _ in range(40):
        L.gemm(M=M, N=N, K=K, A=[a1], B=[b1], Cs=[c1], ct=L.BF16X3, lda=K, ldb=K, ldc=N)


def bg_nt():
    for _ in range(6):
        L.gemm(M=16384, N=2048, K=256, A=[big_a], B=[big_b], Cs=[big_c], ct=L.BF16, lda=256, ldb=256, ldc=2048)


def bg_dw():
    for _ in range(12):
        L.gemm(M=2048, N=256, K=800, A=[ga], B=[gb_], Cs=[gw], ct=L.BF16, lda=2048, ldb=256, ldc=256, transA=True,
               transB=True, splitk=4, accumulate=True)


def bg_tt():   # K/V weight gradient style: [256,256] += dKV^T kin over 32768 rows (gemm_tt128), 8 groups
    L.gemm(M=256, N=256, K=8192 * 4, A=[dkv] * 8, B=[kin] * 8, Cs=[gkv] * 8, ct=L.BF16, lda=256, ldb=256, ldc=256,
           transA=True, transB=True, splitk=16, accumulate=True)


main = torch.cuda.Stream()
gC = capture(chain, main)
with torch.cuda.stream(main):
    tc = time_ms(gC.replay)
print(f"chain alone (40 gemm_wk launches, full chip): {tc:.3f} ms", flush=True)
for name, fn in (("nt128 x6", bg_nt), ("wktt dW x12", bg_dw), ("tt128 K/V dW", bg_tt)):
    with torch.cuda.stream(main):
        gfull = capture(fn, main)
        tfull = time_ms(gfull.replay)
    print(f"-- background = {name}: alone on the full chip {tfull:.3f} ms; serial chain+bg {tc + tfull:.3f} ms", flush=True)
    for ncu, pat in ((32, "low"), (64, "low"), (64, "spread"), (96, "spread"), (128, "low"), (128, "spread"), (192, "spread")):
        sB = masked_stream(ncu, pat)
        gB = capture(fn, sB)
        with torch.cuda.stream(sB):
            tb = time_ms(gB.replay)       # replayed graph on the masked stream
            te = time_ms(fn, n=10)        # eager launches on the masked stream

        def step():
            sB.wait_stream(main)
            with torch.cuda.stream(sB):
                gB.replay()
            with torch.cuda.stream(main):
                gC.replay()
                main.wait_stream(sB)
        tp = time_ms(step)
        print(f"   mask {ncu:3d} CUs ({pat:6s}): bg alone graph {tb:.3f} / eager {te:.3f} ms; chain || bg {tp:.3f} ms "
              f"(serial {tc + tfull:.3f}, ideal {max(tc, tb):.3f})", flush=True)
