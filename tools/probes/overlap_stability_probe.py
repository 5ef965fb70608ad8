"""Probe (round 4): is the two-stream overlap stable?  The first probes showed occasional 4x-slower-than-serial results.
Fixed stream pairs, repeated measurements, with / without cross-stream joins, per-iteration times."""
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
print("GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"), flush=True)


def masked_stream(ncu):
    bits = [1 if i < ncu else 0 for i in range(256)]
    words = (C.c_uint32 * 8)(*[sum(bits[32 * w + b] << b for b in range(32)) for w in range(8)])
    p = P.op_stream_masked(words, 8)
    assert p
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


M, N, K = 800, 256, 256
a1, b1, c1 = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.empty(M, N, device=dev)
big_a = torch.randn(16384, 256, device=dev).bfloat16()
big_b = torch.randn(2048, 256, device=dev).bfloat16()
big_c = torch.empty(16384, 2048, device=dev, dtype=torch.bfloat16)


def chain():
    for _ in range(40):
        L.gemm(M=M, N=N, K=K, A=[a1], B=[b1], Cs=[c1], ct=L.BF16X3, lda=K, ldb=K, ldc=N)


def bg_nt():
    for _ in range(6):
        L.gemm(M=16384, N=2048, K=256, A=[big_a], B=[big_b], Cs=[big_c], ct=L.BF16, lda=256, ldb=256, ldc=2048)


def run(step, n=40):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t = time.perf_counter()
        step()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) * 1e3)
    ts.sort()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return ts[0], ts[len(ts) // 2], ts[-1], (time.perf_counter() - t) / n * 1e3


main = torch.cuda.Stream()
gC = capture(chain, main)
for label, mk in (("plain side stream", lambda: torch.cuda.Stream()), ("masked 128", lambda: masked_stream(128)),
                  ("masked 96", lambda: masked_stream(96)), ("plain side stream #2", lambda: torch.cuda.Stream()),
                  ("masked 128 #2", lambda: masked_stream(128))):
    sB = mk()
    gB = capture(bg_nt, sB)

    def joined():
        sB.wait_stream(main)
        with torch.cuda.stream(sB):
            gB.replay()
        with torch.cuda.stream(main):
            gC.replay()
            main.wait_stream(sB)

    def free():
        with torch.cuda.stream(sB):
            gB.replay()
        with torch.cuda.stream(main):
            gC.replay()

    def serial():
        with torch.cuda.stream(main):
            gB.replay()
            gC.replay()

    def eager_joined():     # no graphs: plain launches on the two streams
        sB.wait_stream(main)
        with torch.cuda.stream(sB):
            bg_nt()
        with torch.cuda.stream(main):
            chain()
            main.wait_stream(sB)
    for rep in range(3):
        for nm, fn in (("serial", serial), ("joined", joined), ("free", free), ("eager_joined", eager_joined)):
            mn, md, mx, back = run(fn)
            print(f"{label:22s} rep{rep} {nm:13s} single-step min/med/max {mn:.3f}/{md:.3f}/{mx:.3f} ms; back-to-back {back:.3f} ms/step",
                  flush=True)
