"""Does a chain launch run slower when OTHER kernels ran since its last launch (cold instruction cache / cold per-kernel state)?
Graph A: 20 x chain_ffn_fwd.  Graph B: 20 x chain_ffn_bwd.  Graph C: 20 x (chain_ffn_fwd, chain_ffn_bwd) alternating, same data.
If C > A + B per pair, the difference is what a launch pays for not being the kernel that ran last.
usage (GPU box): python tools/probes/chain_icache.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pq3d_amd import ops
from tests.test_gpu_chain import _five_launches

dev = torch.device("cuda")
B, Nq, d, F_ = 8, 100, 256, 2048
g = torch.Generator().manual_seed(0)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
args = (r(B, Nq, d), r(d, d, sc=0.06), r(d, sc=0.1), r(B, Nq, d), 1 + r(d, sc=0.1), r(d, sc=0.1), 1e-5,
        r(F_, d, sc=0.06), r(F_, sc=0.1), r(d, F_, sc=0.03), r(d, sc=0.1), 1 + r(d, sc=0.1), r(d, sc=0.1), 1e-5)
flags = ops.chain_flags(B * Nq, dev)
f_, x2, mean1, rstd1, h, zp, z, x3, mean2, rstd2 = _five_launches(*args)
x1s, g1, W1, W2, g2 = args[3], args[4], args[7], args[9], args[11]
dx = r(B, Nq, d)
zs = lambda: torch.zeros(d, device=dev)
acc = [zs() for _ in range(4)]
fl3 = ops.chain_flags(B * Nq, dev)
m2, r2 = mean2[:1].contiguous(), rstd2[:1].contiguous()
fwd = lambda: ops.chain_ffn_fwd(*args, flags)
bwd = lambda: ops.chain_ffn_bwd(dx, x2, z, g2, m2, r2, acc[0], acc[1], W2, h, W1, x1s, f_, g1, mean1, rstd1, acc[2], acc[3], fl3)
# a launch with a large, different code footprint and little work: the attention-free GEMM family on a tiny problem
xa, wa = r(64, 256), r(256, 256, sc=0.06)
other = lambda: ops.linear(xa, wa, None, ct=ops.BF16X3)


def time(fns, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            for f in fns:
                f()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(reps):
                keep = [f() for f in fns]
        for _ in range(5):
            gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (20 * reps) * 1e3


a, b, o = time([fwd]), time([bwd]), time([other])
c = time([fwd, bwd])
e = time([fwd, other])
print(f"chain_ffn_fwd alone {a:.2f} us, chain_ffn_bwd alone {b:.2f} us, small gemm alone {o:.2f} us")
print(f"fwd + bwd alternating {c:.2f} us per pair (sum of the two alone: {a + b:.2f})")
print(f"fwd + small gemm alternating {e:.2f} us per pair (sum alone: {a + o:.2f})")
