"""Five launches against the one-launch chain (csrc/chain_ffn.hip) inside HIP graphs: us per layer tail.
usage (GPU box): python tools/probes/chain_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from pq3d_amd import ops
from test_gpu_chain import _five_launches, _three_launches

dev = torch.device("cuda")
for B, Nq in ((8, 100), (16, 100)):
    g = torch.Generator().manual_seed(0)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    d, F_ = 256, 2048
    args = (r(B, Nq, d), r(d, d, sc=0.06), r(d, sc=0.1), r(B, Nq, d), 1 + r(d, sc=0.1), r(d, sc=0.1), 1e-5,
            r(F_, d, sc=0.06), r(F_, sc=0.1), r(d, F_, sc=0.03), r(d, sc=0.1), 1 + r(d, sc=0.1), r(d, sc=0.1), 1e-5)
    flags = ops.chain_flags(B * Nq, dev)
    res = {}
    for name, fn in (("five launches", lambda: _five_launches(*args)), ("chain", lambda: ops.chain_ffn_fwd(*args, flags))):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                for _ in range(20):
                    keep = fn()
            for _ in range(5):
                gr.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                gr.replay()
            e1.record()
            torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 400 * 1e3
    M = 3
    a2 = (r(M, B, Nq, d).bfloat16(), [r(d, d, sc=0.06) for _ in range(M)], [r(d, sc=0.1) for _ in range(M)], r(B, Nq, d),
          [1 + r(d, sc=0.1) for _ in range(M)], [r(d, sc=0.1) for _ in range(M)], 1e-5, None, Nq, r(B, Nq, d),
          [r(d, d, sc=0.06) for _ in range(3)], [r(d, sc=0.1) for _ in range(3)])
    fl2 = ops.chain_flags(B * Nq, dev)
    for name, fn in (("three launches", lambda: _three_launches(*a2)), ("chain_ca", lambda: ops.chain_ca_fwd(*a2, fl2))):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                for _ in range(20):
                    keep = fn()
            for _ in range(5):
                gr.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                gr.replay()
            e1.record()
            torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 400 * 1e3
    print(f"R = {B * Nq}: three launches {res['three launches']:.2f} us, chain_ca {res['chain_ca']:.2f} us")
    print(f"R = {B * Nq}: five launches {res['five launches']:.2f} us, chain {res['chain']:.2f} us per layer tail; hand-off timeouts: {ops.chain_error(dev)}")
