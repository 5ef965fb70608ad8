"""Five launches against the one-launch chain (csrc/chain_ffn.hip) inside HIP graphs: us per layer tail.
usage (GPU box): python tools/probes/chain_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from pq3d_amd import ops
from test_gpu_chain import _five_launches

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
    print(f"R = {B * Nq}: five launches {res['five launches']:.2f} us, chain {res['chain']:.2f} us per layer tail; hand-off timeouts: {ops.chain_error(dev)}")
