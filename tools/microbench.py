#!/usr/bin/env python3
"""Per-op latency of the C-ABI kernels at the BASELINE config-2 shapes: each op is captured 20x into a HIP graph
and replayed, so the number is back-to-back GPU time per launch (no Python / dispatch gaps)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pq3d_amd import ops  # noqa: E402
from pq3d_amd._lib import BF16  # noqa: E402

DEV = "cuda"
REP = 20


def bench(name, fn, flops=0.0, nbytes=0.0):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REP):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (10 * REP)
    print(f"{name:46s} {us:9.2f} us  {flops / us / 1e6 if flops else 0:8.2f} TFLOP/s  {nbytes / us / 1e3 if nbytes else 0:8.1f} GB/s")


def main():
    B, Ns, Nq, d, H, F = 8, 1024, 100, 256, 8, 2048
    bf = torch.bfloat16
    x = torch.randn(B, Nq, d, device=DEV)
    qp = torch.randn(B, Nq, d, device=DEV)
    feat = torch.randn(B, Ns, d, device=DEV)
    pos = torch.randn(B, Ns, d, device=DEV)
    w = torch.randn(d, d, device=DEV) * 0.05
    bias = torch.randn(d, device=DEV) * 0.02
    w1 = torch.randn(F, d, device=DEV) * 0.05
    w2 = torch.randn(d, F, device=DEV) * 0.05
    with torch.no_grad():
        bench("linear  800x256x256 (x+qpos fp32 -> bf16)", lambda: ops.linear(x, w, bias, x2=qp, ct=BF16, out_dtype=bf),
              2.0 * 800 * d * d)
        bench("linear 8192x256x256 (feat+pos fp32 -> bf16)", lambda: ops.linear(feat, w, bias, x2=pos, ct=BF16, out_dtype=bf),
              2.0 * 8192 * d * d)
        bench("linear  800x256x2048 relu -> bf16", lambda: ops.linear(x, w1, None, ct=BF16, act="relu", out_dtype=bf),
              2.0 * 800 * d * F)
        h = torch.randn(B, Nq, F, device=DEV).to(bf)
        bench("linear  800x2048x256 bf16 -> fp32", lambda: ops.linear(h, w2, bias, ct=BF16), 2.0 * 800 * d * F)
        q = torch.randn(B, Nq, d, device=DEV).to(bf)
        k = torch.randn(B, Ns, d, device=DEV).to(bf)
        v = torch.randn(B, Ns, d, device=DEV).to(bf)
        kpm = torch.arange(Ns, device=DEV)[None, :] >= torch.randint(Ns // 2, Ns, (B, 1), device=DEV)
        bench("attention fwd B8 H8 Lq100 Lk1024 dh32", lambda: ops.attention(q, k, v, H=H, ct=BF16, zero_attn=True, kpm=kpm),
              4.0 * B * Nq * Ns * d)
        o = [torch.randn(B, Nq, d, device=DEV) for _ in range(3)]
        gam = [torch.ones(d, device=DEV)] * 3
        bet = [torch.zeros(d, device=DEV)] * 3
        bench("add_ln fwd R800 d256 M3", lambda: ops.add_layernorm(x, o, gam, bet), 0, 5 * 800 * d * 4.0)
        bench("add_ln fwd R8192 d256 M1 (no x)", lambda: ops.add_layernorm(None, [feat], gam[:1], bet[:1]), 0, 2 * 8192 * d * 4.0)
    # backward pieces (through autograd, graph-captured)
    xg = x.clone().requires_grad_(True)
    wg = w.clone().requires_grad_(True)
    bg = bias.clone().requires_grad_(True)
    featg = feat.clone().requires_grad_(True)

    def lin_bwd(inp, rows):
        def f():
            y = ops.linear(inp, wg, bg, ct=BF16, out_dtype=bf)
            y.backward(torch.ones_like(y))
            inp.grad = None; wg.grad = None; bg.grad = None
        return f
    bench("linear fwd+bwd  800x256x256", lin_bwd(xg, 800), 6.0 * 800 * d * d)
    bench("linear fwd+bwd 8192x256x256", lin_bwd(featg, 8192), 6.0 * 8192 * d * d)
    qg, kg, vg = (t.clone().requires_grad_(True) for t in (q, k, v))

    def attn_fb():
        o_ = ops.attention(qg, kg, vg, H=H, ct=BF16, zero_attn=True, kpm=kpm)
        o_.backward(torch.ones_like(o_))
        qg.grad = None; kg.grad = None; vg.grad = None
    bench("attention fwd+bwd B8 H8 Lq100 Lk1024", attn_fb, 18.0 * B * Nq * Ns * d)
    og = [t.clone().requires_grad_(True) for t in o]

    def ln_fb():
        y = ops.add_layernorm(xg, og, gam, bet)
        y.backward(torch.ones_like(y))
        xg.grad = None
        for t in og:
            t.grad = None
    bench("add_ln fwd+bwd R800 d256 M3", ln_fb)

    def ln8_fb():
        y = ops.add_layernorm(None, [featg], gam[:1], bet[:1])
        y.backward(torch.ones_like(y))
        featg.grad = None
    bench("add_ln fwd+bwd R8192 d256 M1", ln8_fb)


if __name__ == "__main__":
    main()
