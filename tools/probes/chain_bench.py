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
    # ---- backward chains against their separate launches
    from pq3d_amd import _lib as L, fused
    f_, x2, mean1, rstd1, h, zp, z, x3, mean2, rstd2 = _five_launches(*args)
    x1s, g1, W1, W2, g2 = args[3], args[4], args[7], args[9], args[11]
    dx = r(B, Nq, d)
    R = B * Nq
    zs = lambda: torch.zeros(d, device=dev)
    acc = [zs() for _ in range(4)]
    fl3 = ops.chain_flags(R, dev)
    m2, r2 = mean2[:1].contiguous(), rstd2[:1].contiguous()
    def sep_ffn_bwd():
        dx2r, dyl = fused._ln_bwd(x2, [z], [g2], [args[12]], 1e-5, None, Nq, m2, r2, dx, [acc[0]], [acc[1]])
        dhp = torch.empty(B, Nq, F_, dtype=torch.bfloat16, device=dev)
        L.gemm(M=R, N=F_, K=d, A=[dyl[0]], B=[W2], Cs=[dhp], aux=[h], act_grad="relu", ct=L.BF16, lda=d, ldb=F_, ldc=F_, transB=True)
        L.gemm(M=R, N=d, K=F_, A=[dhp], B=[W1], Cs=[dx2r], ct=L.BF16, lda=F_, ldb=d, ldc=d, transB=True, splitk=4, accumulate=True)
        return fused._ln_bwd(x1s, [f_], [g1], [args[5]], 1e-5, None, Nq, mean1, rstd1, dx2r, [acc[2]], [acc[3]])
    def chain_ffn_bwd():
        return ops.chain_ffn_bwd(dx, x2, z, g2, m2, r2, acc[0], acc[1], W2, h, W1, x1s, f_, g1, mean1, rstd1, acc[2], acc[3], fl3)
    op_all, x1, meanc, rstdc, qkv = _three_launches(*a2)
    dqkv, dx1r = r(3, B, Nq, d), r(B, Nq, d)
    dgs, dbs = [zs() for _ in range(M)], [zs() for _ in range(M)]
    dxz = torch.zeros(B, Nq, d, device=dev)
    fl4 = ops.chain_flags(R, dev)
    def sep_sa_bwd():
        g3 = torch.empty(3, B, Nq, d, device=dev)
        L.gemm(M=R, N=d, K=d, A=[dqkv[0], dqkv[1], dqkv[2]], B=list(a2[10]), Cs=[g3[0], g3[1], g3[2]], aux=[None, None, dx1r], act_grad="add",
               ct=L.BF16, lda=d, ldb=d, ldc=d, transB=True)
        dxr, dop = fused._ln_bwd(a2[3], [op_all[m] for m in range(M)], a2[4], a2[5], 1e-5, None, Nq, meanc, rstdc, [g3[0], g3[1], g3[2]], dgs, dbs,
                                 dx_zeroed=dxz)
        do = torch.empty(M, B, Nq, d, dtype=torch.bfloat16, device=dev)
        L.gemm(M=R, N=d, K=d, A=[dop[m] for m in range(M)], B=list(a2[1]), Cs=[do[m] for m in range(M)], ct=L.BF16, lda=d, ldb=d, ldc=d, transB=True)
        return do
    def chain_sa_bwd():
        return ops.chain_sa_bwd(dqkv, a2[10], dx1r, a2[3], op_all, a2[4], meanc, rstdc, None, Nq, dgs, dbs, a2[1], fl4)
    for name, fn in (("sep_ffn_bwd", sep_ffn_bwd), ("chain_ffn_bwd", chain_ffn_bwd), ("sep_sa_bwd", sep_sa_bwd), ("chain_sa_bwd", chain_sa_bwd)):
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
    print(f"R = {B * Nq}: backward FFN group: four launches {res['sep_ffn_bwd']:.2f} us, chain {res['chain_ffn_bwd']:.2f} us; "
          f"q/k/v + merged LN + dO group: three launches {res['sep_sa_bwd']:.2f} us, chain {res['chain_sa_bwd']:.2f} us")
    print(f"R = {B * Nq}: five launches {res['five launches']:.2f} us, chain {res['chain']:.2f} us per layer tail; hand-off timeouts: {ops.chain_error(dev)}")

# ---- the mask head's chains (csrc/chain_mh.hip) against their separate launches, config 4's shape
from test_gpu_chain import _mh_five_launches, _mh_bwd_six_launches


def _time(fn):
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
    return e0.elapsed_time(e1) / 400 * 1e3


for B, Nq in ((4, 200), (8, 100)):
    g = torch.Generator().manual_seed(1)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    d, C_, Mm = 256, 201, 3
    cols = torch.tensor([0, 7], dtype=torch.int32, device=dev)
    cf = torch.zeros(C_, dtype=torch.int32, device=dev)
    cf[cols.long()] = 1
    x, W0, b0, gamma, beta = r(B, Nq, d), r(d, d, sc=0.06), r(d, sc=0.1), 1 + r(d, sc=0.1), r(d, sc=0.1)
    W4, b4 = r(C_, d, sc=0.06), r(C_, sc=0.1)
    Wqs, bqs = [r(d, d, sc=0.06) for _ in range(Mm)], [r(d, sc=0.1) for _ in range(Mm)]
    fl = ops.chain_flags(B * Nq, dev)
    t_sep = _time(lambda: _mh_five_launches(x, W0, b0, gamma, beta, 1e-5, W4, b4, cols, Wqs, bqs))
    t_ch = _time(lambda: ops.chain_mh_fwd(x, W0, b0, gamma, beta, 1e-5, W4, b4, cf, float("-inf"), Wqs, bqs, fl))
    h1, h2, mean, rstd, cls, qm = ops.chain_mh_fwd(x, W0, b0, gamma, beta, 1e-5, W4, b4, cf, float("-inf"), Wqs, bqs, fl)
    dc, cur, dqs = r(B, Nq, C_), r(B, Nq, d), [r(B, Nq, d) for _ in range(Mm)]
    dg, db = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
    fb = ops.chain_flags(B * Nq, dev)
    t_sepb = _time(lambda: _mh_bwd_six_launches(dc, cols, W4, h1, mean, rstd, gamma, dg, db, W0, cur, dqs, Wqs))
    t_chb = _time(lambda: ops.chain_mh_bwd(dc, cf, W4, h1, mean, rstd, gamma, dg, db, W0, cur, dqs, Wqs, fb))
    dq_all, Wqc, dxr, gq = r(3, B, Nq, d).bfloat16(), [r(d, d, sc=0.06) for _ in range(3)], r(B, Nq, d), torch.empty(B, Nq, d, device=dev)
    t_chb0 = _time(lambda: ops.chain_mh_bwd(dc, cf, W4, h1, mean, rstd, gamma, dg, db, W0, None, dqs, Wqs, fb, prev=(dq_all, Wqc, dxr, gq)))
    print(f"R = {B * Nq}: mask head forward: five launches {t_sep:.2f} us, chain {t_ch:.2f} us; backward: six launches {t_sepb:.2f} us, "
          f"chain {t_chb:.2f} us, chain incl. the cross-attention query gradient {t_chb0:.2f} us; hand-off timeouts: {ops.chain_error(dev)}")
