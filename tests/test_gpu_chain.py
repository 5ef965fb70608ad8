"""GPU: the one-launch row-local chain (csrc/chain_ffn.hip: self-attention out-projection + LayerNorm + FFN + LayerNorm, a
group of 8 workgroups per 32-row tile handing rows over inside one XCD) against the five launches it replaces -- every
output bit for bit, at the decoder's shapes, repeatedly (the hand-off words carry over from launch to launch)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _five_launches(o_s, Wo, bo, x1s, g1, be1, eps1, W1, b1, W2, b2, g2, be2, eps2):
    from pq3d_amd import _lib as L, fused, ops
    cq = L.BF16X3
    d = o_s.shape[-1]
    R, F_ = o_s.numel() // d, W1.shape[0]
    dev = o_s.device
    f = torch.empty_like(o_s)
    L.gemm(M=R, N=d, K=d, A=[o_s], B=[Wo], bias=[bo], Cs=[f], ct=cq, lda=d, ldb=d, ldc=d)
    x2, mean1, rstd1 = fused._ln_fwd(x1s, [f], [g1], [be1], eps1, None, o_s.shape[-2])
    h = torch.empty(*o_s.shape[:-1], F_, dtype=torch.float32, device=dev)
    L.gemm(M=R, N=F_, K=d, A=[x2], B=[W1], bias=[b1], Cs=[h], ct=cq, lda=d, ldb=d, ldc=F_, act="relu")
    zp = torch.empty(4, *o_s.shape, dtype=torch.float32, device=dev)
    Fk = F_ // 4
    hv = h.view(R, F_)
    L.gemm(M=R, N=d, K=Fk, A=[hv[:, k * Fk:(k + 1) * Fk] for k in range(4)], B=[W2[:, k * Fk:(k + 1) * Fk] for k in range(4)],
           bias=[b2, None, None, None], Cs=[zp[k] for k in range(4)], ct=cq, lda=F_, ldb=F_, ldc=d)
    z = torch.empty_like(o_s)
    x3, mean2, rstd2 = fused._ln_fwd(x2, [zp[k] for k in range(4)], [g2], [be2], eps2, None, o_s.shape[-2], sum_branches=True, osum=z)
    return f, x2, mean1, rstd1, h, zp, z, x3, mean2, rstd2


@pytest.mark.parametrize("B,Nq,F_", [(8, 100, 2048), (4, 200, 2048), (3, 37, 2048), (1, 1, 2048), (10, 100, 2048), (16, 100, 2048), (9, 200, 2048), (1, 2048, 2048), (1, 1025, 2048)])
def test_chain_equals_five_launches(B, Nq, F_):
    from pq3d_amd import ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(B * 1000 + Nq)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    d = 256
    args = (r(B, Nq, d), r(d, d, sc=0.06), r(d, sc=0.1), r(B, Nq, d), 1 + r(d, sc=0.1), r(d, sc=0.1), 1e-5,
            r(F_, d, sc=0.06), r(F_, sc=0.1), r(d, F_, sc=0.03), r(d, sc=0.1), 1 + r(d, sc=0.1), r(d, sc=0.1), 1e-5)
    ref = _five_launches(*args)
    flags = ops.chain_flags(B * Nq, dev)
    names = ("f", "x2", "mean1", "rstd1", "h", "zp", "z", "x3", "mean2", "rstd2")
    for rep in range(4):   # the flags of launch n are the starting state of launch n + 1
        out = ops.chain_ffn_fwd(*args, flags)
        torch.cuda.synchronize()
        assert not ops.chain_error(dev)
        for n, a, b in zip(names, out, ref):
            b = b[:1] if n in ("mean2", "rstd2") else b   # the five-launch path allocates one statistics row per partial sum
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (n, rep, (a - b).abs().max().item())


@pytest.mark.parametrize("B,Nq,bias,pad", [(8, 100, True, True), (4, 200, True, True), (16, 100, True, False), (3, 77, False, True),
                                           (50, 16, True, True), (80, 10, True, True), (2, 240, True, True), (1, 1, False, False),
                                           (5, 33, True, False), (8, 128, False, False)])
def test_chain_with_the_self_attention_core_inside(B, Nq, bias, pad):
    """Step 0 of chain_ffn_fwd (round 6): the split-bf16 self-attention core itself, member j = head j over the tile's rows -- o_s,
    lse and everything downstream bit for bit what pq3d_attn_fwd + the chain write in two launches; tiles that straddle 2 .. 4 scenes
    (N_q = 100, 77, 16, 10), two row tiles per group (R = 1600), one scene's planes at a time (N_q = 240), repeated launches."""
    from pq3d_amd import _lib as L, fused, ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(B * 977 + Nq)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    d, H, F_ = 256, 8, 2048
    assert ops.chain_sa_ok(Nq, H, d)
    qkv = r(3, B, Nq, d)
    sb = r(B, H, Nq, Nq) if bias else None
    kpm = None
    if pad:
        kpm = (torch.rand(B, Nq, generator=g) < 0.2).to(dev)
        kpm[:, 0] = False
        if B > 1:
            kpm[1] = True      # a scene with every key padded: zero output, lse = -inf
    args = (r(d, d, sc=0.06), r(d, sc=0.1), r(B, Nq, d), 1 + r(d, sc=0.1), r(d, sc=0.1), 1e-5,
            r(F_, d, sc=0.06), r(F_, sc=0.1), r(d, F_, sc=0.03), r(d, sc=0.1), 1 + r(d, sc=0.1), r(d, sc=0.1), 1e-5)
    nextq = (r(B, Nq, d), [r(d, d, sc=0.06) for _ in range(3)], [r(d, sc=0.1) for _ in range(3)])
    scale = 1.0 / (32 ** 0.5)
    o_ref = torch.empty(B, Nq, d, device=dev)
    lse_ref = torch.empty(B, H, Nq, device=dev)
    fused._attn(qkv[0], qkv[1], qkv[2], o_ref, lse_ref, H, L.BF16X3, False, kpm=kpm, bias=sb)
    flags = ops.chain_flags(max(B * Nq, 2048), dev)
    ref = ops.chain_ffn_fwd(o_ref, *args, flags, nextq=nextq, q_dtype=torch.float32)
    torch.cuda.synchronize()
    for rep in range(3):
        o_s = torch.full((B, Nq, d), float("nan"), device=dev)
        lse = torch.full((B, H, Nq), float("nan"), device=dev)
        out = ops.chain_ffn_fwd(o_s, *args, flags, nextq=nextq, q_dtype=torch.float32, sa=(qkv[0], qkv[1], qkv[2], sb, kpm, lse, scale))
        torch.cuda.synchronize()
        assert not ops.chain_error(dev)
        assert torch.equal(o_s.view(torch.int32), o_ref.view(torch.int32)), (rep, (o_s - o_ref).abs().max().item())
        assert torch.equal(lse.view(torch.int32), lse_ref.view(torch.int32)), rep
        for i, (a, b) in enumerate(zip(out, ref)):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (i, rep)


def test_chain_refuses_what_it_cannot_hold():
    from pq3d_amd import _lib as L, ops
    assert not ops.chain_ffn_ok(2049, 256, 2048) and not ops.chain_ffn_ok(800, 512, 2048) and ops.chain_ffn_ok(800, 256, 2048) and not ops.chain_ffn_ok(800, 256, 1024)
    c = L.ChainFfnDesc()
    c.R, c.d, c.F = 4000, 256, 2048
    assert L.lib().pq3d_chain_ffn_fwd(L.C.byref(c), None) == -1


def _three_launches(o_all, Wos, bos, x, gammas, betas, eps, coef, rps, qpos, Wqkv, bqkv):
    from pq3d_amd import _lib as L, fused
    M, d = o_all.shape[0], o_all.shape[-1]
    R = x.numel() // d
    dev = x.device
    op_all = torch.empty(M, *x.shape, dtype=torch.float32, device=dev)
    L.gemm(M=R, N=d, K=d, A=[o_all[m] for m in range(M)], B=list(Wos), bias=list(bos), Cs=[op_all[m] for m in range(M)], ct=L.BF16,
           lda=d, ldb=d, ldc=d)
    x1, mean, rstd = fused._ln_fwd(x, [op_all[m] for m in range(M)], list(gammas), list(betas), eps, coef, rps)
    qkv = torch.empty(3, *x.shape, dtype=torch.float32, device=dev)
    L.gemm(M=R, N=d, K=d, A=[x1] * 3, A2=[qpos, qpos, None], B=list(Wqkv), bias=list(bqkv), Cs=[qkv[0], qkv[1], qkv[2]], ct=L.BF16X3,
           lda=d, ldb=d, ldc=d)
    return op_all, x1, mean, rstd, qkv


@pytest.mark.parametrize("B,Nq,M,with_coef", [(8, 100, 3, False), (8, 100, 3, True), (4, 200, 3, False), (16, 100, 3, True), (3, 37, 2, False),
                                               (1, 1, 1, False), (1, 2048, 3, False)])
def test_chain_ca_equals_three_launches(B, Nq, M, with_coef):
    from pq3d_amd import ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(B * 1000 + Nq + M)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    d = 256
    coef = None
    if with_coef:
        coef = torch.rand(M, B, generator=g).to(dev)
        coef = (coef / coef.sum(0, keepdim=True)).contiguous()
    args = (r(M, B, Nq, d).bfloat16(), [r(d, d, sc=0.06) for _ in range(M)], [r(d, sc=0.1) for _ in range(M)], r(B, Nq, d),
            [1 + r(d, sc=0.1) for _ in range(M)], [r(d, sc=0.1) for _ in range(M)], 1e-5, coef, Nq, r(B, Nq, d),
            [r(d, d, sc=0.06) for _ in range(3)], [r(d, sc=0.1) for _ in range(3)])
    ref = _three_launches(*args)
    flags = ops.chain_flags(B * Nq, dev)
    for rep in range(3):
        out = ops.chain_ca_fwd(*args, flags)
        torch.cuda.synchronize()
        assert not ops.chain_error(dev)
        for n, a, b in zip(("op", "x1", "mean", "rstd", "qkv"), out, ref):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (n, rep, (a - b).abs().max().item())


def test_chain_ffn_with_next_query_projection():
    """Sixth step of the FFN chain: the next layer's cross-attention queries, bf16, equal to the grouped launch's."""
    from pq3d_amd import _lib as L, ops
    dev = torch.device("cuda")
    for B, Nq, M in ((8, 100, 3), (16, 100, 3), (2, 50, 1)):
        g = torch.Generator().manual_seed(7 + B)
        r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
        d, F_ = 256, 2048
        args = (r(B, Nq, d), r(d, d, sc=0.06), r(d, sc=0.1), r(B, Nq, d), 1 + r(d, sc=0.1), r(d, sc=0.1), 1e-5,
                r(F_, d, sc=0.06), r(F_, sc=0.1), r(d, F_, sc=0.03), r(d, sc=0.1), 1 + r(d, sc=0.1), r(d, sc=0.1), 1e-5)
        qpos, Wq, bq = r(B, Nq, d), [r(d, d, sc=0.06) for _ in range(M)], [r(d, sc=0.1) for _ in range(M)]
        ref = _five_launches(*args)
        x3 = ref[7]
        qref = torch.empty(M, B, Nq, d, dtype=torch.bfloat16, device=dev)
        L.gemm(M=B * Nq, N=d, K=d, A=[x3] * M, A2=[qpos] * M, B=Wq, bias=bq, Cs=[qref[m] for m in range(M)], ct=L.BF16X3, lda=d, ldb=d, ldc=d)
        flags = ops.chain_flags(B * Nq, dev)
        for rep in range(3):
            out = ops.chain_ffn_fwd(*args, flags, nextq=(qpos, Wq, bq))
            torch.cuda.synchronize()
            assert not ops.chain_error(dev)
            assert torch.equal(out[7].view(torch.int32), x3.view(torch.int32))
            assert torch.equal(out[10].view(torch.int16), qref.view(torch.int16)), (B, rep)


@pytest.mark.parametrize("B,Nq", [(8, 100), (16, 100), (3, 37), (1, 1)])
def test_chain_ffn_bwd_against_separate_launches(B, Nq):
    """Backward chain: g2 and dhp bit for bit; g1 and the LayerNorm parameter gradients to fp32 summation-order accuracy (the
    separate launches add linear1's four k slices with atomics, the chain in a fixed order)."""
    from pq3d_amd import _lib as L, fused, ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(11 + B)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    d, F_ = 256, 2048
    R = B * Nq
    fwd = (r(B, Nq, d), r(d, d, sc=0.06), r(d, sc=0.1), r(B, Nq, d), 1 + r(d, sc=0.1), r(d, sc=0.1), 1e-5,
           r(F_, d, sc=0.06), r(F_, sc=0.1), r(d, F_, sc=0.03), r(d, sc=0.1), 1 + r(d, sc=0.1), r(d, sc=0.1), 1e-5)
    f, x2, mean1, rstd1, h, zp, z, x3, mean2, rstd2 = _five_launches(*fwd)
    x1s, g1, W1, W2, g2 = fwd[3], fwd[4], fwd[7], fwd[9], fwd[11]
    dx = r(B, Nq, d)
    # separate launches (what _DecoderBackward.ffn / .self_attn issue)
    dg2r, db2r, dg1r, db1r = (torch.zeros(d, device=dev) for _ in range(4))
    dx2r, dyl = fused._ln_bwd(x2, [z], [g2], [fwd[12]], 1e-5, None, Nq, mean2[:1], rstd2[:1], dx, [dg2r], [db2r])
    dy_ref = dyl[0]
    dhp_ref = torch.empty(B, Nq, F_, dtype=torch.bfloat16, device=dev)
    L.gemm(M=R, N=F_, K=d, A=[dy_ref], B=[W2], Cs=[dhp_ref], aux=[h], act_grad="relu", ct=L.BF16, lda=d, ldb=F_, ldc=F_, transB=True)
    dx2 = dx2r.clone()
    L.gemm(M=R, N=d, K=F_, A=[dhp_ref], B=[W1], Cs=[dx2], ct=L.BF16, lda=F_, ldb=d, ldc=d, transB=True, splitk=4, accumulate=True)
    dx1r, dfl = fused._ln_bwd(x1s, [f], [g1], [fwd[5]], 1e-5, None, Nq, mean1, rstd1, dx2, [dg1r], [db1r])
    flags = ops.chain_flags(R, dev)
    for rep in range(3):
        dg2, db2, dg1, db1 = (torch.zeros(d, device=dev) for _ in range(4))
        dy, dhp, df = ops.chain_ffn_bwd(dx, x2, z, g2, mean2[:1].contiguous(), rstd2[:1].contiguous(), dg2, db2, W2, h, W1, x1s, f, g1,
                                        mean1, rstd1, dg1, db1, flags)
        torch.cuda.synchronize()
        assert not ops.chain_error(dev)
        assert torch.equal(dy.view(torch.int32), dy_ref.view(torch.int32))
        assert torch.equal(dy.view(torch.int32), dx2r.view(torch.int32))
        assert torch.equal(dhp.view(torch.int16), dhp_ref.view(torch.int16))
        sc = dfl[0].abs().max().item()
        assert (df - dfl[0]).abs().max().item() <= 2e-5 * sc and (df - dx1r).abs().max().item() <= 2e-5 * sc
        for a, b in ((dg2, dg2r), (db2, db2r), (dg1, dg1r), (db1, db1r)):
            assert (a - b).abs().max().item() <= 1e-4 * max(b.abs().max().item(), 1.0), rep


@pytest.mark.parametrize("B,Nq,M,with_coef", [(8, 100, 3, False), (8, 100, 3, True), (16, 100, 3, False), (3, 37, 2, False), (1, 1, 1, False)])
def test_chain_sa_bwd_against_separate_launches(B, Nq, M, with_coef):
    from pq3d_amd import _lib as L, fused, ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(B * 100 + Nq + M)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    d, R = 256, B * Nq
    coef = None
    if with_coef:
        coef = torch.rand(M, B, generator=g).to(dev)
        coef = (coef / coef.sum(0, keepdim=True)).contiguous()
    x, qpos = r(B, Nq, d), r(B, Nq, d)
    o_all = r(M, B, Nq, d).bfloat16()
    Wos, bos = [r(d, d, sc=0.06) for _ in range(M)], [r(d, sc=0.1) for _ in range(M)]
    gam, bet = [1 + r(d, sc=0.1) for _ in range(M)], [r(d, sc=0.1) for _ in range(M)]
    Wl, bl = [r(d, d, sc=0.06) for _ in range(3)], [r(d, sc=0.1) for _ in range(3)]
    op_all, x1, mean, rstd, qkv = _three_launches(o_all, Wos, bos, x, gam, bet, 1e-5, coef, Nq, qpos, Wl, bl)
    dqkv, dx1r = r(3, B, Nq, d), r(B, Nq, d)
    # separate launches
    g3r = torch.empty(3, B, Nq, d, device=dev)
    L.gemm(M=R, N=d, K=d, A=[dqkv[0], dqkv[1], dqkv[2]], B=list(Wl), Cs=[g3r[0], g3r[1], g3r[2]], aux=[None, None, dx1r], act_grad="add",
           ct=L.BF16, lda=d, ldb=d, ldc=d, transB=True)
    dgr, dbr = [torch.zeros(d, device=dev) for _ in range(M)], [torch.zeros(d, device=dev) for _ in range(M)]
    dxz = torch.zeros(B, Nq, d, device=dev)
    dxr_ref, dop_ref = fused._ln_bwd(x, [op_all[m] for m in range(M)], gam, bet, 1e-5, coef, Nq, mean, rstd, [g3r[0], g3r[1], g3r[2]],
                                     dgr, dbr, dx_zeroed=dxz if M > 1 else None)
    do_ref = torch.empty(M, B, Nq, d, dtype=torch.bfloat16, device=dev)
    L.gemm(M=R, N=d, K=d, A=[dop_ref[m] for m in range(M)], B=list(Wos), Cs=[do_ref[m] for m in range(M)], ct=L.BF16, lda=d, ldb=d, ldc=d,
           transB=True)
    flags = ops.chain_flags(R, dev)
    exact = M == 3 and R >= 512   # the merged kernel (one row order); otherwise the reference sums dx with atomics
    for rep in range(3):
        dg, db = [torch.zeros(d, device=dev) for _ in range(M)], [torch.zeros(d, device=dev) for _ in range(M)]
        g3, dop, dxr, do_all = ops.chain_sa_bwd(dqkv, Wl, dx1r, x, op_all, gam, mean, rstd, coef, Nq, dg, db, Wos, flags)
        torch.cuda.synchronize()
        assert not ops.chain_error(dev)
        assert torch.equal(g3.view(torch.int32), g3r.view(torch.int32))
        assert torch.equal(dop.view(torch.int32), dop_ref.view(torch.int32))
        assert torch.equal(do_all.view(torch.int16), do_ref.view(torch.int16))
        if exact:
            assert torch.equal(dxr.view(torch.int32), dxr_ref.view(torch.int32))
        else:
            assert (dxr - dxr_ref).abs().max().item() <= 1e-5 * max(dxr_ref.abs().max().item(), 1e-6)
        for m in range(M):
            assert (dg[m] - dgr[m]).abs().max().item() <= 1e-4 * max(dgr[m].abs().max().item(), 1.0)
            assert (db[m] - dbr[m]).abs().max().item() <= 1e-4 * max(dbr[m].abs().max().item(), 1.0)


def test_chain_ffn_bwd_forms_its_upstream_gradient():
    """Step 0 of the backward chain: dx = sum_m dq_m Wq_m + dxr and the sum without dxr, the grouped launch's bits; the rest of
    the chain on that dx as before."""
    from pq3d_amd import _lib as L, fused, ops
    dev = torch.device("cuda")
    for B, Nq, M in ((8, 100, 3), (16, 100, 3), (2, 33, 1)):
        g = torch.Generator().manual_seed(5 + B)
        r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
        d, F_, R = 256, 2048, B * Nq
        fwd = (r(B, Nq, d), r(d, d, sc=0.06), r(d, sc=0.1), r(B, Nq, d), 1 + r(d, sc=0.1), r(d, sc=0.1), 1e-5,
               r(F_, d, sc=0.06), r(F_, sc=0.1), r(d, F_, sc=0.03), r(d, sc=0.1), 1 + r(d, sc=0.1), r(d, sc=0.1), 1e-5)
        f, x2, mean1, rstd1, h, zp, z, x3, mean2, rstd2 = _five_launches(*fwd)
        x1s, g1, W1, W2, g2 = fwd[3], fwd[4], fwd[7], fwd[9], fwd[11]
        dq_all, Wq, dxr = r(M, B, Nq, d).bfloat16(), [r(d, d, sc=0.06) for _ in range(M)], r(B, Nq, d)
        gq_ref, dx_ref = torch.empty(B, Nq, d, device=dev), torch.empty(B, Nq, d, device=dev)
        L.gemm(M=R, N=d, K=d, A=[dq_all[m] for m in range(M)], B=Wq, Cs=[dx_ref] + [None] * (M - 1), C2=[gq_ref] + [None] * (M - 1),
               aux=[dxr] + [None] * (M - 1), act_grad="add", ct=L.BF16, lda=d, ldb=d, ldc=d, transB=True, kconcat=M)
        m2, r2 = mean2[:1].contiguous(), rstd2[:1].contiguous()
        z4 = lambda: [torch.zeros(d, device=dev) for _ in range(4)]
        flags = ops.chain_flags(R, dev)
        a0 = z4()
        dy0, dhp0, df0 = ops.chain_ffn_bwd(dx_ref, x2, z, g2, m2, r2, a0[0], a0[1], W2, h, W1, x1s, f, g1, mean1, rstd1, a0[2], a0[3], flags)
        for rep in range(3):
            a1 = z4()
            gq = torch.empty(B, Nq, d, device=dev)
            dy, dhp, df, dxo = ops.chain_ffn_bwd(None, x2, z, g2, m2, r2, a1[0], a1[1], W2, h, W1, x1s, f, g1, mean1, rstd1, a1[2], a1[3],
                                                 flags, prev=(dq_all, Wq, dxr, gq))
            torch.cuda.synchronize()
            assert not ops.chain_error(dev)
            assert torch.equal(gq.view(torch.int32), gq_ref.view(torch.int32))
            assert torch.equal(dxo.view(torch.int32), dx_ref.view(torch.int32))
            assert torch.equal(dy.view(torch.int32), dy0.view(torch.int32)) and torch.equal(dhp.view(torch.int16), dhp0.view(torch.int16))
            assert torch.equal(df.view(torch.int32), df0.view(torch.int32))


def _mh_five_launches(x, W0, b0, gamma, beta, eps, W4, b4, cols, Wqs, bqs):
    """The mask head's row-local part as fused._mh_forward launches it without the chain."""
    from pq3d_amd import _lib as L, fused
    d = x.shape[-1]
    R, C_, Mm = x.numel() // d, W4.shape[0], len(Wqs)
    dev = x.device
    h1 = torch.empty_like(x)
    L.gemm(M=R, N=d, K=d, A=[x], B=[W0], bias=[b0], Cs=[h1], ct=L.BF16X3, lda=d, ldb=d, ldc=d, act="relu")
    h2, mean, rstd = fused._ln_fwd(None, [h1], [gamma], [beta], eps, None, x.shape[-2])
    cls_raw = torch.empty(*x.shape[:-1], C_, dtype=torch.float32, device=dev)
    L.gemm(M=R, N=C_, K=d, A=[h2], B=[W4], bias=[b4], Cs=[cls_raw], ct=L.BF16X3, lda=d, ldb=d, ldc=C_)
    cls = cls_raw
    if cols is not None:
        cls = torch.empty_like(cls_raw)
        L.check(L.lib().pq3d_fill_cols(L.ptr(cls_raw), L.ptr(cls), R, C_, L.ptr(cols), cols.numel(), float("-inf"), L.stream()), "fill")
    qm = torch.empty(Mm, *x.shape, dtype=torch.float32, device=dev)
    if Mm:
        L.gemm(M=R, N=d, K=d, A=[x] * Mm, B=list(Wqs), bias=list(bqs), Cs=[qm[m] for m in range(Mm)], ct=L.BF16X3, lda=d, ldb=d, ldc=d)
    return h1, h2, mean, rstd, cls, qm


@pytest.mark.gpu
@pytest.mark.parametrize("B,Nq,C_,Mm,fill", [(4, 200, 201, 3, True), (8, 100, 201, 3, False), (3, 37, 19, 1, True), (1, 1, 1, 0, False),
                                             (1, 2048, 256, 3, True), (9, 200, 607 - 400, 2, True), (16, 100, 32, 3, False)])
def test_chain_mh_equals_five_launches(B, Nq, C_, Mm, fill):
    from pq3d_amd import ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(B * 1000 + Nq + C_)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    d = 256
    cols = flags_c = None
    if fill:
        pick = sorted({0, C_ - 1, C_ // 2})
        cols = torch.tensor(pick, dtype=torch.int32, device=dev)
        flags_c = torch.zeros(C_, dtype=torch.int32, device=dev)
        flags_c[cols.long()] = 1
    x, W0, b0, gamma, beta = r(B, Nq, d), r(d, d, sc=0.06), r(d, sc=0.1), 1 + r(d, sc=0.1), r(d, sc=0.1)
    W4, b4 = r(C_, d, sc=0.06), r(C_, sc=0.1)
    Wqs, bqs = [r(d, d, sc=0.06) for _ in range(Mm)], [r(d, sc=0.1) for _ in range(Mm)]
    ref = _mh_five_launches(x, W0, b0, gamma, beta, 1e-5, W4, b4, cols, Wqs, bqs)
    flags = ops.chain_flags(B * Nq, dev)
    for rep in range(3):
        out = ops.chain_mh_fwd(x, W0, b0, gamma, beta, 1e-5, W4, b4, flags_c, float("-inf"), Wqs, bqs, flags)
        torch.cuda.synchronize()
        assert not ops.chain_error(dev)
        for n, a, b in zip(("h1", "h2", "mean", "rstd", "cls", "qm"), out, ref):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (n, rep)


def _mh_bwd_six_launches(dc, cols, W4, h1, mean, rstd, gamma, dg, db, W0, cur, dqs, Wqs):
    """The mask head's row-local backward as fused._DecoderBackward.mask_head launches it without the chain."""
    from pq3d_amd import _lib as L, fused, ops
    d = h1.shape[-1]
    R, C_, Mm = h1.numel() // d, W4.shape[0], len(dqs)
    dev = h1.device
    dcl = dc
    if cols is not None:
        dcl = torch.empty_like(dc)
        L.check(L.lib().pq3d_fill_cols(L.ptr(dc), L.ptr(dcl), R, C_, L.ptr(cols), cols.numel(), 0.0, L.stream()), "fill")
    dh2 = torch.empty_like(h1)
    L.gemm(M=R, N=d, K=C_, A=[dcl], B=[W4], Cs=[dh2], ct=L.BF16, lda=C_, ldb=d, ldc=d, transB=True)
    _, dh1 = fused._ln_bwd(None, [h1], [gamma], [torch.zeros_like(gamma)], 1e-5, None, h1.shape[-2], mean, rstd, dh2, [dg], [db])
    dpre = ops.act_bwd(dh1[0], h1, "relu", torch.bfloat16)
    nxt = torch.empty_like(h1)
    L.gemm(M=R, N=d, K=d, A=[dpre], B=[W0], Cs=[nxt], aux=[cur], act_grad="add", ct=L.BF16, lda=d, ldb=d, ldc=d, transB=True)
    out = nxt
    if Mm:
        out = torch.empty_like(h1)
        L.gemm(M=R, N=d, K=d, A=list(dqs), B=list(Wqs), Cs=[out] + [None] * (Mm - 1), aux=[nxt] + [None] * (Mm - 1), act_grad="add",
               ct=L.BF16, lda=d, ldb=d, ldc=d, transB=True, kconcat=Mm)
    return dcl, dpre, out


@pytest.mark.parametrize("B,Nq,C_,Mm,fill,f32", [(4, 200, 201, 3, True, True), (8, 100, 201, 3, False, False), (3, 37, 19, 1, True, True),
                                                 (1, 1, 1, 0, False, True), (1, 2048, 256, 3, True, False), (9, 200, 207, 2, True, True),
                                                 (16, 100, 32, 3, False, True)])
def test_chain_mh_bwd_against_separate_launches(B, Nq, C_, Mm, fill, f32):
    from pq3d_amd import ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(B * 1000 + Nq + C_ + 7)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    d = 256
    cols = flags_c = None
    if fill:
        pick = sorted({0, C_ - 1, C_ // 2})
        cols = torch.tensor(pick, dtype=torch.int32, device=dev)
        flags_c = torch.zeros(C_, dtype=torch.int32, device=dev)
        flags_c[cols.long()] = 1
    h1 = torch.relu(r(B, Nq, d))
    mean, rstd = h1.mean(-1).reshape(1, -1).contiguous(), (1.0 / (h1.var(-1, unbiased=False) + 1e-5).sqrt()).reshape(1, -1).contiguous()
    dc, W4, gamma, W0, cur = r(B, Nq, C_), r(C_, d, sc=0.06), 1 + r(d, sc=0.1), r(d, d, sc=0.06), r(B, Nq, d)
    dqs = [r(B, Nq, d) if f32 else r(B, Nq, d).bfloat16() for _ in range(Mm)]
    Wqs = [r(d, d, sc=0.06) for _ in range(Mm)]
    dg0, db0 = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
    ref = _mh_bwd_six_launches(dc, cols, W4, h1, mean, rstd, gamma, dg0, db0, W0, cur, dqs, Wqs)
    flags = ops.chain_flags(B * Nq, dev)
    for rep in range(3):
        dg, db = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
        out = ops.chain_mh_bwd(dc, flags_c, W4, h1, mean, rstd, gamma, dg, db, W0, cur, dqs, Wqs, flags)
        torch.cuda.synchronize()
        assert not ops.chain_error(dev)
        for n, a, b in zip(("dcl", "dpre", "out"), out, ref):
            assert torch.equal(a.view(torch.int16 if a.dtype == torch.bfloat16 else torch.int32),
                               b.view(torch.int16 if b.dtype == torch.bfloat16 else torch.int32)), (n, rep, (a.float() - b.float()).abs().max().item())
        # parameter gradients: the same terms in another summation order
        for n, a, b in (("dgamma", dg, dg0), ("dbeta", db, db0)):
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-4 * max(1.0, float(b.abs().max()))), (n, (a - b).abs().max().item())


@pytest.mark.parametrize("B,Nq,M", [(4, 200, 3), (3, 37, 2), (16, 100, 1)])
def test_chain_mh_bwd_forms_its_upstream_gradient(B, Nq, M):
    """prev = (dq_all, Wq, dxr, gq): cur = sum_m dq_m Wq_m + dxr formed inside the launch, bit for bit pq3d_gemm's (kconcat, C2)."""
    from pq3d_amd import _lib as L, ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(B * 77 + Nq + M)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    d, C_, Mm = 256, 201, 3
    R = B * Nq
    flags_c = torch.zeros(C_, dtype=torch.int32, device=dev)
    flags_c[5] = 1
    h1 = torch.relu(r(B, Nq, d))
    mean, rstd = h1.mean(-1).reshape(1, -1).contiguous(), (1.0 / (h1.var(-1, unbiased=False) + 1e-5).sqrt()).reshape(1, -1).contiguous()
    dc, W4, gamma, W0 = r(B, Nq, C_), r(C_, d, sc=0.06), 1 + r(d, sc=0.1), r(d, d, sc=0.06)
    dqs, Wqs = [r(B, Nq, d) for _ in range(Mm)], [r(d, d, sc=0.06) for _ in range(Mm)]
    dq_all, Wqc, dxr = r(M, B, Nq, d).bfloat16(), [r(d, d, sc=0.06) for _ in range(M)], r(B, Nq, d)
    cur, gq0 = torch.empty_like(dxr), torch.empty_like(dxr)
    L.gemm(M=R, N=d, K=d, A=[dq_all[m] for m in range(M)], B=Wqc, Cs=[cur] + [None] * (M - 1), C2=[gq0] + [None] * (M - 1),
           aux=[dxr] + [None] * (M - 1), act_grad="add", ct=L.BF16, lda=d, ldb=d, ldc=d, transB=True, kconcat=M)
    flags = ops.chain_flags(R, dev)
    z = lambda: torch.zeros(d, device=dev)
    ref = ops.chain_mh_bwd(dc, flags_c, W4, h1, mean, rstd, gamma, z(), z(), W0, cur, dqs, Wqs, flags)
    for rep in range(2):
        gq = torch.empty_like(dxr)
        out = ops.chain_mh_bwd(dc, flags_c, W4, h1, mean, rstd, gamma, z(), z(), W0, None, dqs, Wqs, flags, prev=(dq_all, Wqc, dxr, gq))
        torch.cuda.synchronize()
        assert not ops.chain_error(dev)
        assert torch.equal(gq.view(torch.int32), gq0.view(torch.int32))
        assert torch.equal(out[2].view(torch.int32), ref[2].view(torch.int32)), (out[2] - ref[2]).abs().max().item()
        assert torch.equal(out[1].view(torch.int16), ref[1].view(torch.int16))


@pytest.mark.parametrize("case", ["mask_head", "plain"])
def test_model_step_is_the_same_with_and_without_chains(case):
    """End to end: one forward + backward of the whole model in 'bf16' mode with the chain launches on and off (fused.set_chain)
    -- every output tensor bit for bit (the chains' arithmetic is the separate launches'), every parameter gradient to the
    summation order of the reductions the backward chains do in a fixed order instead of with atomics (see below)."""
    from tests import util
    from pq3d_amd import fused, ops
    from pq3d_amd.modules import set_compute
    dev = torch.device("cuda")
    if case == "mask_head":   # config 4's structure at a reduced segment count (mask head in front of every layer, 3-D self-masks)
        args = dict(B=4, Ns=512, Nq=200, d=256, H=8, L=2, memories=["voxel", "mv", "pc"], heads=["mask"], spatial=True,
                    structure="parallel", use_self_mask=True, C=201, foc=(0, 2), seed=0, data_seed=1234)
    else:                     # config 2's structure
        args = dict(B=8, Ns=256, Nq=100, d=256, H=8, L=2, memories=["voxel", "mv", "pc"], heads=[], spatial=True,
                    structure="parallel", seed=0, data_seed=1234)
    _cfg, model, _sd, dd = util.model_case(args)
    set_compute(model, "bf16")
    model.unified_encoder.fused = True
    model.to(dev)
    ddv = {k: v.to(dev) for k, v in dd.items()}
    res = {}
    try:
        for on in (True, False):
            fused.set_chain(on)
            model.zero_grad()
            out = model(dict(ddv))
            util.synthetic_loss(out, args["heads"], out["query_embeds"]).backward()
            torch.cuda.synchronize()
            assert not ops.chain_error(dev)
            outs = [out["query_embeds"]] + list(out.get("predictions_mask", [])) + list(out.get("predictions_class", []))
            res[on] = ([t.detach().clone() for t in outs], {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
    finally:
        fused.set_chain(True)
    for a, b in zip(res[True][0], res[False][0]):
        assert torch.equal(a, b)
    assert sorted(res[True][1]) == sorted(res[False][1])
    # Gradients: the backward's products take bf16 operands, so a last-bit difference of an fp32 gradient tensor (summation order:
    # the unchained path's split-K atomics are not even the same from run to run, 1-2e-3 on the far parameters) flips operand
    # roundings further down and grows stage by stage towards one bf16 ulp (tools/probes/chain_e2e_diff.py: 2e-7 at the last
    # layer's norms -> 2e-5 -> 4e-4 -> 3e-3 at the input encoders; chains on vs on: 3e-7 everywhere, chains off vs off: 2e-3).
    # So: tight where the difference is born (last layer's FFN / self-attention norm), bf16-rounding level everywhere else.
    gmax = max(float(v.norm()) for v in res[False][1].values())
    last = f"unified_encoder.unified_encoder.{args['L'] - 1}."
    for n, g0 in res[False][1].items():
        err = float((res[True][1][n] - g0).norm() / max(float(g0.norm()), 1e-3 * gmax))
        tight = n.startswith(last + "ffn.") or n.startswith(last + "self_attn.norm")
        assert err < (2e-5 if tight else 1e-2), (n, err)


# ---------------------------------------------------------------------------------------------------- device gate, neighbours, polling
def test_chain_device_gate_measures_the_placement_rule():
    """chains are valid only where workgroup id % 8 is the XCD and all members are co-resident.  The gate checks
    gfx950 / 256 CUs / 160 KB LDS and MEASURES XCC_ID == (id + c) % 8 over 256 workgroups (pq3d_chain_device_ok); on this box it must say
    yes, and fused._chain_on follows it (and fused.set_chain)."""
    from pq3d_amd import _lib as L, fused, ops
    dev = torch.device("cuda", torch.cuda.current_device())
    assert L.lib().pq3d_chain_device_ok(1, L.stream()) == 1
    assert ops.chain_device_ok(dev) and fused._chain_on(dev)
    try:
        fused.set_chain(False)
        assert not fused._chain_on(dev)
        ops._CHAIN_DEV_OK[dev.index] = False      # a device the gate refused: the chains stay off whatever the switch says
        fused.set_chain(True)
        assert not fused._chain_on(dev)
    finally:
        ops._CHAIN_DEV_OK.pop(dev.index, None)
        fused.set_chain(True)
    assert fused._chain_on(dev)


@pytest.mark.parametrize("B", [8, 16], ids=["R800", "R1600"])
@pytest.mark.parametrize("held,lds_kb,us", [(32, 100, 150000), (96, 100, 3000), (64, 8, 150000)],
                         ids=["32CUs-held", "96CUs-held-briefly", "64-small-neighbours"])
@pytest.mark.parametrize("case", ["plain", "mask_head"])
def test_chain_launches_beside_a_cu_holding_kernel(case, B, held, lds_kb, us):
    """the data-parallel step reduces gradient buckets UNDER the decoder backward, i.e. an RCCL kernel (one
    workgroup per channel) holds CUs beside chain_ffn_bwd / chain_sa_bwd.  Stand-in: `held` workgroups that each pin `lds_kb` of a
    CU's LDS for `us` microseconds (150 ms: longer than an eager step) on a second stream (pq3d_test_occupy_cus; 100 KB = no chain workgroup fits beside one) while a
    whole model step (every chain kernel: ffn / ca forward, ffn / sa backward; mask head pair) runs on the main stream at R = 800
    and R = 1600 query rows.  32 held CUs is the head-room chain_nrt() leaves; 96 makes members WAIT for a CU (bounded polls).
    Required: forward outputs bit for bit the undisturbed run's, gradients at the chained backward's run-to-run level, error word
    clear."""
    from tests import util
    from pq3d_amd import _lib as L, fused, ops
    from pq3d_amd.modules import set_compute
    dev = torch.device("cuda")
    if case == "mask_head":
        args = dict(B=B // 2, Ns=512, Nq=200, d=256, H=8, L=2, memories=["voxel", "mv", "pc"], heads=["mask"], spatial=True,
                    structure="parallel", use_self_mask=True, C=201, foc=(0, 2), seed=0, data_seed=1234)
    else:
        args = dict(B=B, Ns=256, Nq=100, d=256, H=8, L=2, memories=["voxel", "mv", "pc"], heads=[], spatial=True,
                    structure="parallel", seed=0, data_seed=1234)
    _cfg, model, _sd, dd = util.model_case(args)
    set_compute(model, "bf16")
    model.unified_encoder.fused = True
    model.to(dev)
    ddv = {k: v.to(dev) for k, v in dd.items()}
    assert fused._chain_on(dev)

    def step():
        model.zero_grad()
        out = model(dict(ddv))
        # a device-only loss (util.synthetic_loss builds its weights on the host: 100+ ms per call, longer than the neighbour)
        loss = out["query_embeds"].float().square().mean()
        for m in out.get("predictions_mask", []):
            loss = loss + m.float().clamp(min=-50.0).mean()
        for c_ in out.get("predictions_class", []):
            loss = loss + torch.where(torch.isfinite(c_), c_, torch.zeros_like(c_)).float().square().mean()
        loss.backward()
        outs = [out["query_embeds"]] + list(out.get("predictions_mask", [])) + list(out.get("predictions_class", []))
        return [t.detach().clone() for t in outs], {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    step()                                   # warm-up (allocations, LDS opt-ins)
    torch.cuda.synchronize()
    ref_o, ref_g = step()
    torch.cuda.synchronize()
    assert not ops.chain_error(dev)
    beside = False
    for attempt in range(6):
        # a fresh stream per attempt: HIP maps streams onto a few hardware queues round-robin, and a side stream that lands on the
        # SAME queue as the step's stream serialises the step behind the neighbour (measured: tools/probes/chain_neighbour_timing.py
        # -- sporadically a step takes exactly the neighbour's 100 ms whatever the neighbour's size); the launches must have run
        # BESIDE the neighbour in at least one attempt, and every attempt must be correct
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            L.check(L.lib().pq3d_test_occupy_cus(held, lds_kb * 1024, us, L.stream()), "pq3d_test_occupy_cus")
            ev = torch.cuda.Event()
            ev.record()
        got_o, got_g = step()
        torch.cuda.current_stream().synchronize()
        beside = beside or not ev.query()
        torch.cuda.synchronize()
        ops.chain_check(dev)                 # raises ChainHandoffError if any hand-off gave up
        for a, b in zip(got_o, ref_o):
            assert torch.equal(a, b)
        if beside or us < 100000:
            break
    if us >= 100000:
        assert beside, "the neighbour ended before the step did in every attempt: the launches never ran beside it"
    gmax = max(float(v.norm()) for v in ref_g.values())
    for n, g0 in ref_g.items():
        assert float((got_g[n] - g0).norm()) <= 1e-2 * max(float(g0.norm()), 1e-3 * gmax), n


def test_chain_check_raises_and_disables_on_a_set_error_word():
    """The polling side of ADVICE r5: a set error word must surface as an exception and switch the chains off (bench.py: exit
    code 3; TrainStep / GraphedQuery3D: every `chain_check_every` steps and on .check())."""
    from pq3d_amd import fused, ops
    dev = torch.device("cuda", torch.cuda.current_device())
    x = torch.randn(1, 32, 256, device=dev)
    flags = ops.chain_flags(32, dev)
    w = lambda *s: torch.randn(*s, device=dev) * 0.05
    ops.chain_ffn_fwd(x, w(256, 256), w(256), x, w(256), w(256), 1e-5, w(2048, 256), w(2048), w(256, 2048), w(256), w(256), w(256), 1e-5, flags)
    torch.cuda.synchronize()
    ops.chain_check(dev)                     # clean
    err = next(iter(ops._CHAIN_ERR.values()))
    try:
        err.fill_(1)
        assert ops.chain_error(dev)
        with pytest.raises(ops.ChainHandoffError):
            ops.chain_check(dev)
        assert not fused._CHAIN and not ops.chain_error(dev)   # switched off, word cleared
    finally:
        err.zero_()
        fused.set_chain(True)
