"""Test of the LayerNorm-prologue variant of the whole-K GEMM (was in tests/test_gpu_ops.py)."""
LNP_CASES = [
    # (R, d, N, groups, M branches, coef, x, A2, sum_branches, act)
    (800, 256, 256, 3, 3, True, True, True, False, None),      # parallel cross-attention LayerNorm -> Q/K/V projections
    (800, 256, 256, 3, 3, False, True, True, False, None),
    (800, 256, 256, 1, 1, False, True, True, False, None),     # prompt LayerNorm -> prompt query projection
    (800, 256, 2048, 1, 1, False, True, False, False, "relu"),  # self-attention LayerNorm -> FFN linear1
    (800, 256, 256, 3, 4, False, True, True, True, None),      # FFN LayerNorm over 4 K-split partial sums -> next Q projection
    (100, 64, 201, 1, 1, False, False, False, False, None),    # narrow rows, no residual, ragged N (class head)
    (37, 128, 64, 2, 2, True, True, False, False, None),
    (800, 768, 256, 1, 1, False, True, False, False, None),    # d > 256: pq3d_gemm launches the LayerNorm itself
]


@pytest.mark.parametrize("case", LNP_CASES, ids=[f"R{c[0]}d{c[1]}N{c[2]}g{c[3]}M{c[4]}{'c' if c[5] else ''}{'s' if c[8] else ''}" for c in LNP_CASES])
def test_gemm_layernorm_prologue_matches_separate_layernorm(case):
    """pq3d_gemm_desc.ln: add+LayerNorm folded into the consuming projection's prologue (gemm_wk.hip) against the
    separate pq3d_add_ln_fwd launch followed by the same projection: y / mean / rstd / osum and the product agree to fp32
    rounding (the row statistics are reduced in another order), and against torch."""
    R, d, N, G, M, use_coef, use_x, use_a2, sumb, act = case
    B = 1 if R % 100 else R // 100
    rps = R // B
    x = rnd(R, d, seed=1).to(DEV) if use_x else None
    os_ = [rnd(R, d, seed=10 + m).to(DEV) for m in range(M)]
    nb = 1 if sumb else M
    gam = [(1.0 + 0.1 * rnd(d, seed=20 + m)).to(DEV) for m in range(nb)]
    bet = [(0.1 * rnd(d, seed=30 + m)).to(DEV) for m in range(nb)]
    coef = None
    if use_coef:
        k = (torch.rand(B, M, generator=torch.Generator().manual_seed(3)) > 0.5)
        k = k | (k.sum(1, keepdim=True) == 0)
        coef = (k / k.sum(1, keepdim=True)).t().contiguous().float().to(DEV)
    W = [(rnd(N, d, seed=100 + g) * 0.1).to(DEV) for g in range(G)]
    bias = [rnd(N, seed=200 + g).to(DEV) for g in range(G)]
    A2 = [rnd(R, d, seed=50 + g).to(DEV) if g != G - 1 or G == 1 else None for g in range(G)] if use_a2 else None
    res = []
    for fused in (True, False):
        y = torch.full((R, d), float("nan"), device=DEV)
        mean, rstd = torch.zeros(M, R, device=DEV), torch.zeros(M, R, device=DEV)
        osum = torch.zeros(R, d, device=DEV) if sumb else None
        C_ = torch.zeros(G, R, N, device=DEV)
        ln = dict(x=x, o=os_, gamma=gam, beta=bet, coef=coef, eps=1e-5, rows_per_scene=rps, y=y, mean=mean, rstd=rstd,
                  sum_branches=sumb, osum=osum)
        if not fused:
            dsc = ops._ln_desc(x, os_, gam, bet, coef, 1e-5, rps, y, mean, rstd, None)
            dsc.sum_branches, dsc.osum = int(sumb), L.ptr(osum)
            L.check(L.lib().pq3d_add_ln_fwd(C.byref(dsc), L.stream()), "ln")
        L.gemm(M=R, N=N, K=d, A=[y] * G, A2=A2, B=W, bias=bias, Cs=[C_[g] for g in range(G)], ct=L.BF16X3, lda=d, ldb=d, ldc=N,
               act=act, ln=ln if fused else None)
        res.append((y, mean, rstd, osum, C_))
    (y1, m1, r1, s1, c1), (y0, m0, r0, s0, c0) = res
    nm = 1 if sumb else M
    close(y1, y0, F32, "ln-prologue y", atol=2e-6, rtol=2e-6)
    close(m1[:nm], m0[:nm], F32, "mean", atol=1e-6, rtol=1e-6)
    close(r1[:nm], r0[:nm], F32, "rstd", atol=2e-6, rtol=2e-6)
    if sumb:
        assert torch.equal(s1, s0)
    close(c1, c0, F32, "product", atol=1e-5, rtol=1e-5)
    # torch
    xs = x if x is not None else 0
    if sumb:
        yt = torch.nn.functional.layer_norm(xs + sum(os_), (d,), gam[0], bet[0], 1e-5)
    else:
        yt = 0
        for m in range(M):
            w = coef[m].repeat_interleave(rps)[:, None] if coef is not None else 1.0 / M
            yt = yt + w * torch.nn.functional.layer_norm(xs + os_[m], (d,), gam[m], bet[m], 1e-5)
    close(y1, yt, F32, "ln-prologue y vs torch", atol=1e-5, rtol=1e-5)
