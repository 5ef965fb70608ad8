"""Test of the A-stationary hoisted-projection kernel (was in tests/test_gpu_ops.py; needs option bit 9 of pq3d_gemm_set_wk)."""
@pytest.mark.parametrize("M,N,fams,per,relu,bias", [(8192, 256, 6, 4, False, True), (8192, 256, 2, 3, True, True), (16384, 128, 3, 5, False, False),
                                                   (4096, 256, 5, 6, False, True), (32768, 256, 1, 2, False, True)])
def test_gemm_a_stationary_kernel_matches_per_tile_kernel_bit_for_bit(M, N, fams, per, relu, bias):
    """Groups that share their A operand at K = 256 (the hoisted K/V projections: one key / value input per memory, one weight
    matrix per decoder layer) take the A-stationary kernel (gemm_astat.hip: A fragments in registers for up to 4 output tiles,
    weights streamed through an LDS ring by counted DMA, dedicated store waves).  Same k order and epilogue as the 128 x 128
    per-tile kernel -> identical bits; the per-tile kernel is forced with option bit 9 of pq3d_gemm_set_wk.  Groups are passed
    interleaved (layer-major, as the fused executor lists them), ragged tile counts per work entry included."""
    K = 256
    As = [rnd(M, K, seed=f).to(DEV).bfloat16() for f in range(fams)]
    G = fams * per
    fam_of = [g % fams for g in range(G)]                      # interleaved: consecutive groups use different A operands
    W = [(rnd(N, K, seed=100 + g) * 0.1).to(DEV).bfloat16() for g in range(G)]
    b = [rnd(N, seed=200 + g).to(DEV) if (bias and g % 5 != 4) else None for g in range(G)] if bias else None
    outs = []
    for opt in (1, 1 | (1 << 9)):
        L.lib().pq3d_gemm_set_wk(opt, 0)
        try:
            C_ = torch.full((G, M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
            L.gemm(M=M, N=N, K=K, A=[As[fam_of[g]] for g in range(G)], B=W, bias=b, Cs=[C_[g] for g in range(G)], ct=BF16, lda=K,
                   ldb=K, ldc=N, act="relu" if relu else None)
            torch.cuda.synchronize()
            outs.append(C_)
        finally:
            L.lib().pq3d_gemm_set_wk(1, 0)
    assert torch.isfinite(outs[0].float()).all()
    assert torch.equal(outs[0], outs[1]), float((outs[0].float() - outs[1].float()).abs().max())
    for g in (0, G - 1):
        ref = As[fam_of[g]].float() @ W[g].float().T + (b[g] if (b is not None and b[g] is not None) else 0)
        close(outs[0][g].float(), ref.relu() if relu else ref, BF16, "gemm_astat")


