"""Test of the one-launch feed-forward variant (was in tests/test_gpu_ops.py)."""
@pytest.mark.gpu
@pytest.mark.parametrize("R,F_,act,p", [(800, 2048, "relu", 0.0), (800, 2048, "gelu", 0.0), (130, 512, "relu", 0.1), (77, 256, "gelu", 0.1),
                                        (64, 1024, "relu", 0.0)])
def test_ffn_one_launch_matches_the_two_products(R, F_, act, p):
    """pq3d_ffn_fwd (csrc/ffn.hip): h / pre bit-comparable with pq3d_gemm's split-bf16 product + epilogue (same arithmetic, same
    dropout site), the sum of the F/256 partial sums equal to linear2 on that h (fp64 reference, fp32-grade tolerance)."""
    from pq3d_amd import fused
    d = 256
    x = rnd(R, d, seed=1).to(DEV)
    w1, b1 = (rnd(F_, d, seed=2) * 0.06).to(DEV), (rnd(F_, seed=3) * 0.1).to(DEV)
    w2, b2 = (rnd(d, F_, seed=4) * 0.03).to(DEV), (rnd(d, seed=5) * 0.1).to(DEV)
    drop = L.Drop(p, 17, torch.tensor([0x1234567], dtype=torch.int64, device=DEV)) if p > 0 else None
    assert fused.ffn_fused_ok(L.BF16X3, d, F_, x, w1, b1, w2, b2)
    h, pre, zp = fused.ffn_fwd(x, w1, b1, w2, b2, act, drop, act == "gelu")
    h_ref, pre_ref = torch.empty(R, F_, device=DEV), (torch.empty(R, F_, device=DEV) if act == "gelu" else None)
    L.gemm(M=R, N=F_, K=d, A=[x], B=[w1], bias=[b1], Cs=[h_ref], C2=[pre_ref], ct=L.BF16X3, lda=d, ldb=d, ldc=F_, act=act, drop=drop)
    torch.testing.assert_close(h, h_ref, rtol=1e-5, atol=1e-6)
    assert (h == 0).eq(h_ref == 0).all()   # same ReLU zeros / dropout mask
    if act == "gelu":
        torch.testing.assert_close(pre, pre_ref, rtol=1e-5, atol=1e-6)
    y = zp.sum(0).double()
    y_ref = h.double() @ w2.double().T + b2.double()
    torch.testing.assert_close(y, y_ref, rtol=2e-5, atol=2e-5)
    # rows are independent of the batch they sit in (bit-exact): the first 40 rows alone
    h2, _, zp2 = fused.ffn_fwd(x[:40].contiguous(), w1, b1, w2, b2, act, None, False)
    if p == 0:
        assert torch.equal(h2, h[:40]) and torch.equal(zp2, zp[:, :40])


