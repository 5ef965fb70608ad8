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


def test_chain_refuses_what_it_cannot_hold():
    from pq3d_amd import _lib as L, ops
    assert not ops.chain_ffn_ok(2049, 256, 2048) and not ops.chain_ffn_ok(800, 512, 2048) and ops.chain_ffn_ok(800, 256, 2048) and not ops.chain_ffn_ok(800, 256, 1024)
    c = L.ChainFfnDesc()
    c.R, c.d, c.F = 4000, 256, 2048
    assert L.lib().pq3d_chain_ffn_fwd(L.C.byref(c), None) == -1
