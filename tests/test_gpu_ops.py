"""GPU: every C-ABI kernel against the oracle / a plain fp32-fp64 restatement, through the ctypes boundary.
fp32 compute type must match to ~1e-5 (exact-f32 MFMA); bf16 compute type to bf16-operand tolerance."""
import ctypes as C
import math
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from oracle import pq3d_oracle as O
from pq3d_amd import _lib as L
from pq3d_amd import ops
from pq3d_amd._lib import BF16, F32

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).float()


def tol(ct):
    return dict(atol=2e-5, rtol=2e-5) if ct == F32 else dict(atol=3e-2, rtol=3e-2)


def close(a, b, ct, what="", scale_norm=True, **kw):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    t = {**tol(ct), **kw}
    fin = torch.isfinite(b)
    assert torch.equal(fin, torch.isfinite(a)), f"{what}: non-finite pattern differs"
    s = max(1.0, float(b[fin].abs().max())) if scale_norm and fin.any() else 1.0
    err = float((a[fin] - b[fin]).abs().max()) if fin.any() else 0.0
    assert err <= t["atol"] * s + t["rtol"] * s, f"{what}: max err {err:.3e} (scale {s:.2e})"


# ---------------------------------------------------------------------------------------------- GEMM / linear
@pytest.mark.parametrize("ct", [F32, BF16])
@pytest.mark.parametrize("R,K,N", [(100, 64, 64), (200, 256, 201), (37, 3, 64), (64, 2048, 256), (129, 5, 8),
                                   (1, 64, 1)])
@pytest.mark.parametrize("act", [None, "relu", "gelu"])
def test_linear_fwd_bwd(ct, R, K, N, act):
    x, x2, w, b = rnd(R, K), rnd(R, K, seed=1), rnd(N, K, scale=0.1), rnd(N, scale=0.1)
    if ct == BF16:  # bf16-representable operands so that the activation kink is hit identically on both sides
        x, x2, w = (t.bfloat16().float() for t in (x, x2, w))
    gy = rnd(R, N, seed=3)
    ref_in = [t.clone().double().requires_grad_(True) for t in (x, x2, w, b)]
    xs = ref_in[0] + ref_in[1]
    if ct == BF16:  # the kernel rounds the fp32 sum x + x2 to bf16 when staging the MFMA operand
        xs = xs + ((xs.float().bfloat16().double() - xs).detach())
    pre = xs @ ref_in[2].t() + ref_in[3]
    yr = O.activation(pre, act) if act else pre
    yr.backward(gy.double())
    dev = [t.to(DEV).requires_grad_(True) for t in (x, x2, w, b)]
    y = ops.linear(dev[0], dev[1 + 1], dev[3], x2=dev[1], ct=ct, act=act)
    y.backward(gy.to(DEV))
    close(y, yr, ct, "y")
    for name, a, r in zip(("dx", "dx2", "dw", "db"), dev, ref_in):
        close(a.grad, r.grad, ct, name)


@pytest.mark.parametrize("ct", [F32, BF16])
def test_linear_row_mask_fill_and_bf16_out(ct):
    R, K, N = 150, 64, 96
    x, w = rnd(R, K), rnd(N, K, scale=0.1)
    keep = torch.rand(R) > 0.3
    fill = torch.rand(R) > 0.8
    xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    y = ops.linear(xd, wd, None, ct=ct, row_mask=keep.to(DEV), out_dtype=ops.act_dtype(ct))
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = (xr @ wr.t()) * keep[:, None]
    close(y, yr, ct, "row_mask")
    gy = rnd(R, N, seed=5)
    y.backward(gy.to(DEV).to(y.dtype))
    yr.backward(gy)
    close(xd.grad, xr.grad, ct, "dx")
    close(wd.grad, wr.grad, ct, "dw")
    y2 = ops.linear(xd, wd, None, ct=ct, fill_flag=fill.to(DEV), fill_value=float("-inf"))
    close(y2, (x @ w.t()).masked_fill(fill[:, None], float("-inf")), ct, "fill")


@pytest.mark.parametrize("ct", [F32, BF16])
def test_mask_logits(ct):
    B, Ns, Nq, d, M = 2, 75, 21, 64, 3
    ks = [rnd(B, Ns, d, seed=m) for m in range(M)]
    qs = [rnd(B, Nq, d, seed=10 + m) for m in range(M)]
    masks = [torch.rand(B, Ns) > 0.7 for _ in range(M)]
    seg_pad = masks[0] & masks[1] & masks[2]
    ksm = [k * (~m)[..., None] for k, m in zip(ks, masks)]
    kd = [k.to(DEV).to(ops.act_dtype(ct)).requires_grad_(True) for k in ksm]
    qd = [q.to(DEV).to(ops.act_dtype(ct)).requires_grad_(True) for q in qs]
    inv_den = ops.mask_inv_den([m.to(DEV) for m in masks])
    den = sum((~m).float() for m in masks)
    close(inv_den, 1.0 / (den + 1e-8), F32, "inv_den", scale_norm=False, rtol=1e-6, atol=1e-3)
    logits, amask = ops.mask_logits(kd, qd, inv_den, seg_pad.to(DEV), ct=ct)
    kr = [k.detach().float().cpu().requires_grad_(True) for k in kd]
    qr = [q.detach().float().cpu().requires_grad_(True) for q in qd]
    lr = sum(torch.einsum("bld,bmd->blm", k, q) for k, q in zip(kr, qr)) / (den[..., None] + 1e-8)
    lr = lr.masked_fill(seg_pad[..., None], -1e6)
    close(logits, lr, ct, "logits", scale_norm=False, atol=2e-4 if ct == F32 else 0.15, rtol=1e-5 if ct == F32 else 2e-2)
    ar = torch.sigmoid(logits.detach().cpu()).permute(0, 2, 1) < 0.5
    assert torch.equal(amask.cpu(), ar)
    gy = rnd(B, Ns, Nq, seed=9)
    logits.backward(gy.to(DEV))
    lr.backward(gy)
    for a, r in zip(kd + qd, kr + qr):
        close(a.grad, r.grad, ct, "dk/dq")


# ---------------------------------------------------------------------------------------------- attention
def attn_ref(q, k, v, H, scale, zero_attn, kpm, mask, row_open, bias):
    B, Lq, d = q.shape
    Lk, dh = k.shape[1], d // H
    sp = lambda t: t.view(B, -1, H, dh).permute(0, 2, 1, 3)
    s = (sp(q) @ sp(k).transpose(-1, -2)) * scale
    if bias is not None:
        s = s + bias
    if mask is not None:
        m = mask.clone()
        if row_open is not None:
            m[row_open] = False
        s = s.masked_fill(m[:, None], float("-inf"))
    if kpm is not None:
        s = s.masked_fill(kpm[:, None, None, :], float("-inf"))
    vv = sp(v)
    if zero_attn:
        s = torch.cat([s, s.new_zeros(B, H, Lq, 1)], -1)
        vv = torch.cat([vv, vv.new_zeros(B, H, 1, dh)], 2)
    o = torch.softmax(s, -1) @ vv
    return o.permute(0, 2, 1, 3).reshape(B, Lq, d)


@pytest.mark.parametrize("ct", [F32, BF16])
@pytest.mark.parametrize("H,dh", [(4, 16), (8, 32), (2, 64)])
@pytest.mark.parametrize("Lq,Lk,mode", [(16, 128, "kpm"), (100, 130, "mask3d"), (37, 70, "bias"), (200, 333, "kpm"),
                                        (20, 20, "self"),
                                        # key-split paths: 2 splits (>= 8 key blocks), ragged length; 4 splits (>= 64 key blocks), incl. the supported maximum
                                        (100, 1100, "mask3d"), (40, 4200, "mask3d"), (9, 16384, "kpm"), (1, 50, "bias")])
def test_attention_fwd_bwd(ct, H, dh, Lq, Lk, mode):
    B, d = 2, H * dh
    q, k, v = rnd(B, Lq, d, seed=1), rnd(B, Lk, d, seed=2), rnd(B, Lk, d, seed=3)
    g = torch.Generator().manual_seed(Lq * Lk + H)
    kpm = mask = row_open = bias = None
    zero_attn = mode in ("kpm", "mask3d")
    if mode in ("kpm", "self", "bias"):
        kpm = torch.arange(Lk)[None, :] >= torch.tensor([Lk, max(1, Lk // 2)])[:, None]
    if mode == "mask3d":
        mask = torch.rand(B, Lq, Lk, generator=g) < 0.6
        mask[:, 1, :] = True
        mask[0, 3, :] = True
        row_open = mask.all(-1)
    if mode == "bias":
        bias = torch.randn(B, H, Lq, Lk, generator=g)
    ad = ops.act_dtype(ct)
    qd, kd, vd = (t.to(DEV).to(ad).requires_grad_(True) for t in (q, k, v))
    bd = bias.to(DEV).requires_grad_(True) if bias is not None else None
    todev = lambda t: None if t is None else t.to(DEV)
    o = ops.attention(qd, kd, vd, H=H, ct=ct, zero_attn=zero_attn, kpm=todev(kpm), mask=todev(mask),
                      row_open=todev(row_open), bias=bd)
    qr, kr, vr = (t.detach().double().cpu().requires_grad_(True) for t in (qd, kd, vd))
    br = bias.double().requires_grad_(True) if bias is not None else None
    orf = attn_ref(qr, kr, vr, H, 1 / math.sqrt(dh), zero_attn, kpm, mask, row_open, br)
    close(o, orf, ct, "o")
    go = rnd(B, Lq, d, seed=7)
    o.backward(go.to(DEV).to(ad))
    orf.backward(go.to(ad).double())
    close(qd.grad, qr.grad, ct, "dq")
    close(kd.grad, kr.grad, ct, "dk")
    close(vd.grad, vr.grad, ct, "dv")
    if bias is not None:
        close(bd.grad, br.grad, ct, "dbias")


def test_attention_fully_masked_row_with_zero_attn_is_zero():
    B, H, dh, Lq, Lk = 1, 4, 16, 16, 40
    q, k, v = (rnd(B, L, H * dh, seed=s).to(DEV) for L, s in ((Lq, 1), (Lk, 2), (Lk, 3)))
    kpm = torch.ones(B, Lk, dtype=torch.bool, device=DEV)
    o = ops.attention(q, k, v, H=H, ct=F32, zero_attn=True, kpm=kpm)
    assert float(o.abs().max()) == 0.0


def test_mask_row_all():
    m = torch.rand(3, 17, 130) < 0.5
    m[1, 4] = True
    m[2, 0] = True
    assert torch.equal(ops.mask_row_all(m.to(DEV)).cpu(), m.all(-1))
    for Lk in (16, 4096, 1040):   # 16-byte path: rows with a single open key at every position class
        m = torch.ones(2, 40, Lk, dtype=torch.bool)
        for r in range(0, 40, 3):
            m[0, r, (r * 37) % Lk] = False
        m[1, 5, Lk - 1] = False
        m[1, 6, 0] = False
        assert torch.equal(ops.mask_row_all(m.to(DEV)).cpu(), m.all(-1))


# ---------------------------------------------------------------------------------------------- layernorm
@pytest.mark.parametrize("d", [64, 256, 768, 100, 1024, 1152, 2048])   # > 1024: 32 values per lane
@pytest.mark.parametrize("M,with_x,with_coef", [(1, True, False), (3, True, False), (2, False, True), (3, True, True)])
def test_add_layernorm(d, M, with_x, with_coef):
    B, Lq = 3, 11
    x = rnd(B, Lq, d) if with_x else None
    os_ = [rnd(B, Lq, d, seed=5 + m) for m in range(M)]
    gam = [rnd(d, seed=20 + m).abs() + 0.5 for m in range(M)]
    bet = [rnd(d, seed=30 + m) for m in range(M)]
    coef = torch.rand(M, B) + 0.1 if with_coef else None
    leaves = [t.to(DEV).requires_grad_(True) for t in ([x] if with_x else []) + os_ + gam + bet]
    xd = leaves[0] if with_x else None
    rest = leaves[1:] if with_x else leaves
    y = ops.add_layernorm(xd, rest[:M], rest[M:2 * M], rest[2 * M:], eps=1e-5,
                          coef=coef.to(DEV) if with_coef else None)
    refl = [t.clone().double().requires_grad_(True) for t in ([x] if with_x else []) + os_ + gam + bet]
    xr = refl[0] if with_x else 0.0
    rr = refl[1:] if with_x else refl
    yr = 0
    for m in range(M):
        w = coef[m].double()[:, None, None] if with_coef else 1.0 / M
        yr = yr + w * O.layer_norm(xr + rr[m], rr[M + m], rr[2 * M + m])
    close(y, yr, F32, "y")
    gy = rnd(B, Lq, d, seed=77)
    y.backward(gy.to(DEV))
    yr.backward(gy.double())
    for a, r in zip(leaves, refl):
        close(a.grad, r.grad, F32, "grad", atol=5e-5, rtol=5e-5)


@pytest.mark.parametrize("d", [256, 512, 768, 100])
@pytest.mark.parametrize("with_coef", [False, True])
@pytest.mark.parametrize("B", [41, 8])
def test_add_layernorm_three_branch_backward_without_dx_atomics(d, with_coef, B):
    """R >= 512 rows, 3 merged branches (the cross-attention sublayer: 800 rows at config 2, 10240 at the shipped stage-2
    shape): the backward runs every branch of a row in one wave and stores dx once (norm.hip add_ln_bwd_merged_kernel) --
    checked against float64 autograd of the same expression, incl. per-scene branch weights."""
    Lq, M = 100, 3     # 4100 / 800 rows
    x = rnd(B, Lq, d)
    os_ = [rnd(B, Lq, d, seed=5 + m) for m in range(M)]
    gam = [rnd(d, seed=20 + m).abs() + 0.5 for m in range(M)]
    bet = [rnd(d, seed=30 + m) for m in range(M)]
    coef = torch.rand(M, B) + 0.1 if with_coef else None
    leaves = [t.to(DEV).requires_grad_(True) for t in [x] + os_ + gam + bet]
    y = ops.add_layernorm(leaves[0], leaves[1:1 + M], leaves[1 + M:1 + 2 * M], leaves[1 + 2 * M:], eps=1e-5,
                          coef=coef.to(DEV) if with_coef else None)
    refl = [t.clone().double().requires_grad_(True) for t in [x] + os_ + gam + bet]
    yr = 0
    for m in range(M):
        w = coef[m].double()[:, None, None] if with_coef else 1.0 / M
        yr = yr + w * O.layer_norm(refl[0] + refl[1 + m], refl[1 + M + m], refl[1 + 2 * M + m])
    close(y, yr, F32, "y")
    gy = rnd(B, Lq, d, seed=77)
    y.backward(gy.to(DEV))
    yr.backward(gy.double())
    for i, (a, r) in enumerate(zip(leaves, refl)):
        err = float((a.grad.double().cpu() - r.grad).abs().max()) / float(r.grad.abs().max())
        assert err < 2e-5, (i, err)


@pytest.mark.parametrize("d", [1152, 2048, 1024])
def test_add_layernorm_wide_rows_many_rows(d):
    """R >= 4096 rows of more than 1024 columns (32 values per lane): the backward must stay on the 4-wave kernel (64 KB of
    LDS); 1024 columns take the 8-wave kernel.  Checked against float64 autograd."""
    R = 4100
    x, o = rnd(R, d), rnd(R, d, seed=5)
    gam, bet = rnd(d, seed=20).abs() + 0.5, rnd(d, seed=30)
    leaves = [t.to(DEV).requires_grad_(True) for t in (x, o, gam, bet)]
    y = ops.add_layernorm(leaves[0], [leaves[1]], [leaves[2]], [leaves[3]], eps=1e-5)
    refl = [t.clone().double().requires_grad_(True) for t in (x, o, gam, bet)]
    yr = O.layer_norm(refl[0] + refl[1], refl[2], refl[3])
    close(y, yr, F32, "y")
    gy = rnd(R, d, seed=77)
    y.backward(gy.to(DEV))
    yr.backward(gy.double())
    for i, (a, r) in enumerate(zip(leaves, refl)):
        err = float((a.grad.double().cpu() - r.grad).abs().max()) / float(r.grad.abs().max())
        assert err < 5e-5, (i, err)


# ---------------------------------------------------------------------------------------------- misc
def test_pairwise_locs_and_fourier_and_spatial_bias():
    c = torch.rand(2, 23, 3, generator=torch.Generator().manual_seed(11)) * 4
    close(ops.pairwise_locs(c.to(DEV)), O.calc_pairwise_locs(c), F32, "pairwise", atol=2e-6, rtol=2e-6)
    G = torch.randn(3, 32, generator=torch.Generator().manual_seed(12))
    cmin, cmax = torch.tensor([[0., 0, 0], [-1, -1, 0]]), torch.tensor([[4., 4, 4], [5, 4, 3]])
    close(ops.fourier(c.to(DEV), cmin.to(DEV), cmax.to(DEV), G.to(DEV)), O.fourier_embed(c, G, cmin, cmax), F32,
          "fourier", atol=2e-5, rtol=0)
    pl = O.calc_pairwise_locs(c)
    W, bw = rnd(4, 5), rnd(4)
    Wd, bd = W.to(DEV).requires_grad_(True), bw.to(DEV).requires_grad_(True)
    bias = ops.spatial_bias(pl.to(DEV), Wd, bd)
    Wr, br = W.clone().requires_grad_(True), bw.clone().requires_grad_(True)
    pre = pl @ Wr.t() + br
    ref = torch.log(torch.clamp(torch.relu(pre), min=1e-6)).permute(0, 3, 1, 2)
    # log() next to the relu/clamp corner (0 < v < 1e-3) amplifies the fp32 rounding of v without bound: compare
    # the well-conditioned points at 1e-5 and the corner points loosely
    wc = ((pre > 1e-3) | (pre < -1e-3)).permute(0, 3, 1, 2)
    close(torch.where(wc.to(DEV), bias, torch.zeros_like(bias)), torch.where(wc, ref, torch.zeros_like(ref)), F32,
          "spatial bias", atol=1e-5, rtol=1e-5)
    close(bias, ref, F32, "spatial bias (corner points)", atol=5e-3, rtol=0)
    gy = rnd(*ref.shape, seed=4)
    bias.backward(gy.to(DEV))
    ref.backward(gy)
    close(Wd.grad, Wr.grad, F32, "dW", atol=1e-4, rtol=1e-4)
    close(bd.grad, br.grad, F32, "dbw", atol=1e-4, rtol=1e-4)


def test_gate_fill_colsum_scatter():
    q, u, g = rnd(5, 7, 16), rnd(5, 7, 16, seed=1), rnd(5, 7, 16, seed=2)
    dl = [t.to(DEV).requires_grad_(True) for t in (q, u, g)]
    rl = [t.clone().requires_grad_(True) for t in (q, u, g)]
    y = ops.gate_mix(*dl)
    s = torch.sigmoid(rl[2])
    yr = (1 - s) * rl[0] + s * rl[1]
    close(y, yr, F32, "gate")
    y.sum().backward(); yr.sum().backward()
    for a, r in zip(dl, rl):
        close(a.grad, r.grad, F32, "gate grad")
    x = rnd(9, 21).to(DEV).requires_grad_(True)
    cols = torch.tensor([0, 2, 20], dtype=torch.int32, device=DEV)
    f = ops.fill_cols(x, cols, float("-inf"))
    assert torch.isinf(f[:, [0, 2, 20]]).all() and torch.equal(f[:, 1], x[:, 1])
    torch.where(torch.isfinite(f), f, torch.zeros_like(f)).sum().backward()
    assert float(x.grad[:, [0, 2, 20]].abs().max()) == 0 and float(x.grad[:, 1].min()) == 1
    big = rnd(1000, 130)
    close(ops.colsum(big.to(DEV)), big.sum(0), F32, "colsum", atol=1e-4, rtol=1e-5)
    src = rnd(5000, 96)
    idx = torch.randint(0, 300, (5000,))
    sd = src.to(DEV).requires_grad_(True)
    out = ops.scatter_mean(sd, idx.to(DEV), 320)
    sr = src.clone().requires_grad_(True)
    outr = O.scatter_mean(sr, idx, 320)
    close(out, outr, F32, "scatter_mean", atol=1e-5, rtol=1e-5)
    gy = rnd(320, 96, seed=3)
    out.backward(gy.to(DEV)); outr.backward(gy)
    close(sd.grad, sr.grad, F32, "scatter_mean grad")


def test_multiscale_segment_pool_matches_oracle():
    """Voxel -> segment pooling of a coarse backbone level through the composed parent index
    (pcd_mask3d_encoder.py:133-152) against the oracle's materialised up-sampling + scatter_mean; two scenes batched
    through offset segment ids; gradient w.r.t. the coarse features; parents recovered from coordinates."""
    g = torch.Generator().manual_seed(5)
    fine = torch.cat([torch.cat([torch.full((n, 1), b), torch.randint(-40, 40, (n, 3), generator=g)], 1)
                      for b, n in ((0, 3000), (1, 2200))]).unique(dim=0)
    N = fine.shape[0]
    maps, coords, cur = [], [], fine
    for lvl in range(1, 4):                      # strides 2, 4, 8: the chain of stride-2 poolings
        cc, par = O.pooling_transpose_parents(cur, 2 ** lvl)
        maps.append(par); coords.append(cc); cur = cc
    S = 57
    seg = torch.randint(0, S, (N,), generator=g) + fine[:, 0] * S          # segment ids offset per scene
    for lvl in (1, 3):
        Nc = coords[lvl - 1].shape[0]
        feat = rnd(Nc, 96, seed=lvl)
        parent = ops.compose_parents([m.to(DEV) for m in maps[:lvl]])
        assert torch.equal(parent.cpu(), ops.compose_parents(maps[:lvl]))
        # the same map straight from the coordinates of the two levels
        assert torch.equal(ops.parents_from_coords(fine.to(DEV), coords[lvl - 1].to(DEV), 2 ** lvl), parent)
        fd = feat.to(DEV).requires_grad_(True)
        out = ops.upsample_scatter_mean(fd, parent, seg.to(DEV), 2 * S)
        fr = feat.clone().requires_grad_(True)
        outr = O.multiscale_segment_pool(fr, ops.compose_parents(maps[:lvl]), seg, 2 * S)
        close(out, outr, F32, f"multiscale pool level {lvl}", atol=1e-5, rtol=1e-5)
        gy = rnd(2 * S, 96, seed=9)
        out.backward(gy.to(DEV)); outr.backward(gy)
        close(fd.grad, fr.grad, F32, f"multiscale pool grad level {lvl}", atol=1e-5, rtol=1e-5)
    missing = ops.parents_from_coords(torch.tensor([[0, 1000, 0, 0]], device=DEV), coords[0].to(DEV), 2)
    assert int(missing[0]) == -1


@pytest.mark.parametrize("M,N,K,G", [(8192, 256, 256, 6), (2100, 128, 64, 32), (4096, 512, 128, 2)])
def test_gemm_nt128_tile_matches_64_tile_bit_for_bit(M, N, K, G):
    """Big plain bf16 NT products take the 128x128-tile kernel (gemm128.hip) when both operands are bf16; with fp32 weights
    the 64x64-tile kernel converts them in flight.  Same rounding points, same k order -> identical bits; also vs torch."""
    A = [rnd(M, K, seed=g).to(DEV).bfloat16() for g in range(G)]
    W = [(rnd(N, K, seed=100 + g) * 0.1).to(DEV) for g in range(G)]
    b = [rnd(N, seed=200 + g).to(DEV) for g in range(G)]
    C_new = torch.zeros(G, M, N, dtype=torch.bfloat16, device=DEV)
    C_old = torch.zeros_like(C_new)
    L.gemm(M=M, N=N, K=K, A=A, B=[w.bfloat16() for w in W], bias=b, Cs=[C_new[g] for g in range(G)], ct=BF16, lda=K, ldb=K, ldc=N)
    L.gemm(M=M, N=N, K=K, A=A, B=W, bias=b, Cs=[C_old[g] for g in range(G)], ct=BF16, lda=K, ldb=K, ldc=N)
    assert torch.equal(C_new, C_old), float((C_new.float() - C_old.float()).abs().max())
    for g in (0, G - 1):
        ref = A[g].float() @ W[g].bfloat16().float().T + b[g]
        close(C_new[g].float(), ref, BF16, "gemm128")


@pytest.mark.parametrize("R,M,N,G,acc", [(8192, 256, 256, 6, False), (4096, 128, 256, 12, True), (1024, 256, 128, 32, True)])
def test_gemm_tt128_weight_gradient_tile(R, M, N, G, acc):
    """Big bf16 weight-gradient products dW = dY^T X (+ fused bias gradient) take the 128x128-tile split-K kernel
    (gemm128.hip): products of bf16 values are exact in fp32, so the result matches the fp32 matmul of the same bf16
    operands up to the summation order; accumulate=True adds onto the existing contents."""
    gs = [rnd(R, M, seed=g).to(DEV).bfloat16() for g in range(G)]
    xs = [rnd(R, N, seed=50 + g).to(DEV).bfloat16() for g in range(G)]
    base = rnd(G, M, N, seed=7).to(DEV) if acc else torch.zeros(G, M, N, device=DEV)
    dW = base.clone() if acc else torch.full((G, M, N), 123.0, device=DEV)      # non-accumulating call must overwrite
    cb = torch.zeros(G, M, device=DEV) if acc else torch.full((G, M), -5.0, device=DEV)
    L.gemm(M=M, N=N, K=R, A=gs, B=xs, Cs=[dW[g] for g in range(G)], ct=BF16, lda=M, ldb=N, ldc=N, transA=True, transB=True,
           splitk=2, accumulate=acc, colsum=[cb[g] for g in range(G)])
    for g in (0, G // 2, G - 1):
        ref = base[g] + gs[g].float().T @ xs[g].float()
        err = float((dW[g] - ref).abs().max()) / float(ref.abs().max())
        assert err < 2e-5, (g, err)
        refb = gs[g].float().sum(0)
        assert float((cb[g] - refb).abs().max()) / float(refb.abs().max()) < 2e-5


@pytest.mark.parametrize("M,kc,outs", [(8192, 8, 3), (8200, 8, 3), (2050, 12, 1)])
def test_gemm_nt128_kconcat_fp32_out(M, kc, outs):
    """K-concatenated NT products with fp32 output on the 128x128-tile kernel (the decoder's input-gradient sums over
    layers): sum over kc (A_g, B_g) pairs per output, several outputs per launch, ragged last row tile."""
    d = 256
    G = kc * outs
    A = [rnd(M, d, seed=g).to(DEV).bfloat16() for g in range(G)]
    Bt = [(rnd(d, d, seed=300 + g) * 0.1).to(DEV).bfloat16() for g in range(G)]       # [n_out, k]
    C = torch.full((outs, M, d), 7.0, device=DEV)
    L.gemm(M=M, N=d, K=d, A=A, B=Bt, Cs=[c for o in range(outs) for c in [C[o]] + [None] * (kc - 1)], ct=BF16, lda=d, ldb=d,
           ldc=d, kconcat=kc)
    for o in range(outs):
        ref = sum(A[o * kc + g].float() @ Bt[o * kc + g].float().T for g in range(kc))
        assert float((C[o] - ref).abs().max()) / float(ref.abs().max()) < 2e-5
    # "+ aux" epilogue (gradient accumulation onto an earlier product)
    aux = rnd(outs, M, d, seed=9).to(DEV)
    C2 = torch.empty_like(C)
    L.gemm(M=M, N=d, K=d, A=A, B=Bt, Cs=[c for o in range(outs) for c in [C2[o]] + [None] * (kc - 1)],
           aux=[c for o in range(outs) for c in [aux[o]] + [None] * (kc - 1)], act_grad="add", ct=BF16, lda=d, ldb=d, ldc=d,
           kconcat=kc)
    assert float((C2 - (C + aux)).abs().max()) <= 2e-6 * float(C.abs().max())


def test_upsample_scatter_mean_skips_out_of_range_rows():
    """Voxels whose parent (-1: the coarse level has no such voxel) or segment id is out of range contribute nothing,
    forward and backward; empty segments are zero."""
    src = rnd(6, 8, seed=1).to(DEV).requires_grad_(True)
    parent = torch.tensor([0, 5, -1, 2, 2, 7, 3], device=DEV)          # 7 and -1 are out of range for 6 coarse rows
    seg = torch.tensor([1, 1, 1, 9, 0, 0, -2], device=DEV)              # 9 and -2 are out of range for 4 segments
    out = ops.upsample_scatter_mean(src, parent, seg, 4)
    want = torch.zeros(4, 8, device=DEV)
    want[1] = (src[0] + src[5]).detach() / 2
    want[0] = src[2].detach()
    assert torch.allclose(out, want, atol=1e-6) and float(out[2:].detach().abs().max()) == 0.0
    out.sum().backward()
    g = torch.zeros(6, 8, device=DEV)
    g[0] = 0.5; g[5] = 0.5; g[2] = 1.0
    assert torch.allclose(src.grad, g, atol=1e-6)


def test_colsum_accumulate_long_reduction_row_slices():
    """Bias gradients into the arena: long accumulating column sums run as row slices with atomic adds (misc.hip)."""
    from pq3d_amd import fused as F
    for R, N, G, dt in ((10240, 768, 3, torch.float32), (4099, 200, 2, torch.float32), (8192, 256, 4, torch.bfloat16), (600, 64, 2, torch.float32)):
        xs = [rnd(R, N, seed=90 + g).to(DEV).to(dt) for g in range(G)]
        base = [rnd(N, seed=190 + g).to(DEV) for g in range(G)]
        outs = [b.clone() for b in base]
        F._colsum_acc(xs, outs, R)
        for x, b, o in zip(xs, base, outs):
            ref = b.double() + x.double().sum(0)
            assert float((o.double() - ref).abs().max()) <= 2e-3 * max(1.0, float(ref.abs().max())) * (8 if dt == torch.bfloat16 else 1) * 1e-1 + 1e-3


# ---------------------------------------------------------------------------------------------- small bf16 cross-attention
@pytest.mark.parametrize("za", [True, False])
@pytest.mark.parametrize("B,Lq,Lk,H,dh", [(6, 80, 80, 12, 64), (4, 80, 32, 12, 64), (5, 100, 32, 8, 32), (3, 17, 5, 2, 32),
                                          (2, 128, 128, 4, 64), (3, 1, 128, 2, 32), (2, 33, 97, 2, 64)])
def test_attention_small_cross_kernels(B, Lq, Lk, H, dh, za):
    """attn_ca.hip (bf16, at most 128 queries and 128 keys per (scene, head): the shipped stage-2 decoder's cross-attention
    over <= 80 objects, prompt tokens) against an fp64 reference of the same bf16 inputs -- output and every gradient
    within the bf16 bars of the general kernels -- and against the general kernels on the same inputs; a scene's result
    does not depend on the amount of key padding behind it (bit-identical)."""
    d = H * dh
    q, k, v = rnd(B, Lq, d, seed=11), rnd(B, Lk, d, seed=12), rnd(B, Lk, d, seed=13)
    vl = torch.tensor([Lk] + [max(1, (Lk * (3 + i)) // 7) for i in range(B - 1)])
    kpm = torch.arange(Lk)[None, :] >= vl[:, None]
    go = rnd(B, Lq, d, seed=17)
    lib = L.lib()
    res = {}
    for ca in (1, 0):
        old = lib.pq3d_attn_resident((16 if ca else 0) | 15)
        try:
            qd, kd, vd = (t.to(DEV).bfloat16().requires_grad_(True) for t in (q, k, v))
            o = ops.attention(qd, kd, vd, H=H, ct=BF16, zero_attn=za, kpm=kpm.to(DEV))
            o.backward(go.to(DEV).bfloat16())
            res[ca] = (o.detach(), qd.grad, kd.grad, vd.grad)
        finally:
            lib.pq3d_attn_resident(old)
    qr, kr, vr = (t.bfloat16().double().requires_grad_(True) for t in (q, k, v))
    orf = attn_ref(qr, kr, vr, H, 1 / math.sqrt(dh), za, kpm, None, None, None)
    orf.backward(go.bfloat16().double())
    for name, a, a_old, r in zip(("o", "dq", "dk", "dv"), res[1], res[0], (orf, qr.grad, kr.grad, vr.grad)):
        assert torch.isfinite(a.float()).all(), name
        assert _relL2(a, r) <= 2e-2, f"{name}: small cross-attention vs fp64 relL2 {_relL2(a, r):.2e}"
        assert _relL2(a, a_old) <= 2e-2, f"{name}: small cross-attention vs general kernels relL2 {_relL2(a, a_old):.2e}"
    # padded keys behind a scene: exact zeros in dK / dV, and the scene's results do not change with their number
    for bi in range(1, B):
        n = int(vl[bi])
        assert float(res[1][2][bi, n:].float().abs().max() if n < Lk else 0.0) == 0.0
        assert float(res[1][3][bi, n:].float().abs().max() if n < Lk else 0.0) == 0.0
    if Lk >= 48:
        Ls = Lk - 16     # scene 1 alone with fewer padded keys behind it (its valid keys fit: vl[1] <= 4/7 Lk)
        assert int(vl[1]) <= Ls
        qd, kd, vd = (t[1:2].to(DEV).bfloat16().requires_grad_(True) for t in (q, k[:, :Ls].contiguous(), v[:, :Ls].contiguous()))
        o2 = ops.attention(qd, kd, vd, H=H, ct=BF16, zero_attn=za, kpm=kpm[1:2, :Ls].to(DEV))
        o2.backward(go[1:2].to(DEV).bfloat16())
        assert torch.equal(o2.detach(), res[1][0][1:2]) and torch.equal(qd.grad, res[1][1][1:2])
        assert torch.equal(kd.grad, res[1][2][1:2, :Ls]) and torch.equal(vd.grad, res[1][3][1:2, :Ls])


# ---------------------------------------------------------------------------------------------- resident backward
def _relL2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


@pytest.mark.parametrize("H,dh", [(8, 32), (2, 64)])
@pytest.mark.parametrize("Lq,Lk,mode,B", [(100, 1024, "kpm", 3), (100, 1024, "mask3d", 2), (200, 4096, "mask3d", 2),
                                          (64, 256, "mask3d", 2), (224, 512, "kpm", 2), (7, 160, "kpm", 2),
                                          (33, 2048, "kpm", 2)])
def test_attention_backward_resident_path(H, dh, Lq, Lk, mode, B):
    """The all-queries-resident single-pass backward (attn_resident.hip; cross-attention shape) against an fp64 reference
    -- relative L2 <= 2e-2 per gradient (bf16 operands) -- and against the general two-kernel backward on the same inputs
    (same arithmetic: they may differ by bf16 rounding of dS only)."""
    d = H * dh
    q, k, v = rnd(B, Lq, d, seed=1), rnd(B, Lk, d, seed=2), rnd(B, Lk, d, seed=3)
    g = torch.Generator().manual_seed(Lq * Lk + H)
    vl = torch.tensor([Lk] + [max(1, (Lk * (3 + i)) // 7) for i in range(B - 1)])
    kpm = torch.arange(Lk)[None, :] >= vl[:, None]
    mask = row_open = None
    if mode == "mask3d":
        mask = torch.rand(B, Lq, Lk, generator=g) < 0.6
        mask[:, 1, :] = True
        row_open = mask.all(-1)
        kpm = None
    todev = lambda t: None if t is None else t.to(DEV)
    go = rnd(B, Lq, d, seed=7)
    res = {}
    lib = L.lib()
    for resident in (1, 0):
        old = lib.pq3d_attn_resident(2 | resident)
        try:
            qd, kd, vd = (t.to(DEV).bfloat16().requires_grad_(True) for t in (q, k, v))
            o = ops.attention(qd, kd, vd, H=H, ct=BF16, zero_attn=True, kpm=todev(kpm), mask=todev(mask),
                              row_open=todev(row_open))
            o.backward(go.to(DEV).bfloat16())
            res[resident] = (qd.grad, kd.grad, vd.grad)
        finally:
            lib.pq3d_attn_resident(old)
    qr, kr, vr = (t.bfloat16().double().requires_grad_(True) for t in (q, k, v))
    orf = attn_ref(qr, kr, vr, H, 1 / math.sqrt(dh), True, kpm, mask, row_open, None)
    orf.backward(go.bfloat16().double())
    for name, a, a_old, r in zip(("dq", "dk", "dv"), res[1], res[0], (qr.grad, kr.grad, vr.grad)):
        assert torch.isfinite(a.float()).all(), name
        assert _relL2(a, r) <= 2e-2, f"{name}: resident vs fp64 relL2 {_relL2(a, r):.2e}"
        assert _relL2(a_old, r) <= 2e-2, f"{name}: two-kernel vs fp64 relL2 {_relL2(a_old, r):.2e}"
        assert _relL2(a, a_old) <= 1e-2, f"{name}: resident vs two-kernel relL2 {_relL2(a, a_old):.2e}"
    if kpm is not None:   # gradients of padded keys are exactly zero
        pad = kpm.to(DEV)
        assert float(res[1][1].float()[pad].abs().max()) == 0.0 and float(res[1][2].float()[pad].abs().max()) == 0.0


@pytest.mark.parametrize("case", ["c2", "c5_split", "padded_tail", "dropout", "no_zero_key", "short_keys"])
def test_attention_forward_resident_path(case):
    """The all-keys-resident forward (attn_resident.hip: the whole key/value slice of a (scene, head) in LDS behind one
    barrier, no key split up to 1024 keys) against the streaming kernel on the same inputs -- same arithmetic, bf16
    rounding points and dropout masks, only the block order of the online softmax differs -- and against an fp64
    reference.  The backward consumes the saved log-sum-exp, so the gradients check that too."""
    cfg = {"c2": dict(B=6, Lq=100, Lk=1024), "c5_split": dict(B=3, Lq=100, Lk=2048),
           "padded_tail": dict(B=5, Lq=37, Lk=1000, pad=True), "dropout": dict(B=4, Lq=100, Lk=512, drop=True),
           "no_zero_key": dict(B=2, Lq=128, Lk=640, zero=False), "short_keys": dict(B=2, Lq=16, Lk=128, pad=True)}[case]
    B, H, dh, Lq, Lk = cfg["B"], 8, 32, cfg["Lq"], cfg["Lk"]
    d = H * dh
    q, k, v, go = rnd(B, Lq, d, seed=1), rnd(B, Lk, d, seed=2), rnd(B, Lk, d, seed=3), rnd(B, Lq, d, seed=7)
    kpm = None
    if cfg.get("pad"):
        kpm = torch.zeros(B, Lk, dtype=torch.bool)
        for b in range(B):
            kpm[b, Lk - 1 - (Lk // 8) * b:] = True     # ragged tails
        kpm[B - 1, :64] = True                         # a fully padded leading block
    zero = cfg.get("zero", True)
    drop = ops.make_drop(0.1, 4242, DEV) if cfg.get("drop") else None
    lib = L.lib()
    res = {}
    for mode in (7, 3):
        old = lib.pq3d_attn_resident(mode)
        try:
            qd, kd, vd = (t.to(DEV).bfloat16().requires_grad_(True) for t in (q, k, v))
            o = ops.attention(qd, kd, vd, H=H, ct=BF16, zero_attn=zero, kpm=None if kpm is None else kpm.to(DEV), drop=drop)
            o.backward(go.to(DEV).bfloat16())
            res[mode] = (o.detach(), qd.grad, kd.grad, vd.grad)
        finally:
            lib.pq3d_attn_resident(old)
    for name, a, a_old in zip(("o", "dq", "dk", "dv"), res[7], res[3]):
        assert torch.isfinite(a.float()).all(), name
        assert _relL2(a, a_old.double()) <= 1e-2, f"{case} {name}: resident vs streaming relL2 {_relL2(a, a_old.double()):.2e}"
    if drop is None:
        qr, kr, vr = (t.bfloat16().double() for t in (q, k, v))
        orf = attn_ref(qr, kr, vr, H, 1 / math.sqrt(dh), zero, kpm, None, None, None)
        assert _relL2(res[7][0], orf) <= 1e-2, f"{case}: resident vs fp64 relL2 {_relL2(res[7][0], orf):.2e}"
        assert _relL2(res[3][0], orf) <= 1e-2


@pytest.mark.parametrize("H,dh", [(4, 16), (8, 32), (2, 64)])
@pytest.mark.parametrize("Lq,Lk,mode", [(100, 100, "bias"), (100, 100, "kpm"), (128, 128, "bias"), (16, 16, "kpm"), (1, 37, "bias"),
                                        (77, 5, "kpm")])
def test_attention_small_fp32_kernels(H, dh, Lq, Lk, mode):
    """The one-workgroup-per-head fp32 kernels (attn_small.hip; the decoder's self-attention) against an fp64 reference and
    against the general streaming kernels on the same inputs."""
    B, d = 3, H * dh
    q, k, v = rnd(B, Lq, d, seed=1), rnd(B, Lk, d, seed=2), rnd(B, Lk, d, seed=3)
    kpm = torch.arange(Lk)[None, :] >= torch.tensor([Lk, max(1, Lk // 2), max(1, Lk - 3)])[:, None]
    bias = torch.randn(B, H, Lq, Lk, generator=torch.Generator().manual_seed(5)) if mode == "bias" else None
    go = rnd(B, Lq, d, seed=7)
    lib = L.lib()
    res = {}
    for small in (2, 0):
        old = lib.pq3d_attn_resident(1 | small)
        try:
            qd, kd, vd = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
            bd = bias.to(DEV).requires_grad_(True) if bias is not None else None
            o = ops.attention(qd, kd, vd, H=H, ct=F32, kpm=kpm.to(DEV), bias=bd)
            o.backward(go.to(DEV))
            res[small] = (o.detach(), qd.grad, kd.grad, vd.grad, bd.grad if bd is not None else None)
        finally:
            lib.pq3d_attn_resident(old)
    qr, kr, vr = (t.double().requires_grad_(True) for t in (q, k, v))
    br = bias.double().requires_grad_(True) if bias is not None else None
    orf = attn_ref(qr, kr, vr, H, 1 / math.sqrt(dh), False, kpm, None, None, br)
    orf.backward(go.double())
    refs = (orf, qr.grad, kr.grad, vr.grad, br.grad if br is not None else None)
    for name, a, a_old, r in zip(("o", "dq", "dk", "dv", "dbias"), res[2], res[0], refs):
        if r is None:
            continue
        close(a, r, F32, name + " small-kernel vs fp64", atol=5e-5, rtol=5e-5)
        close(a, a_old, F32, name + " small-kernel vs streaming", atol=5e-5, rtol=5e-5)


def test_copy_many_one_launch_pack():
    """pq3d_copy_many: many small fp32 copies in one launch (the gradient pack of the data-parallel reducer), incl. odd
    lengths, unaligned slices of a flat buffer and more than 64 pairs (two launches)."""
    g = torch.Generator().manual_seed(0)
    sizes = [1, 3, 4, 255, 256, 257, 1000, 65536 + 5] + [7 + i for i in range(70)]
    srcs = [torch.randn(n, generator=g).to(DEV) for n in sizes]
    flat = torch.zeros(sum(sizes) + 3, device=DEV)
    dsts, off = [], 3                      # start at an odd offset: the slices are not 16-byte aligned
    for n in sizes:
        dsts.append(flat[off:off + n]); off += n
    ops.copy_many(dsts, srcs)
    for d_, s_ in zip(dsts, srcs):
        assert torch.equal(d_, s_)
    assert float(flat[:3].abs().sum()) == 0.0


WK_CASES = [
    # (M, N, K, groups, kconcat, ct, A dtype, transB, A2, splitk, act, act_grad, C dtype)
    (800, 256, 256, 3, 0, "x3", "f32", False, True, 1, None, None, "f32"),      # Q / QKV projections (+ query_pos)
    (800, 256, 256, 1, 0, "x3", "f32", False, False, 1, None, None, "f32"),     # self-attention output projection
    (800, 2048, 256, 1, 0, "x3", "f32", False, False, 1, "relu", None, "f32"),  # FFN linear1
    (800, 256, 512, 4, 0, "x3", "f32", False, False, 1, None, None, "f32"),     # FFN linear2, K split over groups
    (800, 256, 256, 3, 0, "bf16", "bf16", False, False, 1, None, None, "f32"),  # cross-attention output projection
    (800, 256, 256, 1, 0, "bf16", "f32", True, False, 1, None, "add", "f32"),   # input gradient + residual
    (800, 256, 256, 3, 3, "bf16", "bf16", True, False, 1, None, "add", "f32"),  # K-concatenated dQ.Wq over memories
    (800, 256, 256, 3, 0, "bf16", "f32", True, False, 1, None, None, "bf16"),   # d(out-proj input), bf16 out
    (800, 2048, 256, 1, 0, "bf16", "f32", True, False, 1, None, "relu", "bf16"),  # FFN backward through linear2 + ReLU mask
    (800, 256, 2048, 1, 0, "bf16", "bf16", True, False, 4, None, "skip_bias", "f32"),  # FFN backward through linear1, split-K
    (100, 201, 256, 1, 0, "x3", "f32", False, False, 1, None, None, "f32"),     # class head: ragged N
    (37, 72, 40, 2, 0, "bf16", "f32", True, False, 1, None, None, "f32"),       # ragged everything (K < one chunk)
    (1600, 512, 520, 2, 2, "x3", "f32", False, False, 1, "gelu", None, "f32"),  # K not a multiple of the chunk, kconcat
]


@pytest.mark.parametrize("case", WK_CASES, ids=[f"M{c[0]}N{c[1]}K{c[2]}g{c[3]}k{c[4]}{c[5]}{'T' if c[7] else 'N'}" for c in WK_CASES])
def test_gemm_whole_k_tiles_match_64_tile_bit_for_bit(case):
    """The small-M launches of the query side take the whole-K kernels (gemm_wk.hip).  Same MFMA order, same epilogue ->
    the same bits as the 64x64-tile pipeline kernel (split-K: atomics, compared to fp32 round-off); also vs torch."""
    M, N, K, G, kc, ct_, adt, tb, a2, sk, act, ag, cdt = case
    no_bias = ag == "skip_bias"      # split-K launches add no bias (atomics epilogue)
    ag = None if no_bias else ag
    ct = L.BF16X3 if ct_ == "x3" else BF16
    tdt = lambda n: torch.bfloat16 if n == "bf16" else torch.float32
    A = [rnd(M, K, seed=g).to(DEV).to(tdt(adt)) for g in range(G)]
    A2 = [rnd(M, K, seed=50 + g).to(DEV) if (a2 and g != G - 1) else None for g in range(G)] if a2 else None
    W = [(rnd(K, N, seed=100 + g) * 0.1).to(DEV) if tb else (rnd(N, K, seed=100 + g) * 0.1).to(DEV) for g in range(G)]
    nout = G // kc if kc else G
    b = [rnd(N, seed=200 + g).to(DEV) if (not kc or g % kc == 0) else None for g in range(G)] if not (ag or no_bias) else None
    aux = None
    if ag:
        aux = [(rnd(M, N, seed=300 + g).to(DEV).to(tdt(cdt if ag == "relu" else "f32"))) if (not kc or g % kc == 0) else None
               for g in range(G)]
    outs = []
    try:
        for wk in (1 | (1 << 8), 1 | (1 << 4) | (1 << 6) | (1 << 8), 1 | (2 << 4) | (1 << 6) | (1 << 8), 1 | (2 << 4) | (2 << 6) | (1 << 8), 0):
            L.lib().pq3d_gemm_set_wk(wk, 0)   # automatic plan, then 32/128, 64/128, 64/256 tiles forced, then the 64x64 kernel
            C_ = torch.zeros(nout, M, N, dtype=tdt(cdt), device=DEV)
            Cs = [C_[g // kc] if (kc and g % kc == 0) else (C_[g] if not kc else None) for g in range(G)]
            L.gemm(M=M, N=N, K=K, A=A, A2=A2, B=W, bias=b, Cs=Cs, aux=aux, ct=ct, lda=K, ldb=N if tb else K, ldc=N, transB=tb,
                   kconcat=kc, splitk=sk, act=act, act_grad=ag)
            outs.append(C_)
    finally:
        L.lib().pq3d_gemm_set_wk(1, 0)
    old = outs[-1]
    for new in outs[:-1]:
        if sk > 1:
            assert float((new - old).abs().max()) <= 1e-4 * float(old.abs().max())
        else:
            assert torch.equal(new, old), float((new.float() - old.float()).abs().max())
    new = outs[0]
    # against torch (first output)
    rd = lambda t: t.float() if ct_ == "x3" else t.bfloat16().float()
    acc = torch.zeros(M, N, device=DEV)
    for g in range(kc or 1):
        a = A[g].float() + (A2[g] if (A2 is not None and A2[g] is not None) else 0)
        acc += rd(a) @ (rd(W[g]) if tb else rd(W[g]).T)
    if b is not None:
        acc += b[0]
    if act == "relu":
        acc = acc.relu()
    elif act == "gelu":
        acc = torch.nn.functional.gelu(acc)
    if ag == "add":
        acc += aux[0].float()
    elif ag == "relu":
        acc = acc * (aux[0].float() > 0)
    close(new[0].float(), acc, F32 if ct_ == "x3" else BF16, "gemm_wk")


# ---------------------------------------------------------------------------------------------- segment pooling, adversarial
def test_scatter_mean_adversarial_cases_match_the_published_definition():
    """pq3d_scatter_mean_* against the oracle's restatement of torch_scatter's published definition on the adversarial
    inputs of tests/test_oracle_golden.py (unsorted / duplicate ids, empty segments, one segment takes all, dim_size beyond
    the largest id, single row), forward and backward."""
    from tests.test_oracle_golden import SCATTER_CASES
    for name, make in sorted(SCATTER_CASES.items()):
        src, idx, n = make(np.random.default_rng(0))
        if len(src) == 0:
            continue        # an empty scene never reaches the kernel (the collate drops it)
        src = torch.from_numpy(np.asarray(src, dtype=np.float32))
        idx = torch.from_numpy(np.asarray(idx, dtype=np.int64))
        sd = src.to(DEV).requires_grad_(True)
        out = ops.scatter_mean(sd, idx.to(DEV), n)
        sr = src.clone().requires_grad_(True)
        outr = O.scatter_mean(sr, idx, n)
        close(out, outr, F32, f"scatter_mean [{name}]", atol=1e-6, rtol=1e-6)
        empty = torch.ones(n, dtype=torch.bool)
        empty[idx] = False
        assert float(out.detach().cpu()[empty].abs().sum()) == 0.0, name
        gy = rnd(n, src.shape[1], seed=3)
        out.backward(gy.to(DEV)); outr.backward(gy)
        close(sd.grad, sr.grad, F32, f"scatter_mean grad [{name}]", atol=1e-6, rtol=1e-6)


def test_parents_from_coords_negative_and_duplicate_coordinates():
    """The coordinate map of the stride-2^k pooling pair: floor division for negative coordinates, voxels sharing a coarse
    cell share the parent, batch items never mix, a missing coarse voxel -> -1 (then skipped by the pooling kernel)."""
    fine = torch.tensor([[0, -1, -1, -1], [0, -2, -2, -2], [0, 0, 1, 1], [0, 1, 0, 0], [1, 0, 1, 1], [0, 3, 3, 3], [0, -3, 5, 0]])
    cc, par = O.pooling_transpose_parents(fine, 2)
    got = ops.parents_from_coords(fine.to(DEV), cc.to(DEV), 2)
    assert got.cpu().tolist() == par.tolist() == [0, 0, 1, 1, 2, 3, 4]
    # shuffled coarse rows: the map follows the rows; a removed coarse voxel orphans exactly its children
    perm = torch.tensor([3, 0, 4, 2, 1])
    got = ops.parents_from_coords(fine.to(DEV), cc[perm].to(DEV), 2)
    inv = torch.empty_like(perm); inv[perm] = torch.arange(5)
    assert got.cpu().tolist() == inv[par].tolist()
    got = ops.parents_from_coords(fine.to(DEV), cc[1:].to(DEV), 2)
    assert got.cpu().tolist() == [-1, -1, 0, 0, 1, 2, 3]


def test_mean_many_matches_torch_forward_and_backward():
    """ops.mean_many: sum_g mean(f_g(x_g)) with f in {identity, clamp(min), non-finite -> 0}, one launch each way."""
    xs = [rnd(800, 256, seed=1), rnd(4, 4096, 200, seed=2) * 40.0, rnd(4, 200, 201, seed=3), rnd(7, 3, seed=4), rnd(5, seed=5) * 100]
    xs[2][..., [0, 2]] = float("-inf")
    modes = ["plain", "clamp_min", "finite", "clamp_min", "plain"]
    a = [x.clone().to(DEV).requires_grad_(True) for x in xs]
    b = [x.clone().requires_grad_(True) for x in xs]
    la = ops.mean_many(a, modes, clamp_min=-50.0)
    f = {"plain": lambda t: t, "clamp_min": lambda t: t.clamp(min=-50.0), "finite": lambda t: torch.where(torch.isfinite(t), t, torch.zeros_like(t))}
    lb = sum(f[m](t).mean() for m, t in zip(modes, b))
    assert abs(float(la) - float(lb)) <= 1e-5 * max(1.0, abs(float(lb)))
    (la * 3.0).backward(); (lb * 3.0).backward()
    for t, r in zip(a, b):
        assert torch.allclose(t.grad.cpu(), r.grad, atol=1e-9, rtol=1e-5)
    l2 = ops.mean_many([t.detach() for t in a], modes, clamp_min=-50.0)      # deterministic
    assert float(l2) == float(la)


@pytest.mark.parametrize("Lq,Lk,mode,H,dh", [(100, 100, "bias", 8, 32), (100, 100, "kpm", 8, 32), (200, 200, "bias", 8, 32),
                                             (240, 240, "kpm", 2, 32), (16, 16, "kpm", 4, 32), (1, 37, "bias", 4, 32),
                                             (77, 5, "kpm", 4, 32), (130, 97, "bias", 3, 32),
                                             (120, 120, "bias", 12, 64), (80, 80, "bias", 12, 64), (128, 128, "kpm", 3, 64),
                                             (1, 37, "bias", 2, 64), (77, 5, "kpm", 2, 64), (97, 128, "bias", 2, 64)])
def test_attention_self_mfma_split_bf16_kernels(Lq, Lk, mode, H, dh):
    """Compute type BF16X3 (fp32 storage, split-bf16 MFMA arithmetic; csrc/attn_sa.hip: the decoder's self-attention at
    d_h = 32 and at d_h = 64, the shipped decoders' head width) against an fp64 reference and against the exact-fp32 kernels
    on the same inputs: outputs, lse-dependent gradients of q / k / v and the bias gradient, incl. padded keys, ragged sizes
    and a query count above 128."""
    B = 3
    d = H * dh
    q, k, v = rnd(B, Lq, d, seed=1), rnd(B, Lk, d, seed=2), rnd(B, Lk, d, seed=3)
    kpm = torch.arange(Lk)[None, :] >= torch.tensor([Lk, max(1, Lk // 2), max(1, Lk - 3)])[:, None]
    bias = torch.randn(B, H, Lq, Lk, generator=torch.Generator().manual_seed(5)) if mode == "bias" else None
    go = rnd(B, Lq, d, seed=7)
    res = {}
    for ct in (L.BF16X3, F32):
        qd, kd, vd = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
        bd = bias.to(DEV).requires_grad_(True) if bias is not None else None
        o = ops.attention(qd, kd, vd, H=H, ct=ct, kpm=kpm.to(DEV), bias=bd)
        o.backward(go.to(DEV))
        res[ct] = (o.detach(), qd.grad, kd.grad, vd.grad, bd.grad if bd is not None else None)
    qr, kr, vr = (t.double().requires_grad_(True) for t in (q, k, v))
    br = bias.double().requires_grad_(True) if bias is not None else None
    orf = attn_ref(qr, kr, vr, H, 1 / math.sqrt(dh), False, kpm, None, None, br)
    orf.backward(go.double())
    refs = (orf, qr.grad, kr.grad, vr.grad, br.grad if br is not None else None)
    for name, a, a_old, r in zip(("o", "dq", "dk", "dv", "dbias"), res[L.BF16X3], res[F32], refs):
        if r is None:
            continue
        close(a, r, F32, name + " split-bf16 vs fp64", atol=5e-5, rtol=5e-5)
        close(a, a_old, F32, name + " split-bf16 vs exact fp32", atol=5e-5, rtol=5e-5)


@pytest.mark.parametrize("R,M,N,G,adt,bdt,b2,cs", [(800, 256, 256, 12, "f32", "f32", True, True), (800, 256, 2048, 4, "f32", "f32", False, True),
                                                   (800, 2048, 256, 4, "bf16", "f32", False, True), (800, 256, 256, 3, "bf16", "bf16", False, False),
                                                   (8192, 256, 256, 3, "f32", "f32", False, True), (37, 72, 40, 2, "f32", "bf16", False, True),
                                                   (8992, 256, 256, 1, "f32", "f32", False, True)])
def test_gemm_weight_gradient_whole_k_chunks(R, M, N, G, adt, bdt, b2, cs):
    """dW = g^T (x [+ x2]) (+ fused bias gradient) through the 256-row-chunk kernel (gemm_wktt.hip) against the 64x64-tile
    pipeline kernel's arithmetic (the comparison is against torch in bf16-operand arithmetic
    and against an exact fp32 product within the bf16 bound), accumulating onto existing contents."""
    td = lambda n: torch.bfloat16 if n == "bf16" else torch.float32
    g_ = [rnd(R, M, seed=g).to(DEV).to(td(adt)) for g in range(G)]
    x_ = [rnd(R, N, seed=50 + g).to(DEV).to(td(bdt)) for g in range(G)]
    x2 = [rnd(R, N, seed=90 + g).to(DEV) if g % 2 == 0 else None for g in range(G)] if b2 else None
    base = rnd(G, M, N, seed=7).to(DEV)
    C_ = base.clone()
    csum = torch.zeros(G, M, device=DEV) if cs else None
    L.gemm(M=M, N=N, K=R, A=g_, B=x_, B2=x2, Cs=[C_[g] for g in range(G)], ct=BF16, lda=M, ldb=N, ldc=N, transA=True, transB=True,
           splitk=4, accumulate=True, colsum=[csum[g] for g in range(G)] if cs else None)
    for g in range(G):
        xs = x_[g].float() + (x2[g] if (x2 is not None and x2[g] is not None) else 0)
        ref = base[g] + g_[g].bfloat16().float().T @ xs.bfloat16().float()
        err = float((C_[g] - ref).abs().max()) / max(1.0, float(ref.abs().max()))
        assert err <= 2e-5 * (R ** 0.5), (g, err)          # fp32 accumulation order only (operands rounded identically)
        if cs:
            rs = g_[g].bfloat16().float().sum(0)
            assert float((csum[g] - rs).abs().max()) <= 1e-4 * max(1.0, float(rs.abs().max())) * (R ** 0.5) / 10


def test_entry_points_launch_on_the_device_of_their_stream_not_the_current_one():
    """the library takes the device from its inputs.  With the CURRENT device deliberately set elsewhere, a call
    on a stream (or, with the NULL stream, on pointers) of another device must run there and leave the caller's current
    device untouched.  Needs two visible GPUs (skips on the single-GPU test boxes)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 visible GPUs")
    import ctypes as C_
    d1 = torch.device("cuda", 1)
    dst, sc = torch.zeros(1000, device=d1), torch.full((1,), 3.0, device=d1)
    s1 = torch.cuda.Stream(device=d1)
    torch.cuda.set_device(0)
    L.check(L.lib().pq3d_fill_scaled(L.ptr(dst), 1000, L.ptr(sc), 2.0, C_.c_void_p(s1.cuda_stream)), "fill (stream of device 1)")
    s1.synchronize()
    assert torch.cuda.current_device() == 0 and float(dst.sum()) == 6000.0
    dst.zero_()
    torch.cuda.synchronize(d1)
    L.check(L.lib().pq3d_fill_scaled(L.ptr(dst), 1000, L.ptr(sc), 1.0, C_.c_void_p(0)), "fill (NULL stream, pointer on device 1)")
    torch.cuda.synchronize(d1)
    assert torch.cuda.current_device() == 0 and float(dst.sum()) == 3000.0


@pytest.mark.gpu
def test_launch_merges_pair_kernels_and_summed_layernorm_gradient():
    """fourier_pair == two fourier launches; sum_pair == two sum_n launches with adjacent outputs (and _SplitRows.backward
    returns their buffer instead of a copy); the LayerNorm backward with the upstream gradient in three addends ==
    the same call on their sum; a grouped '+ aux' product whose other groups have no aux."""
    from pq3d_amd import fused
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(5)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    B, Na, Nb, half = 3, 40, 77, 128
    xa, xb = rnd(B, Na, 6), rnd(B, Nb, 3)
    cmin, cmax, G = rnd(B, 3) - 3, rnd(B, 3) + 3, rnd(3, half)
    pair = ops.fourier_pair(xa[:, :, :3], xb, cmin, cmax, G)
    assert torch.equal(pair[:B * Na].view(B, Na, -1), ops.fourier(xa[:, :, :3], cmin, cmax, G))
    assert torch.equal(pair[B * Na:].view(B, Nb, -1), ops.fourier(xb, cmin, cmax, G))
    # sum_pair
    pa, pb = [rnd(B, Na, 256) for _ in range(5)], [rnd(B, Nb, 256) for _ in range(3)]
    sa, sb = ops.sum_pair(pa, pb)
    assert torch.equal(sa, ops.sum_n(pa)) and torch.equal(sb, ops.sum_n(pb))
    y = rnd(B * (Na + Nb), 256).requires_grad_()
    ya, yb = ops.split_rows(y, B * Na)
    (gy,) = torch.autograd.grad([ya, yb], [y], [sa.view(B * Na, 256), sb.view(B * Nb, 256)])
    assert gy.data_ptr() == sa.data_ptr() and torch.equal(gy, torch.cat([sa.view(-1, 256), sb.view(-1, 256)], 0))
    (gy2,) = torch.autograd.grad(ops.split_rows(y, B * Na), [y], [sa.view(B * Na, 256).clone(), sb.view(B * Nb, 256).clone()])
    assert torch.equal(gy2, gy)
    # LayerNorm backward, upstream gradient in three addends (one and four branches)
    for M in (1, 4):
        R, d = 800, 256
        x, os_ = rnd(8, 100, d), [rnd(8, 100, d) for _ in range(M)]
        gam, bet = [rnd(d) for _ in range(M)], [rnd(d) for _ in range(M)]
        coef = torch.softmax(rnd(M, 8), 0).contiguous() if M > 1 else None
        ys = torch.empty(8, 100, d, device=dev)
        mean, rstd = torch.empty(M, R, device=dev), torch.empty(M, R, device=dev)
        dd = ops._ln_desc(x, os_, gam, bet, coef, 1e-5, 100, ys, mean, rstd, None)
        L.check(L.lib().pq3d_add_ln_fwd(C.byref(dd), L.stream()), "fwd")
        dys = [rnd(8, 100, d) for _ in range(3)]
        outs = []
        for dy in (dys, (dys[0] + dys[1]) + dys[2]):
            dgs, dbs = [torch.zeros(d, device=dev) for _ in range(M)], [torch.zeros(d, device=dev) for _ in range(M)]
            dx, d_o = fused._ln_bwd(x, os_, gam, bet, 1e-5, coef, 100, mean, rstd, dy, dgs, dbs)
            outs.append((dx, d_o, torch.stack(dgs), torch.stack(dbs)))
        for a, b in zip(*outs):   # atomics order only (the addends are summed identically): column sums over 800 rows
            torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4)
    # grouped "+ aux" with an aux pointer in the last group only
    A = [rnd(800, 256) for _ in range(3)]
    W = [rnd(256, 256) * 0.05 for _ in range(3)]
    aux = rnd(800, 256)
    out = torch.empty(3, 800, 256, device=dev)
    L.gemm(M=800, N=256, K=256, A=A, B=W, Cs=[out[0], out[1], out[2]], aux=[None, None, aux], act_grad="add", ct=F32, lda=256,
           ldb=256, ldc=256, transB=True)
    for i in range(3):
        ref = A[i].double() @ W[i].double() + (aux.double() if i == 2 else 0)
        torch.testing.assert_close(out[i].double(), ref, rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("Lq,H,mode", [(100, 8, "bias"), (100, 8, "kpm"), (77, 4, "kpm"), (130, 2, "bias"), (16, 8, "kpm")])
def test_attention_self_backward_with_folded_out_projection(Lq, H, mode):
    """pq3d_attn_proj DOUT: dO = g W formed inside the split-bf16 self-attention backward equals the separate
    single-bf16 product followed by the plain backward call (same arithmetic: bf16 operands, fp32 accumulate)."""
    from pq3d_amd import fused
    dh, B = 32, 3
    d = H * dh
    q, k, v = (rnd(B, Lq, d, seed=s).to(DEV) for s in (1, 2, 3))
    kpm = (torch.arange(Lq)[None, :] >= torch.tensor([Lq, max(1, Lq // 2), max(1, Lq - 3)])[:, None]).to(DEV)
    bias = torch.randn(B, H, Lq, Lq, generator=torch.Generator().manual_seed(5)).to(DEV) if mode == "bias" else None
    g, W = rnd(B, Lq, d, seed=7).to(DEV), (rnd(d, d, seed=8) * 0.06).to(DEV)
    o, lse = torch.empty_like(q), torch.empty(B, H, Lq, device=DEV)
    fused._attn(q, k, v, o, lse, H, L.BF16X3, False, kpm=kpm, bias=bias)
    assert fused.sa_fold_ok(BF16, B, H, Lq, d, None, g, W)
    outs = []
    for fold in (False, True):
        dqkv = torch.empty(3, B, Lq, d, device=DEV)
        delta = torch.empty(B, H, Lq, device=DEV)
        dsb = torch.empty_like(bias) if bias is not None else None
        do = None
        if not fold:
            do = torch.empty(B, Lq, d, device=DEV)
            L.gemm(M=B * Lq, N=d, K=d, A=[g], B=[W], Cs=[do], ct=BF16, lda=d, ldb=d, ldc=d, transB=True)
        fused._attn(q, k, v, o, lse, H, L.BF16X3, False, kpm=kpm, bias=bias, bwd=(do, dqkv[0], dqkv[1], dqkv[2], delta, dsb),
                    proj_dout=(g, W) if fold else None)
        outs.append((dqkv, delta, dsb))
    for name, a, b in zip(("dqkv", "delta", "dbias"), outs[1], outs[0]):
        if a is not None:
            torch.testing.assert_close(a, b, rtol=2e-5, atol=2e-6 * float(b.abs().max()), msg=lambda m: f"{name}: {m}")
    # a call the split-bf16 kernels do not take is refused, loudly, without a launch
    dqkv = torch.empty(3, B, Lq, d, device=DEV)
    with pytest.raises(L.Pq3dError, match="proj"):
        fused._attn(q, k, v, o, lse, H, F32, False, kpm=kpm, bias=bias,
                    bwd=(None, dqkv[0], dqkv[1], dqkv[2], torch.empty(B, H, Lq, device=DEV), None), proj_dout=(g, W))


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(800, 256, 201), (800, 256, 13), (130, 64, 77)])
def test_gemm_input_gradient_over_an_unaligned_reduction(M, N, K):
    """dx = g W with g [M, K] of an unaligned row length (the class head: 201 logits) and W [K, N] read transposed: served by
    the whole-K kernel with element loads of the A rows (was the generic slow kernel, 22 us at config 4); bf16 operands."""
    g, w = rnd(M, K, seed=3).to(DEV), (rnd(K, N, seed=4) * 0.1).to(DEV)
    aux = rnd(M, N, seed=5).to(DEV)
    for with_aux in (False, True):
        out = torch.empty(M, N, device=DEV)
        L.gemm(M=M, N=N, K=K, A=[g], B=[w], Cs=[out], aux=[aux] if with_aux else None, act_grad="add" if with_aux else None,
               ct=BF16, lda=K, ldb=N, ldc=N, transB=True)
        ref = g.bfloat16().double() @ w.bfloat16().double() + (aux.double() if with_aux else 0)
        torch.testing.assert_close(out.double(), ref, rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("ct", [F32, BF16])
def test_linear_with_dropout_and_residual_in_the_epilogue_equals_the_separate_ops(ct):
    """x + dropout(o W^T) and dropout(relu(h W^T)) with the dropout (and the residual add) riding the projection's epilogue
    (the caption body, pq3d_amd/t5.py) draw the SAME mask as the stand-alone dropout op on the same site and give the same
    outputs and gradients -- incl. the ReLU case whose backward takes the keep mask from the saved output."""
    R, K, N, p = 96, 128, 64, 0.3
    seed = torch.tensor([0x5eed1234], dtype=torch.int64, device=DEV)
    for act, with_res in ((None, True), ("relu", False), (None, False)):
        res = {}
        for fused in (True, False):
            o = rnd(3, 32, K, seed=1).to(DEV).requires_grad_()
            w = (rnd(N, K, seed=2) * 0.2).to(DEV).requires_grad_()
            x = rnd(3, 32, N, seed=3).to(DEV).requires_grad_()
            d = L.Drop(p, 77, seed)
            if fused:
                y = ops.linear(o, w, None, ct=ct, act=act, drop=d, residual=x if with_res else None)
            else:
                y = ops.dropout(ops.linear(o, w, None, ct=ct, act=act), d)
                if with_res:
                    y = x + y
            (y * rnd(3, 32, N, seed=4).to(DEV)).sum().backward()
            res[fused] = (y.detach(), o.grad, w.grad, x.grad if with_res else None)
        for name, a, b in zip(("y", "do", "dw", "dx"), res[True], res[False]):
            if a is None:
                continue
            assert (a == 0).eq(b == 0).all() or name != "y", "different dropout masks"
            tol = 1e-5 if ct == F32 else 2e-2
            torch.testing.assert_close(a, b, rtol=tol, atol=tol * float(b.abs().max()), msg=lambda m: f"{act} {with_res} {name}: {m}")
    assert R  # (shape constants above are for the reader)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["x3_nt", "f32_nt", "bf16_nn_dx", "nt128", "tt128"])
def test_gemm_xcd_aware_tile_order_of_big_launches(kind):
    """Launches of >= 2048 workgroups (and 128-tile weight gradients with >= 9 tiles per k-split) walk their tiles in the
    XCD-aware order (csrc/common.h tile_index: every XCD gets a contiguous run of logical tiles, sharers adjacent).  The
    order is a permutation of the tiles: every output element must be written exactly once, also when the workgroup count
    is not a multiple of 8 and the last row tile is ragged.  Compared with float64 products of the same operands."""
    if kind in ("x3_nt", "f32_nt"):
        M, N, K, G = 10300, 768, 96, 3      # 161 x 12 x 3 = 5796 workgroups (= 4 mod 8), ragged last row tile
        A = [rnd(M, K, seed=0).to(DEV)] * G if kind == "x3_nt" else [rnd(M, K, seed=g).to(DEV) for g in range(G)]   # q / k / v of one input
        W = [(rnd(N, K, seed=10 + g) * 0.1).to(DEV) for g in range(G)]
        b = [rnd(N, seed=20 + g).to(DEV) for g in range(G)]
        C = torch.full((G, M, N), 7.0, device=DEV)
        L.gemm(M=M, N=N, K=K, A=A, B=W, bias=b, Cs=[C[g] for g in range(G)], ct=L.BF16X3 if kind == "x3_nt" else F32,
               lda=K, ldb=K, ldc=N)
        for g in range(G):
            ref = (A[g].double() @ W[g].double().T + b[g].double())
            assert float((C[g].double() - ref).abs().max()) / float(ref.abs().max()) < (1e-5 if kind == "x3_nt" else 3e-6), g
    elif kind == "bf16_nn_dx":
        M, N, K, G = 10300, 768, 128, 3     # dX = dY W (B not transposed in memory: transB)
        dY = [rnd(M, K, seed=g).to(DEV) for g in range(G)]
        W = [(rnd(K, N, seed=10 + g) * 0.1).to(DEV) for g in range(G)]
        C = torch.full((G, M, N), 7.0, device=DEV)
        L.gemm(M=M, N=N, K=K, A=dY, B=W, Cs=[C[g] for g in range(G)], ct=BF16, lda=K, ldb=N, ldc=N, transB=True)
        for g in range(G):
            ref = dY[g].bfloat16().double() @ W[g].bfloat16().double()
            assert float((C[g].double() - ref).abs().max()) / float(ref.abs().max()) < 2e-5, g
    elif kind == "nt128":
        M, N, K, G = 8200, 768, 128, 6      # 65 x 6 x 6 = 2340 tiles of 128 x 128 (= 4 mod 8)
        A2 = [rnd(M, K, seed=g).to(DEV).bfloat16() for g in range(2)]
        A = [A2[g % 2] for g in range(G)]       # 3 groups per row operand, interleaved: sorted into runs, each walked as one plane
        W = [(rnd(N, K, seed=100 + g) * 0.1).to(DEV) for g in range(G)]
        b = [rnd(N, seed=200 + g).to(DEV) for g in range(G)]
        C_new = torch.zeros(G, M, N, dtype=torch.bfloat16, device=DEV)
        C_old = torch.zeros_like(C_new)
        L.gemm(M=M, N=N, K=K, A=A, B=[w.bfloat16() for w in W], bias=b, Cs=[C_new[g] for g in range(G)], ct=BF16, lda=K, ldb=K, ldc=N)
        L.gemm(M=M, N=N, K=K, A=A, B=W, bias=b, Cs=[C_old[g] for g in range(G)], ct=BF16, lda=K, ldb=K, ldc=N)   # 64 x 64 tiles
        assert torch.equal(C_new, C_old)
        for g in (0, G - 1):
            close(C_new[g].float(), A[g].float() @ W[g].bfloat16().float().T + b[g], BF16, "gemm128 xcd order")
    else:
        R, M, N, G = 4096, 768, 768, 4      # 6 x 6 tiles per (group, k-split); pairs of groups share the column operand
        gs = [rnd(R, M, seed=g).to(DEV).bfloat16() for g in range(G)]
        x2 = [rnd(R, N, seed=50 + g).to(DEV).bfloat16() for g in range(2)]
        xs = [x2[g // 2] for g in range(G)]
        dW = torch.full((G, M, N), 123.0, device=DEV)
        cb = torch.full((G, M), -5.0, device=DEV)
        L.gemm(M=M, N=N, K=R, A=gs, B=xs, Cs=[dW[g] for g in range(G)], ct=BF16, lda=M, ldb=N, ldc=N, transA=True, transB=True,
               splitk=2, colsum=[cb[g] for g in range(G)])
        for g in range(G):
            ref = gs[g].double().T @ xs[g].double()
            assert float((dW[g].double() - ref).abs().max()) / float(ref.abs().max()) < 2e-5, g
            refb = gs[g].double().sum(0)
            assert float((cb[g].double() - refb).abs().max()) / float(refb.abs().max()) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("a2,act,shared", [(False, None, True), (True, None, False), (False, "gelu", False), (True, "relu", True)])
def test_gemm_split_bf16_128_tile_matches_the_64_tile_kernels_bit_for_bit(a2, act, shared):
    """Split-bf16 NT products with >= 512 tiles of 128 x 128 take gemm_cv128.hip (the shipped stage-2 decoder's projections and
    FFN at M = 10240): same hi / lo split, same k order, same three-term order per accumulator as the 64 x 64 and whole-K
    kernels -> identical bits to the same product launched in 1024-row pieces (which stay on those kernels); also against
    float64.  Ragged last row tile, optional addend (x + pos), bias, activation, second (pre-activation) output."""
    M, N, K, G = 10300, 768, 256, 3
    A = [rnd(M, K, seed=0).to(DEV)] * G if shared else [rnd(M, K, seed=g).to(DEV) for g in range(G)]
    P = [rnd(M, K, seed=30).to(DEV)] * G if a2 else None
    W = [(rnd(N, K, seed=10 + g) * 0.1).to(DEV) for g in range(G)]
    b = [rnd(N, seed=20 + g).to(DEV) for g in range(G)]
    C = torch.full((G, M, N), 7.0, device=DEV)
    C2 = torch.full((G, M, N), 7.0, device=DEV) if act else None
    Cp = torch.full((G, M, N), 7.0, device=DEV)
    Cp2 = torch.full((G, M, N), 7.0, device=DEV) if act else None
    L.gemm(M=M, N=N, K=K, A=A, A2=P, B=W, bias=b, Cs=[C[g] for g in range(G)], C2=[C2[g] for g in range(G)] if act else None,
           ct=L.BF16X3, lda=K, ldb=K, ldc=N, act=act)
    for r0 in range(0, M, 1024):
        r1 = min(M, r0 + 1024)
        L.gemm(M=r1 - r0, N=N, K=K, A=[a[r0:r1] for a in A], A2=[p[r0:r1] for p in P] if a2 else None, B=W, bias=b,
               Cs=[Cp[g, r0:r1] for g in range(G)], C2=[Cp2[g, r0:r1] for g in range(G)] if act else None, ct=L.BF16X3,
               lda=K, ldb=K, ldc=N, act=act)
    assert torch.equal(C, Cp)
    if act:
        assert torch.equal(C2, Cp2)
    for g in range(G):
        x = A[g].double() + (P[g].double() if a2 else 0)
        pre = x @ W[g].double().T + b[g].double()
        ref = pre if act is None else (torch.relu(pre) if act == "relu" else torch.nn.functional.gelu(pre))
        assert float((C[g].double() - ref).abs().max()) / float(ref.abs().max()) < 1e-5, g
        if act:
            assert float((C2[g].double() - pre).abs().max()) / float(pre.abs().max()) < 1e-5, g


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["nt_f32", "nn_f32_gelu_grad", "nn_bf16_add_c2", "nn_kconcat", "nt_bf16_relu"])
def test_gemm_converting_128_tile_matches_the_64_tile_kernels_bit_for_bit(kind):
    """bf16-compute products with fp32-stored operands and >= 256 / 512 tiles of 128 x 128 take gemm_cv128.hip (input
    gradients dX = dY W and the K-concatenated sums of the shipped stage-2 decoder at M = 10240): same rounding points and
    k order as the 64 x 64 kernels -> identical bits to the same product launched in 1024-row pieces, with the fused
    epilogue options the decoder's backward uses (activation gradient, "+ aux", second output, dropout-free)."""
    M, N, K = 10300, 768, 768
    nn = kind.startswith("nn")
    G = 3
    kc = 3 if kind == "nn_kconcat" else 0
    adt = torch.bfloat16 if "bf16" in kind else torch.float32
    A = [rnd(M, K, seed=g).to(DEV).to(adt) for g in range(G)]
    W = [(rnd(K, N, seed=10 + g) * 0.1).to(DEV) for g in range(G)] if nn else [(rnd(N, K, seed=10 + g) * 0.1).to(DEV) for g in range(G)]
    outs = 1 if kc else G
    aux = [rnd(M, N, seed=40 + o).to(DEV) for o in range(outs)] if kind != "nt_f32" else None
    bias = [rnd(N, seed=20 + g).to(DEV) for g in range(G)] if not nn else None
    kw = dict(ct=BF16, lda=K, ldb=N if nn else K, ldc=N, transB=nn, kconcat=kc)
    if kind == "nn_f32_gelu_grad":
        kw.update(act_grad="gelu")
    elif kind in ("nn_bf16_add_c2", "nn_kconcat"):
        kw.update(act_grad="add")
    elif kind == "nt_bf16_relu":
        kw.update(act="relu")
        aux = None
    pad = lambda ts: [t for o in range(outs) for t in [ts[o]] + [None] * (kc - 1)] if kc else ts

    def run(r0, r1, C, C2):
        L.gemm(M=r1 - r0, N=N, K=K, A=[a[r0:r1] for a in A], B=W, bias=bias,
               Cs=pad([C[o, r0:r1] for o in range(outs)]), C2=pad([C2[o, r0:r1] for o in range(outs)]) if C2 is not None else None,
               aux=pad([x[r0:r1] for x in aux]) if aux is not None else None, **kw)
    C = torch.full((outs, M, N), 7.0, device=DEV)
    Cp = torch.full((outs, M, N), 7.0, device=DEV)
    C2 = torch.full((outs, M, N), 7.0, device=DEV) if kind == "nn_bf16_add_c2" else None
    Cp2 = torch.full((outs, M, N), 7.0, device=DEV) if C2 is not None else None
    run(0, M, C, C2)
    for r0 in range(0, M, 1024):
        run(r0, min(M, r0 + 1024), Cp, Cp2)
    assert torch.equal(C, Cp)
    if C2 is not None:
        assert torch.equal(C2, Cp2)
    for o in range(outs):
        gs = range(o * kc, (o + 1) * kc) if kc else [o]
        pre = sum(A[g].bfloat16().double() @ (W[g].bfloat16().double() if nn else W[g].bfloat16().double().T) for g in gs)
        if bias is not None:
            pre = pre + bias[o].double()
        if C2 is not None:
            assert float((C2[o].double() - pre).abs().max()) / float(pre.abs().max()) < 2e-5
        if kind == "nn_f32_gelu_grad":
            x = aux[o].double()
            ref = pre * (0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-x * x / 2) / (2 * torch.pi) ** 0.5)
        elif kw.get("act_grad") == "add":
            ref = pre + aux[o].double()
        elif kw.get("act") == "relu":
            ref = torch.relu(pre)
        else:
            ref = pre
        assert float((C[o].double() - ref).abs().max()) / float(ref.abs().max()) < 2e-5, o


@pytest.mark.gpu
@pytest.mark.parametrize("acc", [False, True])
def test_gemm_split_k_request_on_a_big_launch_is_one_deterministic_pass(acc):
    """A split-K input-gradient product with enough 128 x 128 tiles (the FFN's dx += dh W1 at M = 10240) is served by
    gemm_cv128.hip as one pass -- C = / += product through the epilogue, no atomics: two runs give identical bits."""
    M, N, K = 10240, 768, 2048
    A = rnd(M, K, seed=1).to(DEV)
    W = (rnd(K, N, seed=2) * 0.05).to(DEV)
    base = rnd(M, N, seed=3).to(DEV)
    outs = []
    for _ in range(2):
        C = base.clone() if acc else torch.full((M, N), 9.0, device=DEV)
        L.gemm(M=M, N=N, K=K, A=[A], B=[W], Cs=[C], ct=BF16, lda=K, ldb=N, ldc=N, transB=True, splitk=4, accumulate=acc)
        outs.append(C)
    assert torch.equal(outs[0], outs[1])
    ref = A.bfloat16().double() @ W.bfloat16().double() + (base.double() if acc else 0)
    assert float((outs[0].double() - ref).abs().max()) / float(ref.abs().max()) < 2e-5


# ---------------------------------------------------------------------------------------------- 3-D mask as bits
@pytest.mark.parametrize("Lk", [4096, 1056, 100])
def test_mask_pack_matches_row_all_and_bytes(Lk):
    g = torch.Generator().manual_seed(Lk)
    m = torch.rand(3, 37, Lk, generator=g) < 0.5
    m[1, 4] = True                      # an all-masked row: opened (bits cleared)
    m[2, 9] = False
    ro, bits = ops.mask_pack(m.to(DEV))
    assert torch.equal(ro.cpu(), m.all(-1))
    W = (Lk + 31) // 32
    mm = m & ~m.all(-1, keepdim=True)
    pad = torch.zeros(3, 37, W * 32, dtype=torch.bool)
    pad[..., :Lk] = mm
    want = (pad.view(3, 37, W, 32).long() << torch.arange(32)).sum(-1)
    want = torch.where(want >= 2 ** 31, want - 2 ** 32, want).int()
    assert torch.equal(bits.cpu(), want)


@pytest.mark.parametrize("B,Lq,Lk", [(6, 200, 4096), (3, 100, 1024), (3, 57, 544)])
def test_resident_backward_mask_bits_equal_mask_bytes(B, Lq, Lk):
    """The all-queries-resident cross-attention backward with the 3-D self-mask as bit words (pq3d_mask_pack, row_open folded
    in) gives the very gradients of the byte-mask path (memories stacked along the batch share the mask: mask_bmod)."""
    from pq3d_amd import fused as F
    H, d = 8, 256
    g = torch.Generator().manual_seed(B + Lq)
    q = torch.randn(B, Lq, d, generator=g).to(DEV).bfloat16()
    k = torch.randn(B, Lk, d, generator=g).to(DEV).bfloat16()
    v = torch.randn(B, Lk, d, generator=g).to(DEV).bfloat16()
    nb = B // 3
    m = (torch.rand(nb, Lq, Lk, generator=g) < 0.6)
    m[0, 3] = True
    m = m.to(DEV)
    ro, bits = ops.mask_pack(m)
    o = torch.empty_like(q); lse = torch.empty(B, H, Lq, device=DEV)
    F._attn(q, k, v, o, lse, H, L.BF16, True, mask=m, row_open=ro, mask_bmod=nb)
    o2, lse2 = torch.empty_like(o), torch.empty_like(lse)      # the streaming forward reads the same bits: identical outputs
    F._attn(q, k, v, o2, lse2, H, L.BF16, True, mask=m, row_open=ro, mask_bmod=nb, mask_bits=bits)
    assert torch.equal(o, o2) and torch.equal(lse, lse2)
    do = torch.randn(B, Lq, d, generator=g).to(DEV).bfloat16()
    res = []
    for mb in (None, bits):
        dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
        delta = torch.empty_like(lse)
        F._attn(q, k, v, o, lse, H, L.BF16, True, mask=m, row_open=ro, mask_bmod=nb, mask_bits=mb,
                bwd=(do, dq, dk, dv, delta, None))
        res.append((dq, dk, dv))
    for a, b_, name in zip(res[0], res[1], ("dq", "dk", "dv")):
        if name == "dq":     # summed over the waves in a fixed order, over key slices by the combine kernel: deterministic too
            assert float((a.float() - b_.float()).abs().max()) <= 1e-2 * float(a.float().abs().max()), name
        else:
            assert torch.equal(a, b_), name
    assert float(res[1][0].float().abs().max()) > 0 and torch.isfinite(res[1][1].float()).all()


def test_gemm_tt_multi_heterogeneous_weight_gradients_in_one_launch():
    """pq3d_gemm_tt_multi: weight (+ bias) gradients of different shapes and operand dtypes in one launch -- against fp64 on the
    bf16-rounded operands, accumulating onto running slots, duplicated outputs (a weight shared by two layer applications)."""
    g = torch.Generator().manual_seed(3)
    R = 800
    specs = [(256, 256, torch.float32, torch.float32, True, True), (2048, 256, torch.float32, torch.float32, False, True),
             (256, 2048, torch.float32, torch.bfloat16, False, True), (256, 256, torch.bfloat16, torch.float32, True, False),
             (200, 264, torch.float32, torch.float32, False, True), (8, 8, torch.bfloat16, torch.bfloat16, False, False)]
    probs, refs = [], []
    for N, K, dg, dx, has2, hasb in specs:
        gt = torch.randn(R, N, generator=g).to(DEV).to(dg)
        xt = torch.randn(R, K, generator=g).to(DEV).to(dx)
        x2 = torch.randn(R, K, generator=g).to(DEV) if has2 else None
        dw = torch.randn(N, K, generator=g).to(DEV)
        db = torch.randn(N, generator=g).to(DEV) if hasb else None
        gb = gt.float().bfloat16().double()
        xb = (xt.float() + (x2 if has2 else 0)).bfloat16().double()
        refs.append((dw.double() + gb.t() @ xb, (db.double() + gb.sum(0)) if hasb else None))
        probs.append((gt, xt, x2, dw, db))
    # the first weight once more (shared across num_blocks): its slot takes both contributions
    gt, xt, x2, dw, db = probs[0]
    probs.append((gt, xt, x2, dw, db))
    gb = gt.float().bfloat16().double(); xb = (xt + x2).bfloat16().double()
    refs[0] = (refs[0][0] + gb.t() @ xb, refs[0][1] + gb.sum(0))
    assert all(ops.tt_multi_ok(*p, p[3].shape[0], p[3].shape[1], R) for p in probs)
    ops.tt_multi(probs)
    for (gt, xt, x2, dw, db), (rw, rb) in zip(probs[:len(specs)], refs):
        assert float((dw.double() - rw).abs().max()) <= 2e-4 * float(rw.abs().max()), dw.shape
        if db is not None:
            assert float((db.double() - rb).abs().max()) <= 2e-4 * float(rb.abs().max())
    # not eligible: an odd width; long reductions over many tiles stay on the 128 x 128-tile kernel
    assert ops.dw_long_path(768, 768, 10240, 16, BF16) and not ops.dw_long_path(256, 256, 8192, 3, BF16)
    assert not ops.tt_multi_ok(torch.zeros(100, 12, device=DEV), torch.zeros(100, 20, device=DEV), None,
                               torch.zeros(12, 20, device=DEV), None, 12, 20, 100)


@pytest.mark.gpu
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
def test_scale_rows_many_equals_per_tensor(out_dtype):
    """pq3d_scale_rows_grouped: several same-shape tensors sharing scales / flags in one launch -- the same bits as one launch each."""
    from pq3d_amd import ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(3)
    R, C_ = 4 * 333, 200
    xs = [torch.randn(4, 333, C_, generator=g).to(dev) for _ in range(5)]
    xs[1][0, 5, 3] = float("nan")
    scale = torch.rand(R, generator=g).to(dev)
    zf = (torch.rand(R, generator=g) < 0.2).to(dev)
    zf[5] = True   # the row with the NaN is dropped: exactly zero
    ref = [ops.scale_rows(x, R, out_dtype, scale=scale, zero_flag=zf) for x in xs]
    out = ops.scale_rows_many(xs, R, out_dtype, scale=scale, zero_flag=zf)
    it = torch.int16 if out_dtype == torch.bfloat16 else torch.int32
    for a, b in zip(out, ref):
        assert a.shape == b.shape and torch.equal(a.view(it), b.view(it))


@pytest.mark.parametrize("R", [300, 1024])
def test_linear_ln_group_backward_tt_multi_path(R):
    """below 2048 rows the grouped Linear+LayerNorm backward sends its weight / bias gradients through the one-launch
    pq3d_gemm_tt_multi (fused bias column sums) next to the LayerNorm parameter-gradient atomics -- against float64 autograd."""
    G, K, N = 3, 256, 256
    xs = [rnd(R, K, seed=40 + g) for g in range(G)]
    Ws = [rnd(N, K, seed=50 + g, scale=0.06) for g in range(G)]
    bs = [rnd(N, seed=60 + g, scale=0.1) for g in range(G)]
    gs = [1 + rnd(N, seed=70 + g, scale=0.1) for g in range(G)]
    be = [rnd(N, seed=80 + g, scale=0.1) for g in range(G)]
    gy = [rnd(R, N, seed=90 + g) for g in range(G)]
    dev = lambda ts, rg=True: [t.to(DEV).requires_grad_(rg) for t in ts]
    xd, Wd, bd, gd, bed = dev(xs), dev(Ws), dev(bs), dev(gs), dev(be)
    assert ops.tt_multi_ok(torch.empty(R, N, device=DEV), xd[0].detach(), None, torch.empty(N, K, device=DEV), torch.empty(N, device=DEV), N, K, R)
    ys = ops.linear_ln_group(xd, Wd, bd, gd, bed, ct=BF16)
    torch.autograd.backward(list(ys), [g.to(DEV) for g in gy])
    for g in range(G):
        ref = [t.double().requires_grad_(True) for t in (xs[g].bfloat16().float(), Ws[g].bfloat16().float(), bs[g], gs[g], be[g])]
        y = torch.nn.functional.layer_norm(ref[0] @ ref[1].t() + ref[2], (N,), ref[3], ref[4], 1e-5)
        y.backward(gy[g].double())
        close(ys[g], y, BF16, f"y{g}")
        for name, a, r in zip(("dx", "dW", "db", "dgamma", "dbeta"), (xd[g], Wd[g], bd[g], gd[g], bed[g]), ref):
            close(a.grad, r.grad, BF16, f"{name}{g}")


def test_row_ce_loss_can_be_modified_in_place():
    """the scalar returned by cross_entropy_rows must not share a version counter with the tensors saved for its backward."""
    from pq3d_amd.losses import cross_entropy_rows
    x = rnd(64, 50, seed=5).to(DEV).requires_grad_(True)
    t = torch.randint(0, 50, (64,), generator=torch.Generator().manual_seed(3)).to(DEV)
    loss = cross_entropy_rows(x, t)
    loss /= 4.0
    loss += 1.0
    loss.backward()
    xr = x.detach().double().cpu().requires_grad_(True)
    (torch.nn.functional.cross_entropy(xr, t.cpu()) / 4.0 + 1.0).backward()
    close(x.grad, xr.grad, F32, "dlogits")


def test_sa_backward_phase_split_keeps_the_bits(tmp_path):
    """csrc/attn_sa.hip, round 6: with few (scene, head) units the backward's phase A (dQ, dbias) and phase B (dK, dV) -- and halves
    of each -- run on separate workgroups (grid.z = 2 / 4).  Every output element keeps its one writer and its arithmetic: the
    gradients are bit for bit those of the one-workgroup form (PQ3D_SA_BWD_SPLIT = 0 / 1 / 2 read once per process -> subprocesses)."""
    import subprocess
    import sys
    code = r'''
import sys, torch
sys.path.insert(0, %r)
from pq3d_amd import ops, _lib as L
torch.manual_seed(5)
out = {}
for name, (B, H, Lq) in {"c2": (8, 8, 100), "c4": (4, 8, 200), "odd": (3, 8, 77)}.items():
    d = 32 * H
    q, k, v = [torch.randn(B, Lq, d, device="cuda", requires_grad=True) for _ in range(3)]
    bias = torch.randn(B, H, Lq, Lq, device="cuda", requires_grad=True)
    kpm = torch.rand(B, Lq, device="cuda") < 0.1
    kpm[:, 0] = False
    o = ops.attention(q, k, v, H=H, ct=L.BF16X3, kpm=kpm, bias=bias)
    g = torch.randn_like(o)
    o.backward(g)
    out[name] = [t.detach().cpu() for t in (o, q.grad, k.grad, v.grad, bias.grad)]
torch.save(out, sys.argv[1])
''' % ROOT
    script = tmp_path / "sa_split.py"
    script.write_text(code)
    res = {}
    for mode in ("0", "1", "2"):
        f = tmp_path / f"out_{mode}.pt"
        env = dict(os.environ, PQ3D_SA_BWD_SPLIT=mode)
        p = subprocess.run([sys.executable, str(script), str(f)], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        res[mode] = torch.load(f)
    for mode in ("1", "2"):
        for name in res["0"]:
            for a, b in zip(res["0"][name], res[mode][name]):
                assert torch.equal(a, b), (mode, name)
            assert all(torch.isfinite(t).all() and float(t.abs().max()) > 0 for t in res[mode][name])
