"""Fused forward/backward executor for the parallel-structure query decoder (QueryMaskEncoder, query_encoder.py:52-181,
+ MaskHeadSegLevel, mask_head.py:11-57) -- the production fast path.

The modular path (modules.py + ops.py: one autograd Function per kernel) is launch-bound: ~900 dependent dispatches
per config-2 step.  This executor runs the same arithmetic with a hand-written backward so that
  * the M scene memories of a layer are ONE launch each for Q-projection, attention, out-projection, LayerNorm
    (grouped GEMM / memories stacked along the attention batch);
  * K/V projections of every (layer, memory) are hoisted out of the layer loop into ONE grouped GEMM (their inputs
    are layer-invariant), and their backward is one K-concatenated GEMM per gradient;
  * input gradients that sum over consumers are produced by K-concatenated GEMMs / the "+ aux" epilogue instead
    of separate add kernels;
  * every parameter gradient is accumulated (split-K atomics / accumulating column sums) into one flat fp32 arena
    that is zeroed once per backward -- no per-GEMM memsets, no autograd slice/zero/add kernels for the packed
    in_proj weights, and weight sharing across num_blocks accumulates for free.
It is numerically the same computation as the modular path (same kernels, same rounding points).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, List, Optional, Sequence

import torch
from torch.autograd import Function

from . import _lib as L
from . import ops
from ._lib import BF16
from .profiler import timed

MAXG = L.MAXG


def _chunks(seq, n):
    for i in range(0, len(seq), n):
        yield seq[i:i + n]


def _gemm_groups(n_per_call_limit, **kw):
    """L.gemm over arbitrarily many groups (split into calls of <= limit groups; kconcat chains handled by caller)."""
    L.gemm(**kw)


def _colsum_acc(xs: Sequence[torch.Tensor], outs: Sequence[torch.Tensor], rows: int) -> None:
    """outs[g] += column sums of xs[g] viewed as [rows, N] (bias gradients into the arena)."""
    N = xs[0].numel() // rows
    # operands of one output that lie back to back in memory (the mask head's calls write slices of one buffer) are one taller
    # operand: (first tensor, output, number of row blocks)
    merged = []
    for xt, ot in zip(xs, outs):
        if merged and merged[-1][1].data_ptr() == ot.data_ptr() and xt.dtype == merged[-1][0].dtype and \
                xt.data_ptr() == merged[-1][0].data_ptr() + merged[-1][2] * rows * N * xt.element_size():
            merged[-1][2] += 1
        else:
            merged.append([xt, ot, 1])
    for nblk in sorted({m_[2] for m_ in merged}):
        # a launch adds with one (non-atomic) writer per output element: outputs must be unique within a launch
        # (shared weights across num_blocks put the same bias slice in several groups) -> greedy batching
        batches, cur, seen = [], ([], []), set()
        for xt, ot, nb_ in merged:
            if nb_ != nblk:
                continue
            if ot.data_ptr() in seen or len(cur[0]) == MAXG:
                batches.append(cur)
                cur, seen = ([], []), set()
            cur[0].append(xt); cur[1].append(ot); seen.add(ot.data_ptr())
        batches.append(cur)
        for xc, oc in batches:
            xa = (C.c_void_p * len(xc))(*[L.ptr(t) for t in xc])
            oa = (C.c_void_p * len(oc))(*[L.ptr(t) for t in oc])
            # accumulate = 2: the one-writer (bit-reproducible) form at every row count; 1 lets long columns add row slices atomically
            L.check(L.lib().pq3d_colsum_grouped(xa, oa, len(xc), L.dt_of(xc[0]), rows * nblk, N, N, 2 if ops.DETERMINISTIC else 1,
                                                L.stream()), "pq3d_colsum_grouped")


def _splitk(tiles: int, k: int, ct: int, groups: int) -> int:
    nkt = max(1, k // (64 if ct == BF16 else 32))
    return max(1, min(nkt // 2 if nkt >= 2 else 1, max(1, 768 // max(tiles * groups, 1)), 64))


def _dw_acc(gs, xs, x2s, outs, ct, bias_outs=None):
    """outs[g][N,K] += gs[g]^T @ (xs[g] + x2s[g])  (grouped split-K, accumulating atomics) and, fused into the same
    launch, bias_outs[g][N] += column sums of gs[g] (the bias gradient of the same linear layer)."""
    N, K = outs[0].shape
    R = gs[0].numel() // N
    tiles = ((N + 63) // 64) * ((K + 63) // 64)
    epl = 8 if ct == BF16 else 4
    fusable = bias_outs is not None and N % epl == 0 and K % epl == 0 and N >= epl and K >= epl and \
        all(t.data_ptr() % 16 == 0 for t in list(gs) + list(xs))
    gs, xs, x2s = ops.dw_operands(gs, xs, x2s, N, K, R, ct)   # long reductions: bf16 operands once, 128 x 128 tiles
    for i in range(0, len(gs), MAXG):
        g_, x_, o_ = gs[i:i + MAXG], xs[i:i + MAXG], outs[i:i + MAXG]
        x2_ = x2s[i:i + MAXG] if x2s is not None else None
        L.gemm(M=N, N=K, K=R, A=g_, B=x_, B2=x2_, Cs=o_, ct=ct, lda=N, ldb=K, ldc=K, transA=True, transB=True,
               splitk=max(2, _splitk(tiles, R, ct, len(g_))), accumulate=True,
               colsum=bias_outs[i:i + MAXG] if fusable else None)
    if bias_outs is not None and not fusable:
        _colsum_acc(gs, bias_outs, R)


# one-launch row-local chain (csrc/chain_ffn.hip) for the forward's out-projection + LayerNorm + FFN + LayerNorm; PQ3D_CHAIN=0 or
# fused.set_chain(False) keeps the five launches (same bits: A/B measurements, tests)
_CHAIN = os.environ.get("PQ3D_CHAIN", "1") != "0"


def set_chain(on: bool) -> None:
    global _CHAIN
    _CHAIN = bool(on)


def _chain_on(dev) -> bool:
    """Chains requested AND valid on this device (ops.chain_device_ok: MI355X in SPX mode, round-robin workgroup -> XCD placement measured)."""
    return _CHAIN and ops.chain_device_ok(dev)


class _DwQueue:
    """Deferred weight-gradient GEMMs.  Every dW = g^T (x [+ x2]) of the backward pass only feeds the gradient arena,
    so they are queued and flushed at the end as a few grouped launches per (shape, dtype) bucket instead of one
    launch per linear layer (c2: 41 launches -> 6)."""

    def __init__(self, ct):
        self.ct, self.buckets = ct, {}

    def add(self, gs, xs, x2s, outs, ct=None, bias_outs=None):
        for i in range(len(gs)):
            N, K = outs[i].shape
            key = (N, K, gs[i].numel() // N, gs[i].dtype, xs[i].dtype, bias_outs is not None)
            b = self.buckets.setdefault(key, ([], [], [], [], []))
            b[0].append(gs[i]); b[1].append(xs[i]); b[2].append(x2s[i] if x2s is not None else None)
            b[3].append(outs[i]); b[4].append(bias_outs[i] if bias_outs is not None else None)

    def flush(self):
        # (forking the buckets over 3 / 5 streams -- parallel branches of the captured graph -- was measured at config 2:
        # 1.50 -> 1.80 / 2.17 ms; branch joins cost far more than the overlapped tails save)
        multi, rest = [], {}
        for key, (g, x, x2, o, bo) in self.buckets.items():
            N, K, R = key[0], key[1], key[2]
            if self.ct == BF16 and not ops.dw_long_path(N, K, R, len(g), self.ct) and \
                    all(ops.tt_multi_ok(g[i], x[i], x2[i], o[i], bo[i], N, K, R) for i in range(len(g))):
                multi += list(zip(g, x, x2, o, bo))
            else:
                rest[key] = (g, x, x2, o, bo)
        if multi and ops.tt_multi_pays(multi):   # every short-reduction weight / bias gradient of the flush: ONE launch
            ops.tt_multi(multi)                  # (csrc/gemm_ttmulti.hip)
        else:
            rest = self.buckets
        for key, (g, x, x2, o, bo) in rest.items():
            _dw_acc(g, x, x2 if any(t is not None for t in x2) else None, o, self.ct, bo if key[5] else None)
        self.buckets = {}


class FusedSpec:
    """Static description of one fused decoder invocation (non-tensor state handed to the Function)."""

    def __init__(self, enc, mh, mems, ct, act, use_self_mask, num_blocks, spatial, mh_count, offline, skip_pred,
                 drop_base=None, mh_drop=False):
        self.enc, self.mh, self.mems, self.ct, self.act = enc, mh, list(mems), ct, act
        self.use_self_mask, self.num_blocks, self.spatial = use_self_mask, num_blocks, spatial
        self.mh_count, self.offline, self.skip_pred = mh_count, offline, skip_pred
        self.stacked_kpm = None      # optional [M, B, Ns] key-padding masks of the memories, stacked (set by fused_decoder)
        # memory inputs are passed as U unique source tensors: src[i][j] = source of memory j in layer i (a multi-scale
        # voxel memory has one source per layer, query_encoder.py:90-91), mh_src[k] = source of mask-head memory k
        # (the LAST scale of a multi-scale memory, query3d_unified.py:163-165)
        self.src, self.mh_src, self.n_src = None, None, len(self.mems)
        self.prompt = False          # structure 'mixed': a sequential prompt cross-attention follows the parallel scene memories
        self.kv3 = False             # compute mode 'bf16x3': split-bf16 key/value side (set by fused_decoder)
        self.drop_base = drop_base   # dropout-site base of the encoder when train-mode dropout is active, else None
        self.mh_drop = mh_drop       # mask head's cls_head dropout active

    def drop(self, module, app, kind, device, m=0):
        """Dropout site of `module` (its own p) at layer application `app`, or None when dropout is off."""
        if self.drop_base is None or not (module.dropout_p > 0.0):
            return None
        return ops.make_drop(module.dropout_p, ops.drop_site(self.drop_base, app, kind, m), device)


def _mh_forward(spec, x, keys, inv_den, seg_pad, rec, call):
    """One MaskHeadSegLevel call (number `call` of this forward) on the fused path; returns (cls, mlog, amask)."""
    mh = spec.mh
    ct = ops.small_ct(spec.ct)     # query-side GEMMs: split-bf16 in 'bf16' mode (fp32 operands and outputs)
    ad = ops.act_dtype(ct)
    B, Nq, d = x.shape
    R = B * Nq
    c0, c2, c4 = mh.cls_head[0], mh.cls_head[2], mh.cls_head[4]
    Mm = len(keys)
    mps = list(mh.mask_pred_list)[:Mm]
    Ns = keys[0].shape[1]
    if _chain_on(x.device) and ct == L.BF16X3 and not spec.mh_drop and x.dtype == torch.float32 and x.is_contiguous() and \
            ops.chain_mh_ok(d, c0.out_features, c4.out_features, Mm, R) and c4.bias is not None:
        # the row-local part (class MLP + the mask predictions' query projections) in one launch (csrc/chain_mh.hip)
        flags = getattr(mh, "_chain_flags", None)
        if flags is None or flags.device != x.device:
            flags = mh._chain_flags = ops.chain_flags(2048, x.device)
        colfill = mh._foc_flags if mh._foc_cols.numel() else None
        h1, h2, mean, rstd, cls, qm = ops.chain_mh_fwd(
            x, c0.weight.detach(), c0.bias.detach(), c2.weight.detach(), c2.bias.detach(), c2.eps, c4.weight.detach(),
            c4.bias.detach(), colfill, float("-inf"), [mp.q_proj.weight.detach() for mp in mps],
            [mp.q_proj.bias.detach() for mp in mps], flags)
        mlog = torch.empty(B, Ns, Nq, dtype=torch.float32, device=x.device)
        amask = torch.empty(B, Nq, Ns, dtype=torch.bool, device=x.device)
        L.gemm(M=Ns, N=Nq, K=d, A=list(keys), B=[qm[m] for m in range(Mm)], Cs=[mlog] + [None] * (Mm - 1), ct=ct, lda=d,
               ldb=d, ldc=Nq, batch=B, strideA=Ns * d, strideB=Nq * d, strideC=Ns * Nq, kconcat=Mm, row_scale=inv_den,
               row_fill_flag=seg_pad, row_fill=-1e6, mask_out=amask)
        rec.update(mh_x=x.detach(), mh_h1=h1, mh_h2=h2, mh_mean=mean, mh_rstd=rstd, mh_qm=qm, mh_drop=None)
        return cls, mlog, amask
    h1 = torch.empty(B, Nq, c0.out_features, dtype=torch.float32, device=x.device)
    L.gemm(M=R, N=c0.out_features, K=d, A=[x], B=[c0.weight.detach()], bias=[c0.bias.detach()], Cs=[h1], ct=ct,
           lda=d, ldb=d, ldc=c0.out_features, act="relu")
    h2, mean, rstd = _ln_fwd(None, [h1], [c2.weight.detach()], [c2.bias.detach()], c2.eps, None, Nq)
    hdrop = None
    if spec.mh_drop:   # nn.Dropout between LayerNorm and the classifier (utils.py:23), site (mask-head base, call)
        hdrop = ops.make_drop(mh.dropout_p, ops.drop_site(mh._drop_base, call, ops.DROP_MLP_HEAD), x.device)
        h2 = ops._dropout_apply(h2, hdrop)
    C_ = c4.out_features
    cls_raw = torch.empty(B, Nq, C_, dtype=torch.float32, device=x.device)
    L.gemm(M=R, N=C_, K=c0.out_features, A=[h2], B=[c4.weight.detach()], bias=[c4.bias.detach()], Cs=[cls_raw], ct=ct,
           lda=c0.out_features, ldb=c0.out_features, ldc=C_)
    cls = cls_raw
    if mh._foc_cols.numel():
        cls = torch.empty_like(cls_raw)
        L.check(L.lib().pq3d_fill_cols(L.ptr(cls_raw), L.ptr(cls), R, C_, L.ptr(mh._foc_cols), mh._foc_cols.numel(),
                                       float("-inf"), L.stream()), "pq3d_fill_cols")
    qm = torch.empty(Mm, B, Nq, d, dtype=ad, device=x.device)
    L.gemm(M=R, N=d, K=d, A=[x] * Mm, B=[mp.q_proj.weight.detach() for mp in mps],
           bias=[mp.q_proj.bias.detach() for mp in mps], Cs=[qm[m] for m in range(Mm)], ct=ct, lda=d, ldb=d, ldc=d)
    mlog = torch.empty(B, Ns, Nq, dtype=torch.float32, device=x.device)
    amask = torch.empty(B, Nq, Ns, dtype=torch.bool, device=x.device)
    L.gemm(M=Ns, N=Nq, K=d, A=list(keys), B=[qm[m] for m in range(Mm)], Cs=[mlog] + [None] * (Mm - 1), ct=ct, lda=d,
           ldb=d, ldc=Nq, batch=B, strideA=Ns * d, strideB=Nq * d, strideC=Ns * Nq, kconcat=Mm, row_scale=inv_den,
           row_fill_flag=seg_pad, row_fill=-1e6, mask_out=amask)
    rec.update(mh_x=x.detach(), mh_h1=h1, mh_h2=h2, mh_mean=mean, mh_rstd=rstd, mh_qm=qm, mh_drop=hdrop)
    return cls, mlog, amask


def _ln_fwd(x, os_, gammas, betas, eps, coef, rows_per_scene, out_dtype=torch.float32, drop=None, sum_branches=False,
            osum=None):
    M = len(os_)
    dm = os_[0].shape[-1]
    R = os_[0].numel() // dm
    y = torch.empty(os_[0].shape, dtype=out_dtype, device=os_[0].device)
    mean = torch.empty(M, R, dtype=torch.float32, device=y.device)
    rstd = torch.empty_like(mean)
    d = ops._ln_desc(x, os_, gammas, betas, coef, eps, rows_per_scene, y, mean, rstd, drop)
    d.sum_branches, d.osum = int(sum_branches), L.ptr(osum)
    nb = (M + 1 + (x is not None)) * R * dm * 4.0
    L.check(timed("pq3d_add_ln_fwd", f"R{R}d{dm}M{M}", 0.0, nb, L.lib().pq3d_add_ln_fwd, C.byref(d), L.stream()),
            "pq3d_add_ln_fwd")
    return y, mean, rstd


def _ln_bwd(x, os_, gammas, betas, eps, coef, rows_per_scene, mean, rstd, dy, dgs, dbs, want_dx=True, dup_dx=False,
            drop=None, sum_branches=False, dx_zeroed=None):
    """Returns (dx or None, d_o [M,...] fp32 stacked); dgamma/dbeta accumulate into the arena views dgs/dbs.
    dup_dx: also write the (single-branch) input gradient to a second buffer even without a residual input."""
    M = len(os_)
    dm = os_[0].shape[-1]
    R = os_[0].numel() // dm
    dys = list(dy) if isinstance(dy, (list, tuple)) else [dy]   # up to three addends of the upstream gradient, summed in the kernel
    dy = dys[0]
    dev = dy.device
    d_o = torch.empty(1 if sum_branches else M, *os_[0].shape, dtype=torch.float32, device=dev)
    dx = torch.empty(os_[0].shape, dtype=torch.float32, device=dev) if ((x is not None and want_dx) or dup_dx) else None
    d = ops._ln_desc(x, os_, gammas, betas, coef, eps, rows_per_scene, None, mean, rstd, drop)
    if dx_zeroed is not None and dx is not None:   # a slice of the caller's buffer zeroed once for the whole backward
        dx, d.dx_zeroed = dx_zeroed, 1
    d.dy, d.dx, d.accumulate, d.sum_branches = L.ptr(dy), L.ptr(dx), 1, int(sum_branches)
    if len(dys) > 1:
        d.dy2, d.dy3 = L.ptr(dys[1]), L.ptr(dys[2]) if len(dys) > 2 else None
    for m in range(1 if sum_branches else M):
        d.d_o[m], d.dgamma[m], d.dbeta[m] = L.ptr(d_o[m]), L.ptr(dgs[m]), L.ptr(dbs[m])
    nb = (3 * M + 1 + (x is not None)) * R * dm * 4.0
    L.check(timed("pq3d_add_ln_bwd", f"R{R}d{dm}M{M}", 0.0, nb, L.lib().pq3d_add_ln_bwd, C.byref(d), L.stream()),
            "pq3d_add_ln_bwd")
    return dx, d_o


def kv3_ks(Lk: int, dh: int = 32) -> int:
    """Key splits of the split-bf16 cross-attention forward (csrc/attn_x3.hip: at most 1024 keys per workgroup at d_h = 32, 512 at
    d_h = 64) -- a function of the key length only, like every forward split factor (a scene's result must not depend on how scenes
    are batched)."""
    return max(1, -(-((Lk + 63) // 64) // (16 if dh == 32 else 8)))


def _attn(q, k, v, o, lse, H, ct, zero_attn, kpm=None, mask=None, row_open=None, bias=None, mask_bmod=0, bwd=None,
          drop=None, drop_bmod=0, proj_dout=None, mask_bits=None, planes=None):
    """proj_dout = (g, W): backward only -- dO = g W is formed inside the attention kernel (pq3d_attn_proj, DOUT) and
    bwd[0] is ignored; the caller checks sa_fold_ok() first.
    planes = (k_lo, v_lo, q_bf, o_bf): forward only, compute mode 'bf16x3' -- q / o fp32, k / v the hi planes (csrc/attn_x3.hip)."""
    d = ops._attn_desc(q, k, v, o, lse, H, ct, zero_attn, 1.0 / math.sqrt(q.shape[-1] // H), kpm, mask, row_open, bias,
                       drop, drop_bmod, bwd=bwd is not None, mask_bits=mask_bits)
    d.mask_bmod = mask_bmod
    B, Lq, dm = q.shape
    Lk = k.shape[1]
    if planes is not None:
        assert bwd is None and ct == L.BF16X3 and q.dtype == torch.float32 and k.dtype == torch.bfloat16
        d.k_lo, d.v_lo, d.q_bf, d.o_bf = map(L.ptr, planes)
        ks = kv3_ks(Lk, dm // H)
        d.ksplit, d.ws, d._ws_keepalive = 1, None, None
        if ks > 1:
            d._ws_keepalive = torch.empty(ks * B * H * Lq * (dm // H + 2), dtype=torch.float32, device=q.device)
            d.ksplit, d.ws = ks, L.ptr(d._ws_keepalive)
    key = f"B{B}H{H}Lq{Lq}Lk{Lk}dh{dm // H}ct{ct}" + ("m3" if mask is not None else "")
    if bwd is None:
        L.check(timed("pq3d_attn_fwd", key, 4.0 * B * Lq * Lk * dm, (q.numel() * 2 + k.numel() * 2) * q.element_size(),
                      L.lib().pq3d_attn_fwd, C.byref(d), L.stream()), "pq3d_attn_fwd")
    else:
        d.dout, d.dq, d.dk, d.dv, d.delta, d.dbias = map(L.ptr, bwd)
        if proj_dout is not None:
            d.proj.mode, d.proj.dm, d.proj.x = 2, dm, L.ptr(proj_dout[0])
            d.proj.w[0] = L.ptr(proj_dout[1])
        L.check(timed("pq3d_attn_bwd", key, 8.0 * B * Lq * Lk * dm, (q.numel() * 3 + k.numel() * 4) * q.element_size(),
                      L.lib().pq3d_attn_bwd, C.byref(d), L.stream()), "pq3d_attn_bwd")


def sa_fold_ok(ct, B, H, L_, dm, drop, df, W) -> bool:
    """The split-bf16 self-attention backward kernel can form dO itself (attn_sa.hip): its shape limits + 160 KB of LDS."""
    if ops.sa_ct(ct) != L.BF16X3 or dm != 32 * H or dm % 32 or drop is not None:
        return False
    if df.dtype != torch.float32 or W.dtype != torch.float32 or not df.is_contiguous() or not W.is_contiguous():
        return False
    if (df.data_ptr() | W.data_ptr()) & 15:
        return False
    lp2, lpk = (L_ + 31) & ~31, (L_ + 31) & ~31
    lds = (4 * lp2 + 4 * lpk + dm) * 40 * 2 + (lpk + 2 * lp2) * 4 + 16
    return L_ <= 240 and lds <= 160 * 1024


class _PendingDx:
    """Upstream gradient of a layer application that the NEXT backward chain launch forms itself (csrc/chain_ffn_bwd.hip, step 0):
    sum_m dq_m Wq_m + dxr."""
    __slots__ = ("dq_all", "Wqs", "dxr", "gq")

    def __init__(self, dq_all, Wqs, dxr, gq):
        self.dq_all, self.Wqs, self.dxr, self.gq = dq_all, Wqs, dxr, gq


class _DecoderBackward:
    """One backward pass of the fused decoder: the state every sublayer step shares (saved tensors, sizes, the gradient
    arena, the deferred weight-gradient queue, the K/V gradient buffers) and one method per step, run in reverse order of the
    forward: mask head -> FFN -> self-attention -> (prompt cross-attention) -> cross-attention over the scene memories, per
    layer application; then the hoisted K/V projections' backward.  `_FusedDecoder.backward` is `_DecoderBackward(...).run()`."""

    def __init__(self, ctx, dxf, dheads):
        self.ctx, self.dxf = ctx, dxf
        self.spec, self.tape = spec, _ = ctx.spec, ctx.tape
        self.enc, self.ct = enc, ct = spec.enc, spec.ct
        self.ad = ops.act_dtype(ct)
        self.M, self.U, self.src, self.mh_src = M, U, _, _ = ctx.M, ctx.U, ctx.src, ctx.mh_src
        sv = ctx.saved_tensors
        self.x0, self.qpos, self.qmask, self.pos, self.pl, self.seg_pad, self.coef = sv[:7]
        self.feats, self.masks = list(sv[7:7 + U]), list(sv[7 + U:7 + U + M])
        self.params = ctx.params
        self.layers = list(enc.unified_encoder)
        self.Ln, self.H = len(self.layers), enc.num_heads
        self.B, self.Nq, self.d = B, Nq, d = self.qpos.shape
        self.Ns = Ns = self.feats[0].shape[1]
        self.R, self.Rk = B * Nq, B * Ns
        self.dev = dev = self.qpos.device
        self.cas, self.KV = ctx.cas, ctx.KV
        self.n_mh = n_mh = ctx.n_mh
        self.dcls, self.dmlog = list(dheads[:n_mh]), list(dheads[n_mh:2 * n_mh])
        self.n_app = n_app = len(self.tape)
        self._open_arena()   # self.gv (parameter -> gradient view), self.accumulate, self.in_place, self.dxr_zero
        self.dwq = _DwQueue(ct)
        self.sb_queue = []   # (W, b, d bias, dW, db) of every spatial self-attention application
        self.dqpos_parts: List[torch.Tensor] = []
        self.dKV = torch.empty(n_app, 2, M, B, Ns, d, dtype=self.ad, device=dev)
        self.dPKV = torch.empty(n_app, 2, B, ctx.prompt.shape[1], d, dtype=self.ad, device=dev) if spec.prompt else None
        self._mh_calls, self._mh_pre, self._mh_dcl = 0, None, None   # mask_head_chain(): per-pass buffers shared by its calls
        self.dkeys = None    # accumulated gradient of the mask-head key projections [Mm,B,Ns,d] fp32->ad
        self.dk_terms = []   # queued (g, q_m) terms of it (one K-concatenated launch at the end)
        ready_cb = getattr(enc, "grads_ready", None) if self.in_place else None   # only when the owner's buffers were written
        self.per_layer = bool(getattr(enc, "grad_bucket_per_layer", False)) and ready_cb is not None
        if ready_cb is not None:
            def ready(tag):
                # weight gradients queued by ops.linear layers that ran backward BEFORE the decoder (heads; ops._DwDeferred)
                # must be in their slots before an owner is told that a bucket is final
                ops.dw_deferred_flush()
                ready_cb(tag)
        else:
            ready = lambda tag: None
        self.ready = ready
        # the output heads consume the decoder's outputs, so their backward has run when this one starts (the reference:
        # query3d_unified.py:193-218): whatever they wrote into their arena slots is final now -- an owner may start that
        # bucket's all-reduce under the whole decoder backward (config 5: the caption body's 240 MB)
        ready("heads")

    def _open_arena(self):
        """Gradient arena: every parameter gradient of the decoder is a view of a flat zeroed fp32 buffer (the data-parallel
        reducer's / optimizer's, or one of this pass); decides fresh vs accumulating pass."""
        tape, enc, M, params, B, Nq, d = self.tape, self.enc, self.M, self.params, self.B, self.Nq, self.d
        dev, = self.dev,
        # ---- gradient arena: one flat zeroed fp32 buffer, every parameter gradient is a view of it
        sizes = [p.numel() for p in params]
        ext = getattr(enc, "grad_arena", None)   # {id(param): (flat, offset, numel)} of a DP reducer's flat buffers
        gv = {}
        n_app = len(tape)
        # input gradients of the M-branch cross-attention LayerNorms are accumulated with atomics by the M branch blocks; that
        # buffer and the gradient arena are zeroed by ONE launch for the whole backward
        dxr_zero = torch.empty(n_app, B, Nq, d, dtype=torch.float32, device=dev) if M > 1 else None
        accumulate = False
        in_place = ext is not None and all(id(p) in ext for p in params)
        if in_place:
            # gradients go straight into the owner's flat buffer (data-parallel bucket / optimizer arena): no pack copy.
            for p in params:
                flat, o_, n_ = ext[id(p)]
                gv[id(p)] = flat[o_:o_ + n_].view(p.shape)
            # torch semantics of a second backward before zero_grad: gradients ACCUMULATE (the reference trains under
            # accelerator.accumulate, trainer/query3d_trainer.py:35).  Every parameter gradient of this backward is formed
            # by accumulating launches (split-K atomics, accumulating column sums), so accumulation = not zeroing the arena.
            # Which case this is is read off the parameters: .grad still aliasing the arena -> the owner has not consumed the
            # previous micro-batch -> add in place (and hand autograd nothing: .grad already is the arena);
            # .grad None / foreign everywhere -> fresh step: zero, then hand fresh views to autograd (adopted without a copy).
            req = [p for p in params if p.requires_grad]
            alias = [p.grad is not None and p.grad.data_ptr() == gv[id(p)].data_ptr() for p in req]
            # parameters outside the decoder (the input encoders) with slots in the buffers zeroed here: offered to the
            # backward functions that run after this one in the same pass (ops.arena_offer)
            bufs = list(getattr(enc, "grad_arena_buffers", ()))
            zeroed, own = {b.data_ptr() for b in bufs}, {id(p) for p in params}
            offer = {}
            for i_, q_ in getattr(ext, "params", {}).items():
                if i_ not in own and q_.requires_grad and ext[i_][0].data_ptr() in zeroed:
                    fl_, o_, n_ = ext[i_]
                    offer[q_.data_ptr()] = (q_, fl_, o_, n_)
            mixed = "fused decoder backward: some parameters' .grad alias the shared gradient arena and others do not -- " \
                    "zero ALL gradients (set_to_none=True) or none between micro-batches"
            if ops._Arena.whole_pass and ops._Arena.mode is not None and \
                    all(p.data_ptr() in ops._Arena.by_ptr for p in req):
                # the owner opened the arena around the whole pass (ops.grad_arena): already zeroed / accumulating
                accumulate = ops._Arena.mode == "accumulate"
                if accumulate and not all(alias):
                    raise RuntimeError(mixed)
                offer = {}
                if not ops.arena_flush_zero([dxr_zero]):   # first consumer of a fresh pass: arena + own scratch, one launch
                    ops.zero_many([dxr_zero])
            elif req and all(alias):
                accumulate = True
                ops.zero_many([dxr_zero])
            elif any(alias) or any(q_.grad is not None and q_.grad.data_ptr() == f_.data_ptr() + 4 * o_ for q_, f_, o_, n_ in offer.values()):
                raise RuntimeError(mixed)
            else:
                ops.zero_many(bufs + [dxr_zero])
            if offer:
                ops.arena_offer(offer, "accumulate" if accumulate else "fresh")
        else:
            arena = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
            ops.zero_many([arena, dxr_zero])
            off = 0
            for p, n in zip(params, sizes):
                gv[id(p)] = arena[off:off + n].view(p.shape)
                off += n
        self.gv, self.dxr_zero, self.accumulate, self.in_place = gv, dxr_zero, accumulate, in_place

    def G(self, p):
        return self.gv[id(p)]

    def mask_head(self, rec, dc, dm, dx_in):
        """Backprop one mask-head call; returns dx_in + its contribution to d(query)."""
        ctx, spec, ct, ad, M, seg_pad, B = self.ctx, self.spec, self.ct, self.ad, self.M, self.seg_pad, self.B
        Nq, d, Ns, R, dev, G, dwq = self.Nq, self.d, self.Ns, self.R, self.dev, self.G, self.dwq
        mh = spec.mh
        c0, c2, c4 = mh.cls_head[0], mh.cls_head[2], mh.cls_head[4]
        x_in = rec["mh_x"]
        Hd, C_ = c0.out_features, c4.out_features
        cur = dx_in
        if self.mask_head_chain_ok(rec, dc, dm, cur):
            return self.mask_head_chain(rec, dc, dm, cur)
        assert not isinstance(cur, _PendingDx)
        if dc is not None:
            dcl = dc.contiguous()
            if mh._foc_cols.numel():
                t = torch.empty_like(dcl)
                L.check(L.lib().pq3d_fill_cols(L.ptr(dcl), L.ptr(t), R, C_, L.ptr(mh._foc_cols),
                                               mh._foc_cols.numel(), 0.0, L.stream()), "pq3d_fill_cols")
                dcl = t
            dh2 = torch.empty(B, Nq, Hd, dtype=torch.float32, device=dev)
            L.gemm(M=R, N=Hd, K=C_, A=[dcl], B=[c4.weight.detach()], Cs=[dh2], ct=ct, lda=C_, ldb=Hd, ldc=Hd, transB=True)
            dwq.add([dcl], [rec["mh_h2"]], None, [G(c4.weight)], ct, [G(c4.bias)])
            if rec.get("mh_drop") is not None:
                dh2 = ops._dropout_apply(dh2, rec["mh_drop"])
            _, dh1 = _ln_bwd(None, [rec["mh_h1"]], [c2.weight.detach()], [c2.bias.detach()], c2.eps, None, Nq,
                             rec["mh_mean"], rec["mh_rstd"], dh2, [G(c2.weight)], [G(c2.bias)])
            dpre = ops.act_bwd(dh1[0], rec["mh_h1"], "relu", ad)
            nxt = torch.empty(B, Nq, d, dtype=torch.float32, device=dev)
            L.gemm(M=R, N=d, K=Hd, A=[dpre], B=[c0.weight.detach()], Cs=[nxt], aux=[cur], act_grad="add", ct=ct,
                   lda=Hd, ldb=d, ldc=d, transB=True)
            dwq.add([dpre], [x_in], None, [G(c0.weight)], ct, [G(c0.bias)])
            cur = nxt
        if dm is not None:
            Mm = spec.mh_count
            mps = list(mh.mask_pred_list)[:Mm]
            qm = rec["mh_qm"]
            g = ops.scale_rows(dm.contiguous(), B * Ns, ad, scale=ctx.inv_den, zero_flag=seg_pad)
            # d keys = sum over the mask-head calls of g_c @ q_m,c: nothing consumes it before the end of the backward, so
            # the terms are queued and formed by ONE K-concatenated launch there (each call used to re-read and re-write
            # the [Mm, B, Ns, d] fp32 sum through the "+ aux" epilogue: 100 MB per call at config 4)
            if Mm * (len(self.dk_terms) + 1) <= MAXG:
                self.dk_terms.append((g, qm))
            else:
                newk = torch.empty(Mm, B, Ns, d, dtype=torch.float32, device=dev)
                L.gemm(M=Ns, N=d, K=Nq, A=[g] * Mm, B=[qm[m] for m in range(Mm)], Cs=[newk[m] for m in range(Mm)],
                       aux=[self.dkeys[m] for m in range(Mm)] if self.dkeys is not None else None,
                       act_grad="add" if self.dkeys is not None else None, ct=ct, lda=Nq, ldb=d, ldc=d, transB=True, batch=B,
                       strideA=Ns * Nq, strideB=Nq * d, strideC=Ns * d)
                self.dkeys = newk
            # [Nq x d] outputs per (memory, scene) over a reduction of Ns segments: few tiles, long K -> split-K into an
            # fp32 buffer once Ns is large (c4: 192 workgroups x 64 k-tiles otherwise)
            sk = min(8, Ns // 512) if Ns >= 1024 else 1
            dqm = torch.empty(Mm, B, Nq, d, dtype=torch.float32 if sk > 1 else ad, device=dev)
            L.gemm(M=Nq, N=d, K=Ns, A=[g] * Mm, B=list(ctx.keys), Cs=[dqm[m] for m in range(Mm)], ct=ct, lda=Nq,
                   ldb=d, ldc=d, transA=True, transB=True, batch=B, strideA=Ns * Nq, strideB=Ns * d, strideC=Nq * d,
                   splitk=sk)
            nxt = torch.empty(B, Nq, d, dtype=torch.float32, device=dev)
            L.gemm(M=R, N=d, K=d, A=[dqm[m] for m in range(Mm)], B=[mp.q_proj.weight.detach() for mp in mps],
                   Cs=[nxt] + [None] * (Mm - 1), aux=[cur] + [None] * (Mm - 1), act_grad="add", ct=ct, lda=d, ldb=d,
                   ldc=d, transB=True, kconcat=Mm)
            dwq.add([dqm[m] for m in range(Mm)], [x_in] * Mm, None, [G(mp.q_proj.weight) for mp in mps], ct,
                    [G(mp.q_proj.bias) for mp in mps])
            cur = nxt
        return cur

    def mask_head_chain_ok(self, rec, dc, dm, cur) -> bool:
        mh = self.spec.mh
        c0, c4 = mh.cls_head[0], mh.cls_head[4]
        if isinstance(cur, _PendingDx):
            cur = cur.dxr
        return (dc is not None and dm is not None and _chain_on(self.dev) and self.ct == BF16 and self.ad == torch.bfloat16 and
                rec.get("mh_drop") is None and ops.chain_mh_ok(self.d, c0.out_features, c4.out_features, self.spec.mh_count, self.R)
                and rec["mh_h1"].dtype == torch.float32 and isinstance(cur, torch.Tensor) and cur.dtype == torch.float32
                and cur.is_contiguous())

    def _mh_logit_grads(self):
        """{id(rec): (g, dqm)} of every mask-head call of the pass: the query-side gradient of the mask logits, dqm_c =
        g_c^T keys, depends on the loss' gradient of that call's logits and on the keys only -- not on anything the backward
        pass computes -- so the products of ALL calls are formed by ONE launch when the first call runs backward (config 4: 5
        launches of 31-34 us -> one of ~105; their split-K outputs are one buffer zeroed by one launch)."""
        if self._mh_pre is not None:
            return self._mh_pre
        ctx, spec, ct, ad, B, Nq, d, Ns, dev = self.ctx, self.spec, self.ct, self.ad, self.B, self.Nq, self.d, self.Ns, self.dev
        Mm = spec.mh_count
        calls = []
        if ctx.final_rec is not None:
            calls.append((ctx.final_rec, self.dcls[-1], self.dmlog[-1]))
        if spec.mh is not None and not spec.skip_pred:
            calls += [(self.tape[a], self.dcls[a], self.dmlog[a]) for a in range(self.n_app - 1, -1, -1)]
        calls = [(r_, dm_) for r_, dc_, dm_ in calls if self.mask_head_chain_ok(r_, dc_, dm_, self.qpos)]   # (qpos: any fp32 rows)
        self._mh_pre = {}
        per = max(1, MAXG // max(Mm, 1))
        if not calls or Mm < 1:
            return self._mh_pre
        sk = min(8, Ns // 512) if Ns >= 1024 else 1
        buf = torch.empty(len(calls), Mm, B, Nq, d, dtype=torch.float32 if sk > 1 else ad, device=dev)
        if sk > 1:
            ops.zero_many([buf])   # split-K partial sums are added into the outputs
        gs = ops.scale_rows_many([dm_ for _r, dm_ in calls], B * Ns, ad, scale=ctx.inv_den, zero_flag=self.seg_pad)   # one launch
        for c0_ in range(0, len(calls), per):
            idx = range(c0_, min(c0_ + per, len(calls)))
            L.gemm(M=Nq, N=d, K=Ns, A=[gs[c_] for c_ in idx for _m in range(Mm)], B=list(ctx.keys) * len(idx),
                   Cs=[buf[c_, m] for c_ in idx for m in range(Mm)], ct=ct, lda=Nq, ldb=d, ldc=d, transA=True, transB=True, batch=B,
                   strideA=Ns * Nq, strideB=Ns * d, strideC=Nq * d, splitk=sk, accumulate=sk > 1)
        for c_, (r_, _dm) in enumerate(calls):
            self._mh_pre[id(r_)] = (gs[c_], buf[c_])
        return self._mh_pre

    def mask_head_chain(self, rec, dc, dm, cur):
        """mask_head() with the row-local steps in one launch (csrc/chain_mh.hip): the mask logits' query-side gradient first
        (not row-local: a reduction over the scene's segments), then class-MLP backward + both input-gradient products."""
        ctx, spec, ct, ad, seg_pad, B = self.ctx, self.spec, self.ct, self.ad, self.seg_pad, self.B
        Nq, d, Ns, dev, G, dwq = self.Nq, self.d, self.Ns, self.dev, self.G, self.dwq
        mh = spec.mh
        c0, c2, c4 = mh.cls_head[0], mh.cls_head[2], mh.cls_head[4]
        x_in = rec["mh_x"]
        Mm = spec.mh_count
        mps = list(mh.mask_pred_list)[:Mm]
        qm = rec["mh_qm"]
        pre = self._mh_logit_grads().get(id(rec))
        g = pre[0] if pre is not None else ops.scale_rows(dm.contiguous(), B * Ns, ad, scale=ctx.inv_den, zero_flag=seg_pad)
        if Mm * (len(self.dk_terms) + 1) <= MAXG:
            self.dk_terms.append((g, qm))
        else:
            newk = torch.empty(Mm, B, Ns, d, dtype=torch.float32, device=dev)
            L.gemm(M=Ns, N=d, K=Nq, A=[g] * Mm, B=[qm[m] for m in range(Mm)], Cs=[newk[m] for m in range(Mm)],
                   aux=[self.dkeys[m] for m in range(Mm)] if self.dkeys is not None else None,
                   act_grad="add" if self.dkeys is not None else None, ct=ct, lda=Nq, ldb=d, ldc=d, transB=True, batch=B,
                   strideA=Ns * Nq, strideB=Nq * d, strideC=Ns * d)
            self.dkeys = newk
        call = self._mh_calls
        self._mh_calls += 1
        if pre is not None:
            dqm = pre[1]
        else:
            sk = min(8, Ns // 512) if Ns >= 1024 else 1
            dqm = torch.empty(Mm, B, Nq, d, dtype=torch.float32 if sk > 1 else ad, device=dev)
            L.gemm(M=Nq, N=d, K=Ns, A=[g] * Mm, B=list(ctx.keys), Cs=[dqm[m] for m in range(Mm)], ct=ct, lda=Nq,
                   ldb=d, ldc=d, transA=True, transB=True, batch=B, strideA=Ns * Nq, strideB=Ns * d, strideC=Nq * d,
                   splitk=sk)
        if self._mh_dcl is None and mh._foc_cols.numel():   # adjacent rows call after call: their column sums (the bias
            self._mh_dcl = torch.empty(self.n_mh, B, Nq, c4.out_features, dtype=torch.float32, device=dev)   # gradient) in one launch
        flags = getattr(mh, "_chain_flags_bwd", None)
        if flags is None or flags.device != dev:
            flags = mh._chain_flags_bwd = ops.chain_flags(2048, dev)
        dcl, dpre, out = ops.chain_mh_bwd(
            dc.contiguous(), mh._foc_flags if mh._foc_cols.numel() else None, c4.weight.detach(), rec["mh_h1"], rec["mh_mean"],
            rec["mh_rstd"], c2.weight.detach(), G(c2.weight), G(c2.bias), c0.weight.detach(),
            None if isinstance(cur, _PendingDx) else cur,
            [dqm[m] for m in range(Mm)], [mp.q_proj.weight.detach() for mp in mps], flags,
            dcl_out=self._mh_dcl[call] if self._mh_dcl is not None else None,
            prev=(cur.dq_all, cur.Wqs, cur.dxr, cur.gq) if isinstance(cur, _PendingDx) else None)
        dwq.add([dcl], [rec["mh_h2"]], None, [G(c4.weight)], ct, [G(c4.bias)])
        dwq.add([dpre], [x_in], None, [G(c0.weight)], ct, [G(c0.bias)])
        dwq.add([dqm[m] for m in range(Mm)], [x_in] * Mm, None, [G(mp.q_proj.weight) for mp in mps], ct,
                [G(mp.q_proj.bias) for mp in mps])
        return out

    def kv_terms(self, apps, into_queue):
        """(dK|dV, W) operand lists of the hoisted K/V projections' backward for the applications `apps`; with
        into_queue their weight / bias gradient products are queued."""
        ctx, spec, tape, ct, src, d, cas = self.ctx, self.spec, self.tape, self.ct, self.src, self.d, self.cas
        G, dwq, dKV, dPKV = self.G, self.dwq, self.dKV, self.dPKV
        A_, B_, Xf, X2, GWs, Gbs = [], [], [], [], [], []
        for a_ in apps:
            i_ = tape[a_]["i"]
            for j, ca in enumerate(cas[i_]):
                w = ca.multihead_attn.in_proj_weight.detach()
                gw, gb = G(ca.multihead_attn.in_proj_weight), G(ca.multihead_attn.in_proj_bias)
                A_ += [dKV[a_, 0, j], dKV[a_, 1, j]]
                B_ += [w[d:2 * d], w[2 * d:]]
                Xf += [ctx.kin[src[i_][j]], ctx.vin[src[i_][j]]]
                X2 += [ctx.kin2[src[i_][j]], None]
                GWs += [gw[d:2 * d], gw[2 * d:]]
                Gbs += [gb[d:2 * d], gb[2 * d:]]
        if into_queue:
            dwq.add(A_, Xf, X2, GWs, ct, Gbs)
            if spec.prompt:   # the prompt memory's K / V rows of its cross-attention's in_proj weights
                for a_ in apps:
                    pc_ = ctx.pcas[tape[a_]["i"]]
                    gw, gb = G(pc_.multihead_attn.in_proj_weight), G(pc_.multihead_attn.in_proj_bias)
                    dwq.add([dPKV[a_, 0], dPKV[a_, 1]], [ctx.prompt, ctx.prompt], None, [gw[d:2 * d], gw[2 * d:]], ct,
                            [gb[d:2 * d], gb[2 * d:]])
        return A_, B_

    def flush_spatial(self):
        pl, H, B, Nq, sb_queue = self.pl, self.H, self.B, self.Nq, self.sb_queue
        for i0 in range(0, len(sb_queue), MAXG):
            chunk = sb_queue[i0:i0 + MAXG]
            arrs = [(C.c_void_p * len(chunk))(*[L.ptr(t[k]) for t in chunk]) for k in range(5)]
            L.check(L.lib().pq3d_spatial_bias_bwd_grouped(L.ptr(pl), *arrs, len(chunk), B, H, Nq, L.stream()),
                    "pq3d_spatial_bias_bwd_grouped")
        del sb_queue[:]

    def ffn(self, rec, layer, dx):
        """FFN sublayer: returns d(x2), the gradient of the sublayer's input (residual + linear1 path)."""
        spec, ct, ad, M, B, Nq, d = self.spec, self.ct, self.ad, self.M, self.B, self.Nq, self.d
        R, dev, accumulate, G, dwq = self.R, self.dev, self.accumulate, self.G, self.dwq
        x2 = rec["x2"]
        # ---------------- FFN backward
        ffn = layer.ffn
        F_ = ffn.linear1.out_features
        # dx2r = residual-branch gradient, dy = (dropout-masked) gradient of the linear2 output (sum of the partials)
        dx2r, dy = _ln_bwd(x2, [rec["z"]], [ffn.norm.weight.detach()], [ffn.norm.bias.detach()], ffn.norm.eps, None, Nq,
                           rec["mean_f"][:1], rec["rstd_f"][:1], dx, [G(ffn.norm.weight)], [G(ffn.norm.bias)],
                           drop=rec["dr_fr"])
        dy = dy[0]
        dhp = torch.empty(B, Nq, F_, dtype=ad, device=dev)
        # inner dropout.  ReLU: the saved h is post-dropout, so [h > 0] already carries the keep-mask and only the
        # 1/(1-p) factor is left -> alpha.  GELU (round 3): the epilogue regenerates the forward's mask on the incoming
        # gradient (same site, same [R, F] indices) before multiplying by gelu'(pre) -- epi_row's order: dropout, then
        # the activation gradient
        relu = spec.act != "gelu"
        L.gemm(M=R, N=F_, K=d, A=[dy], B=[ffn.linear2.weight.detach()], Cs=[dhp],
               aux=[rec["h"] if relu else rec["pre"]], act_grad=spec.act, ct=ct, lda=d, ldb=F_, ldc=F_,
               transB=True, alpha=1.0 / (1.0 - rec["dr_fi"].p) if (rec["dr_fi"] is not None and relu) else 1.0,
               drop=rec["dr_fi"] if not relu else None)
        dwq.add([dy], [rec["h"]], None, [G(ffn.linear2.weight)], ct, [G(ffn.linear2.bias)])
        dx2 = dx2r   # dx2 = dx2r + dhp W1: split-K accumulated in place onto the residual-branch gradient
        L.gemm(M=R, N=d, K=F_, A=[dhp], B=[ffn.linear1.weight.detach()], Cs=[dx2], ct=ct, lda=F_, ldb=d, ldc=d,
               transB=True, splitk=4, accumulate=True)
        dwq.add([dhp], [x2], None, [G(ffn.linear1.weight)], ct, [G(ffn.linear1.bias)])
        return dx2

    def sa_chain_ok(self, rec) -> bool:
        """q / k / v input gradients + merged LayerNorm backward + cross-attention d O as one launch (csrc/chain_sa_bwd.hip)."""
        return (_chain_on(self.dev) and self.ct == BF16 and self.ad == torch.bfloat16 and not self.spec.prompt and rec["dr_cr"] is None
                and ops.chain_ca_ok(self.R, self.d, self.M) and rec["op_all"].dtype == torch.float32)

    def ffn_chain_ok(self, rec, layer, dx) -> bool:
        """The FFN backward + the self-attention post-norm backward can run as ONE launch (csrc/chain_ffn_bwd.hip)."""
        ffn = layer.ffn
        return (_chain_on(self.dev) and self.ct == BF16 and self.ad == torch.bfloat16 and self.spec.act == "relu" and isinstance(dx, (torch.Tensor, _PendingDx))
                and rec["dr_fr"] is None and rec["dr_fi"] is None and rec["dr_sr"] is None and rec["h"].dtype == torch.float32
                and ops.chain_ffn_ok(self.R, self.d, ffn.linear1.out_features))

    def ffn_chain(self, rec, layer, dx):
        """FFN sublayer + the self-attention post-norm in one launch: returns (dx1r, df), both the same gradient (no dropout);
        queues the two weight-gradient products of the FFN exactly as ffn() does."""
        ct, G, dwq, enc, dev = self.ct, self.G, self.dwq, self.enc, self.dev
        ffn, sa = layer.ffn, layer.self_attn
        flags = getattr(enc, "_chain_flags_bwd", None)
        if flags is None or flags.device != dev:
            flags = enc._chain_flags_bwd = ops.chain_flags(2048, dev)
        prev = (dx.dq_all, dx.Wqs, dx.dxr, dx.gq) if isinstance(dx, _PendingDx) else None
        dy, dhp, df = ops.chain_ffn_bwd(
            None if prev is not None else dx.contiguous(), rec["x2"], rec["z"], ffn.norm.weight.detach(), rec["mean_f"][:1], rec["rstd_f"][:1], G(ffn.norm.weight),
            G(ffn.norm.bias), ffn.linear2.weight.detach(), rec["h"], ffn.linear1.weight.detach(), rec["x1s"], rec["f"],
            sa.norm.weight.detach(), rec["mean_s"], rec["rstd_s"], G(sa.norm.weight), G(sa.norm.bias), flags, prev=prev)[:3]
        dwq.add([dy], [rec["h"]], None, [G(ffn.linear2.weight)], ct, [G(ffn.linear2.bias)])
        dwq.add([dhp], [rec["x2"]], None, [G(ffn.linear1.weight)], ct, [G(ffn.linear1.bias)])
        return df, df

    def self_attn(self, rec, layer, dx2, pre=None):
        """Self-attention sublayer: returns the three addends of d(x1s) (from q, from k, from v + residual)."""
        spec, ct, M, qpos, qmask, H, B = self.spec, self.ct, self.M, self.qpos, self.qmask, self.H, self.B
        Nq, d, R, dev, G, dwq, sb_queue = self.Nq, self.d, self.R, self.dev, self.G, self.dwq, self.sb_queue
        dqpos_parts, = self.dqpos_parts,
        # ---------------- self-attention backward
        sa = layer.self_attn
        if spec.spatial:
            msa = sa.self_attn
            Wl = [msa.w_qs.weight, msa.w_ks.weight, msa.w_vs.weight]
            GW = [G(w) for w in Wl]
            Gb = [G(msa.w_qs.bias), G(msa.w_ks.bias), G(msa.w_vs.bias)]
            Wl = [w.detach() for w in Wl]
            Wo, GWo, Gbo = msa.fc.weight.detach(), G(msa.fc.weight), G(msa.fc.bias)
        else:
            w, gw, gb = sa.self_attn.in_proj_weight.detach(), G(sa.self_attn.in_proj_weight), G(sa.self_attn.in_proj_bias)
            Wl = [w[:d], w[d:2 * d], w[2 * d:]]
            GW = [gw[:d], gw[d:2 * d], gw[2 * d:]]
            Gb = [gb[:d], gb[d:2 * d], gb[2 * d:]]
            Wo, GWo, Gbo = sa.self_attn.out_proj.weight.detach(), G(sa.self_attn.out_proj.weight), G(sa.self_attn.out_proj.bias)
        x1s = rec["x1s"]     # the self-attention sublayer's input (x1, or the prompt cross-attention's output)
        if pre is not None:   # formed by the chain launch (ffn_chain)
            dx1r, df = pre
        else:
            dx1r, df = _ln_bwd(x1s, [rec["f"]], [sa.norm.weight.detach()], [sa.norm.bias.detach()], sa.norm.eps, None, Nq,
                               rec["mean_s"], rec["rstd_s"], dx2, [G(sa.norm.weight)], [G(sa.norm.bias)],
                               drop=rec["dr_sr"])
            df = df[0]
        fold = sa_fold_ok(ct, B, H, Nq, d, rec["dr_sa"], df, Wo)
        do_s = None
        if not fold:
            do_s = torch.empty(B, Nq, d, dtype=torch.float32, device=dev)
            L.gemm(M=R, N=d, K=d, A=[df], B=[Wo], Cs=[do_s], ct=ct, lda=d, ldb=d, ldc=d, transB=True)
        dwq.add([df], [rec["o_s"]], None, [GWo], ct, [Gbo])
        qkv = rec["qkv"]
        dqkv = torch.empty(3, B, Nq, d, dtype=torch.float32, device=dev)
        delta = torch.empty(B, H, Nq, dtype=torch.float32, device=dev)
        dsb = torch.empty_like(rec["sbias"]) if spec.spatial else None
        _attn(qkv[0], qkv[1], qkv[2], rec["o_s"], rec["lse_s"], H, ops.sa_ct(ct), False, kpm=qmask, bias=rec["sbias"],
              bwd=(do_s, dqkv[0], dqkv[1], dqkv[2], delta, dsb), drop=rec["dr_sa"],
              proj_dout=(df, Wo) if fold else None)
        if spec.spatial:   # deferred: one grouped launch for all layer applications at the end of the backward
            sb_queue.append((msa.pairwise_loc_fc.weight.detach(), msa.pairwise_loc_fc.bias.detach(), dsb,
                             G(msa.pairwise_loc_fc.weight), G(msa.pairwise_loc_fc.bias)))
        # d(x1 + qpos) from q and k, d(x1) from v (+ the residual-branch gradient): three independent products, ONE
        # launch; their sum is formed by the consumer (the next LayerNorm backward reads three addends) instead of
        # by a second, dependent "+ aux" launch
        self._sa_chain = None
        if self.sa_chain_ok(rec):
            # these products, the merged cross-attention post-norm backward and the cross-attention d O in ONE launch
            # (csrc/chain_sa_bwd.hip); cross_attn() picks the results up
            cl = self.cas[rec["i"]]
            enc = self.enc
            flags = getattr(enc, "_chain_flags_sab", None)
            if flags is None or flags.device != dev:
                flags = enc._chain_flags_sab = ops.chain_flags(2048, dev)
            a_ = self._cur_app
            g3, dop, dxr, do_all = ops.chain_sa_bwd(
                dqkv, [w_.contiguous() for w_ in Wl], dx1r, rec["x_in"], rec["op_all"], [ca.norm.weight.detach() for ca in cl],
                rec["mean_c"], rec["rstd_c"], self.coef[a_] if self.coef is not None else None, Nq, [G(ca.norm.weight) for ca in cl],
                [G(ca.norm.bias) for ca in cl], [ca.multihead_attn.out_proj.weight.detach() for ca in cl], flags)
            self._sa_chain = (dxr, dop, do_all)
        else:
            g3 = torch.empty(3, B, Nq, d, dtype=torch.float32, device=dev)
            L.gemm(M=R, N=d, K=d, A=[dqkv[0], dqkv[1], dqkv[2]], B=list(Wl), Cs=[g3[0], g3[1], g3[2]],
                   aux=[None, None, dx1r], act_grad="add", ct=ct, lda=d, ldb=d, ldc=d, transB=True)
        dqpos_parts += [g3[0], g3[1]]
        dx1 = [g3[0], g3[1], g3[2]]
        dwq.add([dqkv[0], dqkv[1], dqkv[2]], [x1s] * 3, [qpos, qpos, None], GW, ct, Gb)
        return dx1

    def prompt_cross_attn(self, a, rec, dx1):
        """structure 'mixed': the prompt memory's sequential cross-attention; dx1 is d(x1s), returns d(x1)."""
        ctx, ct, ad, M, qpos, H, B = self.ctx, self.ct, self.ad, self.M, self.qpos, self.H, self.B
        Nq, d, R, dev, G, dwq, dqpos_parts = self.Nq, self.d, self.R, self.dev, self.G, self.dwq, self.dqpos_parts
        dPKV, = self.dPKV,
        i, x1 = rec["i"], rec["x1"]
        # ---------------- prompt cross-attention backward (sequential, single memory): dx1 is d(x1s) here
        pc = ctx.pcas[i]
        wp = pc.multihead_attn.in_proj_weight.detach()
        gwp, gbp = G(pc.multihead_attn.in_proj_weight), G(pc.multihead_attn.in_proj_bias)
        dx1pr, dopp = _ln_bwd(x1, [rec["opp"]], [pc.norm.weight.detach()], [pc.norm.bias.detach()], pc.norm.eps, None,
                              Nq, rec["mean_p"], rec["rstd_p"], dx1, [G(pc.norm.weight)], [G(pc.norm.bias)],
                              drop=rec["dr_pr"])
        do_p = torch.empty(B, Nq, d, dtype=ad, device=dev)
        L.gemm(M=R, N=d, K=d, A=[dopp[0]], B=[pc.multihead_attn.out_proj.weight.detach()], Cs=[do_p], ct=ct, lda=d,
               ldb=d, ldc=d, transB=True)
        dwq.add([dopp[0]], [rec["o_p"]], None, [G(pc.multihead_attn.out_proj.weight)], ct,
                [G(pc.multihead_attn.out_proj.bias)])
        dq_p = torch.empty(B, Nq, d, dtype=ad, device=dev)
        delta_p = torch.empty(B, H, Nq, dtype=torch.float32, device=dev)
        _attn(rec["qp"], ctx.PKV[i, 0], ctx.PKV[i, 1], rec["o_p"], rec["lse_p"], H, ct, True, kpm=ctx.pmask,
              bwd=(do_p, dq_p, dPKV[a, 0], dPKV[a, 1], delta_p, None), drop=rec["dr_pa"])
        gq_p = torch.empty(B, Nq, d, dtype=torch.float32, device=dev)
        dx1n = torch.empty(B, Nq, d, dtype=torch.float32, device=dev)
        L.gemm(M=R, N=d, K=d, A=[dq_p], B=[wp[:d]], Cs=[dx1n], C2=[gq_p], aux=[dx1pr], act_grad="add", ct=ct, lda=d,
               ldb=d, ldc=d, transB=True)
        dqpos_parts.append(gq_p)
        dwq.add([dq_p], [x1], [qpos], [gwp[:d]], ct, [gbp[:d]])
        dx1 = dx1n
        return dx1

    def cross_attn(self, a, rec, dx1):
        """Cross-attention over the M scene memories (one launch each for out-projection, attention, query projection):
        returns d(x_in) of the layer application; dK / dV of the application go to self.dKV[a]."""
        ctx, spec, ct, ad, M, qpos, coef = self.ctx, self.spec, self.ct, self.ad, self.M, self.qpos, self.coef
        H, B, Nq, d, Ns, R, dev = self.H, self.B, self.Nq, self.d, self.Ns, self.R, self.dev
        cas, KV, dxr_zero, G, dwq, dqpos_parts = self.cas, self.KV, self.dxr_zero, self.G, self.dwq, self.dqpos_parts
        dKV, = self.dKV,
        i, x_in = rec["i"], rec["x_in"]
        # ---------------- cross-attention backward (M memories per launch)
        cl = cas[i]
        pre = getattr(self, "_sa_chain", None)
        self._sa_chain = None
        if pre is not None:   # formed by self_attn()'s chain launch
            dxr, dop, do_all = pre
        else:
            dxr, dop = _ln_bwd(x_in, [rec["op_all"][m] for m in range(M)], [ca.norm.weight.detach() for ca in cl],
                               [ca.norm.bias.detach() for ca in cl], cl[0].norm.eps, coef[a] if coef is not None else None,
                               Nq, rec["mean_c"], rec["rstd_c"],
                               dx1, [G(ca.norm.weight) for ca in cl], [G(ca.norm.bias) for ca in cl], drop=rec["dr_cr"],
                               dx_zeroed=dxr_zero[a] if dxr_zero is not None else None)
            do_all = torch.empty(M, B, Nq, d, dtype=ad, device=dev)
            L.gemm(M=R, N=d, K=d, A=[dop[m] for m in range(M)], B=[ca.multihead_attn.out_proj.weight.detach() for ca in cl],
                   Cs=[do_all[m] for m in range(M)], ct=ct, lda=d, ldb=d, ldc=d, transB=True)
        dwq.add([dop[m] for m in range(M)], [rec["o_all"][m] for m in range(M)], None,
                [G(ca.multihead_attn.out_proj.weight) for ca in cl], ct,
                [G(ca.multihead_attn.out_proj.bias) for ca in cl])
        dq_all = torch.empty(M, B, Nq, d, dtype=ad, device=dev)
        delta_c = torch.empty(M * B, H, Nq, dtype=torch.float32, device=dev)
        mb = dict(mask=rec["attn_mask"], row_open=rec["row_open"], mask_bmod=B, mask_bits=rec.get("mask_bits")) if spec.use_self_mask \
            else dict(kpm=ctx.kpm_all)
        _attn(rec["q_all"].view(M * B, Nq, d), KV[i, 0].view(M * B, Ns, d), KV[i, 1].view(M * B, Ns, d),
              rec["o_all"].view(M * B, Nq, d), rec["lse"], H, ct, True,
              bwd=(do_all.view(M * B, Nq, d), dq_all.view(M * B, Nq, d), dKV[a, 0].view(M * B, Ns, d),
                   dKV[a, 1].view(M * B, Ns, d), delta_c, None), drop=rec["dr_ca"], drop_bmod=B, **mb)
        ws = [ca.multihead_attn.in_proj_weight.detach() for ca in cl]
        gq = torch.empty(B, Nq, d, dtype=torch.float32, device=dev)
        # the input gradient of the query projections feeds the FFN backward of the application in front of this one: when that
        # one runs as a chain launch (and no mask-head call sits in between) the launch forms it itself (step 0)
        if spec.mh is None or spec.skip_pred:
            fold = a > 0 and M <= 3 and dq_all.dtype == torch.bfloat16 and \
                self.ffn_chain_ok(self.tape[a - 1], self.layers[self.tape[a - 1]["i"]], dxr)
        else:   # a mask-head call sits in front of this application: its chain launch forms the sum (csrc/chain_mh.hip)
            fold = M <= 3 and dq_all.dtype == torch.bfloat16 and self.mask_head_chain_ok(rec, self.dcls[a], self.dmlog[a], dxr)
        if fold:
            dxn = _PendingDx(dq_all, [w[:d] for w in ws], dxr, gq)
        else:
            dxn = torch.empty(B, Nq, d, dtype=torch.float32, device=dev)
            L.gemm(M=R, N=d, K=d, A=[dq_all[m] for m in range(M)], B=[w[:d] for w in ws], Cs=[dxn] + [None] * (M - 1),
                   C2=[gq] + [None] * (M - 1), aux=[dxr] + [None] * (M - 1), act_grad="add", ct=ct, lda=d, ldb=d, ldc=d,
                   transB=True, kconcat=M)
        dqpos_parts.append(gq)
        dwq.add([dq_all[m] for m in range(M)], [x_in] * M, [qpos] * M,
                [G(ca.multihead_attn.in_proj_weight)[:d] for ca in cl], ct,
                [G(ca.multihead_attn.in_proj_bias)[:d] for ca in cl])
        dx = dxn
        return dx

    def layer(self, a, dx):
        """One layer application in reverse: FFN, self-attention, (prompt) cross-attention, the mask-head call in front of it."""
        spec, layers, Ln, dcls, dmlog, n_app, dwq = self.spec, self.layers, self.Ln, self.dcls, self.dmlog, self.n_app, self.dwq
        ready, per_layer = self.ready, self.per_layer
        rec = self.tape[a]
        i = rec["i"]
        layer = self.layers[i]
        self._cur_app = a
        if self.ffn_chain_ok(rec, layer, dx):
            dx1 = self.self_attn(rec, layer, None, pre=self.ffn_chain(rec, layer, dx))
        else:
            dx2 = self.ffn(rec, layer, dx)
            dx1 = self.self_attn(rec, layer, dx2)
        if self.spec.prompt:
            dx1 = self.prompt_cross_attn(a, rec, dx1)
        dx = self.cross_attn(a, rec, dx1)
        # ---------------- mask-head call that preceded this layer
        if spec.mh is not None and not spec.skip_pred:
            dx = self.mask_head(rec, dcls[a], dmlog[a], dx)
        # ---------------- per-layer gradient buckets (data parallel, SURVEY 8e: "bucketed per decoder layer in reverse
        # execution order"): the first-block application of layer i is the LAST to run backward, so every gradient of
        # layer i's parameters -- its queued weight-gradient products, the K/V rows of its in_proj weights (from the
        # dK / dV of all its applications) and its spatial-bias projection -- is complete once they are flushed here;
        # ready(i) lets the owner start that bucket's all-reduce while the earlier layers still run backward
        if per_layer and a < Ln:
            self.kv_terms(range(i, n_app, Ln), into_queue=True)
            dwq.flush()
            self.flush_spatial()
            ready(i)
        return dx

    def run(self):
        ctx, B, Nq, d, dev, dcls, dmlog = self.ctx, self.B, self.Nq, self.d, self.dev, self.dcls, self.dmlog
        n_app, = self.n_app,
        dxf = self.dxf
        dx = dxf.contiguous().float() if dxf is not None else torch.zeros(B, Nq, d, device=dev)
        if ctx.final_rec is not None:
            dx = self.mask_head(ctx.final_rec, dcls[-1], dmlog[-1], dx)
        for a in range(n_app - 1, -1, -1):
            dx = self.layer(a, dx)
        return self.inputs(dx)

    def inputs(self, dx):
        """After the last layer: the deferred weight-gradient flush, then the input gradients -- the hoisted K/V projections'
        backward (sum over all layer applications per source tensor), d pos, d query_pos, d prompt."""
        ctx, spec, tape, enc, ct, ad, M = self.ctx, self.spec, self.tape, self.enc, self.ct, self.ad, self.M
        U, src, mh_src, pos, params, Ln, B = self.U, self.src, self.mh_src, self.pos, self.params, self.Ln, self.B
        Nq, d, Ns, Rk, dev, gv, n_app = self.Nq, self.d, self.Ns, self.Rk, self.dev, self.gv, self.n_app
        accumulate, G, dwq, dqpos_parts, dKV, dPKV, ready = self.accumulate, self.G, self.dwq, self.dqpos_parts, self.dKV, self.dPKV, self.ready
        per_layer, = self.per_layer,
        # ---- hoisted K/V projection backward (sum over all applications)
        need_feat = [ctx.needs_input_grad[11 + u] for u in range(U)]
        dfeats: List[Optional[torch.Tensor]] = [None] * U
        single = U == M and all(src[i][j] == j for i in range(Ln) for j in range(M)) and mh_src[:M] == list(range(M))[:len(mh_src)]
        Akv, Bkv = self.kv_terms(range(n_app), into_queue=not per_layer)
        # ---- every parameter gradient of the decoder (+ mask head) is complete after this flush: in a data-parallel step
        # its all-reduce starts HERE (enc.grads_ready, set by the step owner) and overlaps the key/value input-gradient
        # products below and the encoders' backward that autograd runs after this function returns
        dkm_list = {}
        if self.dk_terms:
            Mm_, nc = spec.mh_count, len(self.dk_terms)
            newk = torch.empty(Mm_, B, Ns, d, dtype=torch.float32, device=dev)
            L.gemm(M=Ns, N=d, K=Nq, A=[t_[0] for m in range(Mm_) for t_ in self.dk_terms],
                   B=[t_[1][m] for m in range(Mm_) for t_ in self.dk_terms],
                   Cs=[c_ for m in range(Mm_) for c_ in [newk[m]] + [None] * (nc - 1)],
                   aux=[c_ for m in range(Mm_) for c_ in [self.dkeys[m]] + [None] * (nc - 1)] if self.dkeys is not None else None,
                   act_grad="add" if self.dkeys is not None else None, ct=ct, lda=Nq, ldb=d, ldc=d, transB=True, batch=B,
                   strideA=Ns * Nq, strideB=Nq * d, strideC=Ns * d, kconcat=nc)
            self.dkeys, self.dk_terms = newk, []
        if self.dkeys is not None:
            for j in range(spec.mh_count):
                mp = list(spec.mh.mask_pred_list)[j]
                dkm_list[j] = ops.scale_rows(self.dkeys[j], Rk, ad, keep_mask=ctx.mh_valid[j])
                dwq.add([dkm_list[j]], [ctx.mh_feats[j]], None, [G(mp.k_proj.weight)], ct)
        dwq.flush()
        self.flush_spatial()
        ready("decoder")   # every parameter gradient of the decoder (+ mask head) is final
        # bf16 path: the input-gradient products read TRANSPOSED bf16 copies of the K/V weights (one copy launch from the
        # forward's pre-cast rows), which turns them into plain NT products -- the 128x128-tile kernel's layout -- and the
        # memories then share launches (K-concatenation per memory, several outputs per launch)
        tposed = ctx.wkv is not None and dKV.dtype == torch.bfloat16
        if tposed:
            wkvT = ctx.wkvT if ctx.wkvT is not None else \
                ctx.wkv.view(Ln, M, 2, d, d).transpose(-1, -2).contiguous()       # [l, m, t][k_in][n_out]
            Bkv = [wkvT[tape[a]["i"], j, t] for a in range(n_app) for j in range(M) for t in (0, 1)]
            kT = None
            if self.dkeys is not None:
                # transposed bf16 copies of the mask-head key projections: one launch (pq3d_cast_transpose; rows = cols = d)
                kws = [mp.k_proj.weight.detach() for mp in list(spec.mh.mask_pred_list)[:spec.mh_count]]
                if d % 32 == 0 and ad == torch.bfloat16 and len(kws) <= MAXG and all(w_.is_contiguous() and w_.shape == (d, d) for w_ in kws):
                    kT, kS = torch.empty(2, len(kws), d, d, dtype=ad, device=dev)
                    arr = lambda ts: (C.c_void_p * len(ts))(*[L.ptr(t) for t in ts])
                    L.check(L.lib().pq3d_cast_transpose(arr(kws), arr([kS[m_] for m_ in range(len(kws))]),
                                                        arr([kT[m_] for m_ in range(len(kws))]), len(kws), d, d, L.stream()),
                            "pq3d_cast_transpose")
                else:
                    kT = torch.stack(kws).transpose(1, 2).contiguous().to(ad)
        # d source_u = sum over the (application, memory) pairs that read it of (dK Wk + dV Wv) [+ mask-head key path]
        jobs = []
        for u in range(U):
            if not need_feat[u]:
                continue
            pairs = [(a, j) for a in range(n_app) for j in range(M) if src[tape[a]["i"]][j] == u]
            Aj = [Akv[2 * (a * M + j) + t] for a, j in pairs for t in (0, 1)]
            Bj = [Bkv[2 * (a * M + j) + t] for a, j in pairs for t in (0, 1)]
            for j in range(spec.mh_count if self.dkeys is not None else 0):
                if mh_src[j] != u:
                    continue
                mp = list(spec.mh.mask_pred_list)[j]
                Aj.append(dkm_list[j])
                Bj.append(kT[j] if tposed else mp.k_proj.weight.detach())
            jobs.append((u, Aj, Bj))
        per = len(jobs[0][1]) if jobs else 0
        want_dpos = pos is not None and ctx.needs_input_grad[4]
        dqpos_done = None
        dpos = None
        nk, nv = n_app, n_app + (1 if self.dkeys is not None else 0)
        if single and tposed and want_dpos and len(jobs) == M and nk * M <= MAXG and nv * M <= MAXG and \
                (self.dkeys is None or spec.mh_count == M):
            # the position embedding enters every memory's KEY input, so d pos = sum_m (K part of d feat_m): form the K
            # parts once (one launch, M outputs), add them into the V parts through the "+ aux" epilogue (second launch)
            # and sum them for d pos -- instead of a third product over all (layer, memory) key terms
            Kp = torch.empty(M, B, Ns, d, dtype=torch.float32, device=dev)
            L.gemm(M=Rk, N=d, K=d, A=[jb[1][2 * a] for jb in jobs for a in range(n_app)],
                   B=[jb[2][2 * a] for jb in jobs for a in range(n_app)],
                   Cs=[c_ for j in range(M) for c_ in [Kp[j]] + [None] * (nk - 1)], ct=ct, lda=d, ldb=d, ldc=d, kconcat=nk)
            outs = [torch.empty(B, Ns, d, dtype=torch.float32, device=dev) for _ in range(M)]
            vA = [[jb[1][2 * a + 1] for a in range(n_app)] + jb[1][2 * n_app:] for jb in jobs]
            vB = [[jb[2][2 * a + 1] for a in range(n_app)] + jb[2][2 * n_app:] for jb in jobs]
            L.gemm(M=Rk, N=d, K=d, A=[t_ for l_ in vA for t_ in l_], B=[t_ for l_ in vB for t_ in l_],
                   Cs=[c_ for o_ in outs for c_ in [o_] + [None] * (nv - 1)],
                   aux=[c_ for j in range(M) for c_ in [Kp[j]] + [None] * (nv - 1)], act_grad="add", ct=ct, lda=d, ldb=d,
                   ldc=d, kconcat=nv)
            for jb, o_ in zip(jobs, outs):
                dfeats[jb[0]] = o_
            if ctx.needs_input_grad[2] and len(dqpos_parts) > 1:   # d query_pos and d pos: one launch, adjacent outputs
                dqpos_done, dpos = ops.sum_pair(dqpos_parts, [Kp[j] for j in range(M)])
            else:
                dpos = ops.sum_n([Kp[j] for j in range(M)])
            want_dpos = False
        elif tposed and jobs and 0 < per <= MAXG and all(len(jb[1]) == per for jb in jobs):
            cap = max(1, MAXG // per)   # memories per launch
            for c0 in range(0, len(jobs), cap):
                chunk = jobs[c0:c0 + cap]
                outs = [torch.empty(B, Ns, d, dtype=torch.float32, device=dev) for _ in chunk]
                L.gemm(M=Rk, N=d, K=d, A=[t_ for jb in chunk for t_ in jb[1]], B=[t_ for jb in chunk for t_ in jb[2]],
                       Cs=[c_ for o_ in outs for c_ in [o_] + [None] * (per - 1)], ct=ct, lda=d, ldb=d, ldc=d,
                       kconcat=per)
                for jb, o_ in zip(chunk, outs):
                    dfeats[jb[0]] = o_
        else:
            for j, Aj, Bj in jobs:
                out = None
                if not Aj:   # a source no layer reads (e.g. a surplus scale): zero gradient
                    out = torch.zeros(B, Ns, d, dtype=torch.float32, device=dev)
                for s in range(0, len(Aj), MAXG):
                    nxt = torch.empty(B, Ns, d, dtype=torch.float32, device=dev)
                    n = len(Aj[s:s + MAXG])
                    L.gemm(M=Rk, N=d, K=d, A=Aj[s:s + MAXG], B=Bj[s:s + MAXG], Cs=[nxt] + [None] * (n - 1),
                           aux=([out] + [None] * (n - 1)) if out is not None else None,
                           act_grad="add" if out is not None else None, ct=ct, lda=d, ldb=d, ldc=d, transB=not tposed,
                           kconcat=n)
                    out = nxt
                dfeats[j] = out
        if want_dpos:
            Ak, Bk = Akv[0::2], Bkv[0::2]
            for s in range(0, len(Ak), MAXG):
                nxt = torch.empty(B, Ns, d, dtype=torch.float32, device=dev)
                n = len(Ak[s:s + MAXG])
                L.gemm(M=Rk, N=d, K=d, A=Ak[s:s + MAXG], B=Bk[s:s + MAXG], Cs=[nxt] + [None] * (n - 1),
                       aux=([dpos] + [None] * (n - 1)) if dpos is not None else None,
                       act_grad="add" if dpos is not None else None, ct=ct, lda=d, ldb=d, ldc=d, transB=not tposed,
                       kconcat=n)
                dpos = nxt
        dqpos = dqpos_done
        if ctx.needs_input_grad[2] and dqpos is None:
            dqpos = ops.sum_n(dqpos_parts)
        dx0 = dx if ctx.needs_input_grad[1] else None
        dprompt = None
        if spec.prompt and ctx.needs_input_grad[9]:
            # d prompt = sum over layer applications of dK_p Wk + dV_p Wv: one K-concatenated launch
            Ap, Bp = [], []
            for a_ in range(n_app):
                w = ctx.pcas[tape[a_]["i"]].multihead_attn.in_proj_weight.detach()
                Ap += [dPKV[a_, 0], dPKV[a_, 1]]
                Bp += [w[d:2 * d], w[2 * d:]]
            T = ctx.prompt.shape[1]
            for s_ in range(0, len(Ap), MAXG):
                nxt = torch.empty(B, T, d, dtype=torch.float32, device=dev)
                n = len(Ap[s_:s_ + MAXG])
                L.gemm(M=B * T, N=d, K=d, A=Ap[s_:s_ + MAXG], B=Bp[s_:s_ + MAXG], Cs=[nxt] + [None] * (n - 1),
                       aux=([dprompt] + [None] * (n - 1)) if dprompt is not None else None,
                       act_grad="add" if dprompt is not None else None, ct=ct, lda=d, ldb=d, ldc=d, transB=True, kconcat=n)
                dprompt = nxt
        pgrads = [gv[id(p)] if (p.requires_grad and not accumulate) else None for p in params]
        return (None, dx0, dqpos, None, dpos, None, None, None, None, dprompt, None, *dfeats, *([None] * M), *pgrads)


class _FusedDecoder(Function):
    """inputs: spec, x0, qpos, qmask, pos, pairwise_locs, seg_pad, offline_mask, coef, prompt, prompt_mask, U feats,
    M masks, *params."""

    @staticmethod
    def forward(ctx, spec: FusedSpec, x0, qpos, qmask, pos, pl, seg_pad, offline_mask, coef, prompt, prompt_kpm, *rest):
        enc, ct = spec.enc, spec.ct
        ad = ops.act_dtype(ct)
        M, U = len(spec.mems), spec.n_src
        feats, masks, params = list(rest[:U]), list(rest[U:U + M]), rest[U + M:]   # feats: the U unique sources
        layers = list(enc.unified_encoder)
        Ln = len(layers)
        src = spec.src if spec.src is not None else [list(range(M)) for _ in range(Ln)]
        mh_src = spec.mh_src if spec.mh_src is not None else list(range(M))
        H = enc.num_heads
        B, Nq, d = qpos.shape
        Ns = feats[0].shape[1]
        R, Rk = B * Nq, B * Ns
        dev = qpos.device
        mem_idx = [layers[0].memories.index(m) for m in spec.mems]
        cas = [[layers[i].cross_attn_list[j] for j in mem_idx] for i in range(Ln)]
        pcas = [layers[i].memory2ca["prompt"] for i in range(Ln)] if spec.prompt else None
        cq = ops.small_ct(ct)   # query-side GEMMs (M = B*N_q rows): split-bf16 in 'bf16' mode, exact f32 otherwise
        x0, qpos, pos = ops._c(x0), ops._c(qpos), ops._c(pos)
        feats = [ops._c(f) for f in feats]
        mh_feats = [feats[u] for u in mh_src]
        masks = [ops._c(m) for m in masks]
        qmask = ops._c(qmask)

        # ---- layer-invariant MFMA operands, rounded once: kin_m = (feat_m + pos), vin_m = feat_m in the activation
        # dtype (bf16 path: the 2*L*M hoisted GEMMs and their weight-gradient GEMM then read 2 B/element, not 8)
        kv3 = spec.kv3   # compute mode 'bf16x3': split-bf16 key/value side (hi + lo bf16 planes), see fused_decoder()
        kvin_lo = None
        if ct == BF16 and (B * Ns * d) % 8 == 0 and 2 * U <= MAXG:
            kvin = torch.empty(2, U, B, Ns, d, dtype=ad, device=dev)
            srcs = [feats[u] for u in range(U)] * 2
            adds = [pos] * U + [None] * U
            outs = [kvin[0, u] for u in range(U)] + [kvin[1, u] for u in range(U)]
            if pos is None:
                adds = [None] * (2 * U)
            arr = lambda ts: (C.c_void_p * len(ts))(*[L.ptr(t) for t in ts])
            if kv3:   # both planes of (feat + pos) and feat in one launch; the K / V weights' residual planes join it below
                kvin_lo = torch.empty(2, U, B, Ns, d, dtype=ad, device=dev)
                f_srcs, f_adds, f_outs = srcs, adds, outs
            else:
                L.check(L.lib().pq3d_add_cast(arr(srcs), arr(adds), arr(outs), 2 * U, L.BF16, B * Ns * d, L.stream()),
                        "pq3d_add_cast")
            kin, vin, kin2 = [kvin[0, u] for u in range(U)], [kvin[1, u] for u in range(U)], [None] * U
        else:
            kin, vin, kin2 = feats, feats, [pos] * U
        ctx.kin, ctx.vin, ctx.kin2 = kin, vin, kin2
        ctx.wkv = None
        # ---- hoisted K/V projections: KV[l, 0|1, m] = (feat_m [+ pos]) @ W{k,v}_{l,m}^T + b
        KV = torch.empty(Ln, 2, M, B, Ns, d, dtype=ad, device=dev)
        # bf16 path: the K/V rows of every in_proj_weight are rounded ONCE (one launch) instead of by each of the M/64 row
        # tiles that read them; with both operands in bf16 the projection takes the 128x128-tile kernel (gemm128.hip),
        # whose output is bit-identical to converting in flight
        wkv = wkvT = None
        if kin2[0] is None and ct == BF16 and Ln * M <= MAXG and (2 * d * d) % 8 == 0:
            wkv = torch.empty(Ln, M, 2 * d, d, dtype=ad, device=dev)
            srcs = [ca.multihead_attn.in_proj_weight.detach()[d:] for i in range(Ln) for ca in cas[i]]
            outs = [wkv[i, j] for i in range(Ln) for j in range(M)]
            arr = lambda ts: (C.c_void_p * len(ts))(*[L.ptr(t) for t in ts])
            if d % 32 == 0 and any(ctx.needs_input_grad):
                # the same launch also leaves the transposed blocks [l, m, t][k_in][n_out] the backward's input-gradient
                # products read (plain NT products on W^T: the 128x128-tile kernel's layout)
                wkvT = torch.empty(Ln, M, 2, d, d, dtype=ad, device=dev)
                L.check(L.lib().pq3d_cast_transpose(arr(srcs), arr(outs), arr([wkvT[i, j] for i in range(Ln) for j in range(M)]),
                                                    len(srcs), 2 * d, d, L.stream()), "pq3d_cast_transpose")
            else:
                L.check(L.lib().pq3d_add_cast(arr(srcs), arr([None] * len(srcs)), arr(outs), len(srcs), L.BF16, 2 * d * d,
                                              L.stream()), "pq3d_add_cast")
        ctx.wkv, ctx.wkvT = wkv, wkvT
        KV_lo = None
        if kv3:
            # split-bf16 projection of pre-split operands (csrc/gemm_x3p.hip: the four planes of a k slice staged once, lo.hi + hi.lo
            # + hi.hi per term pair), its fp32-grade result leaving as hi / lo bf16 planes (PQ3D_ACT_PLANES): KV = exactly the
            # 'bf16'-mode tensor the backward reads, KV_lo the residual the forward's split-bf16 attention adds (csrc/attn_x3.hip)
            assert wkv is not None and kvin_lo is not None
            wkv_lo = torch.empty_like(wkv)
            ops.split_planes(f_srcs + [ca.multihead_attn.in_proj_weight.detach()[d:] for i in range(Ln) for ca in cas[i]],
                             f_adds + [None] * (Ln * M), f_outs + [None] * (Ln * M),
                             [kvin_lo[0, u] for u in range(U)] + [kvin_lo[1, u] for u in range(U)] +
                             [wkv_lo[i, j] for i in range(Ln) for j in range(M)])
            KV_lo = torch.empty_like(KV)
            A, A2, Bw, B2, bs, Cs, C2 = [], [], [], [], [], [], []
            for i in range(Ln):
                for j, ca in enumerate(cas[i]):
                    b = ca.multihead_attn.in_proj_bias.detach()
                    u = src[i][j]
                    for t in (0, 1):
                        A.append(kvin[t, u]); A2.append(kvin_lo[t, u])
                        Bw.append(wkv[i, j, t * d:(t + 1) * d]); B2.append(wkv_lo[i, j, t * d:(t + 1) * d])
                        bs.append(b[(1 + t) * d:(2 + t) * d])
                        Cs.append(KV[i, t, j]); C2.append(KV_lo[i, t, j])
            for s in range(0, len(A), MAXG):
                sl = slice(s, s + MAXG)
                L.gemm(M=Rk, N=d, K=d, A=A[sl], A2=A2[sl], B=Bw[sl], B2=B2[sl], bias=bs[sl], Cs=Cs[sl], C2=C2[sl],
                       ct=L.BF16X3, lda=d, ldb=d, ldc=d, act_grad="planes")
        else:
            A, A2, Bw, bs, Cs = [], [], [], [], []
            for i in range(Ln):
                for j, ca in enumerate(cas[i]):
                    w, b = ca.multihead_attn.in_proj_weight.detach(), ca.multihead_attn.in_proj_bias.detach()
                    A += [kin[src[i][j]], vin[src[i][j]]]
                    A2 += [kin2[src[i][j]], None]
                    Bw += [w[d:2 * d], w[2 * d:]] if wkv is None else [wkv[i, j, :d], wkv[i, j, d:]]
                    bs += [b[d:2 * d], b[2 * d:]]
                    Cs += [KV[i, 0, j], KV[i, 1, j]]
            for s in range(0, len(A), MAXG):
                L.gemm(M=Rk, N=d, K=d, A=A[s:s + MAXG], A2=A2[s:s + MAXG], B=Bw[s:s + MAXG], bias=bs[s:s + MAXG],
                       Cs=Cs[s:s + MAXG], ct=ct, lda=d, ldb=d, ldc=d)
        # ---- structure 'mixed' (query_encoder.py:162-165): the prompt memory's K / V of every layer, hoisted like the scene
        # memories' (the prompt is layer-invariant, pos = None: query3d_unified.py:134-136): one grouped launch
        PKV = None
        if spec.prompt:
            prompt, prompt_kpm = ops._c(prompt), ops._c(prompt_kpm)
            T = prompt.shape[1]
            # compute mode 'bf16x3': the (short) prompt memory's cross-attention runs at fp32 grade on kernels that exist -- split-bf16
            # projections with fp32 K / V, the exact-f32 attention -- and leaves bf16 copies as the tape of the single-bf16 backward
            PKV_f = torch.empty(Ln, 2, B, T, d, dtype=torch.float32, device=dev) if kv3 else None
            PKV = torch.empty(Ln, 2, B, T, d, dtype=ad, device=dev) if not kv3 else None
            Ap, Bp, bp, Cp = [], [], [], []
            for i in range(Ln):
                w, b = pcas[i].multihead_attn.in_proj_weight.detach(), pcas[i].multihead_attn.in_proj_bias.detach()
                Ap += [prompt, prompt]; Bp += [w[d:2 * d], w[2 * d:]]; bp += [b[d:2 * d], b[2 * d:]]
                Cp += [(PKV_f if kv3 else PKV)[i, 0], (PKV_f if kv3 else PKV)[i, 1]]
            for s_ in range(0, len(Ap), MAXG):
                L.gemm(M=B * T, N=d, K=d, A=Ap[s_:s_ + MAXG], B=Bp[s_:s_ + MAXG], bias=bp[s_:s_ + MAXG], Cs=Cp[s_:s_ + MAXG],
                       ct=cq if kv3 else ct, lda=d, ldb=d, ldc=d)
            if kv3:
                PKV = ops.cast_bf16([PKV_f])[0]
        kpm_all = None
        if not spec.use_self_mask:
            st = spec.stacked_kpm   # [M, B, Ns] already stacked by the model (same memory order): no copy
            kpm_all = st.reshape(M * B, Ns) if st is not None else torch.cat(masks, 0)

        # ---- mask-head keys (layer-invariant)
        keys = inv_den = None
        if spec.mh is not None:
            mps = list(spec.mh.mask_pred_list)[:spec.mh_count]
            valid = ops.mask_not(masks[:spec.mh_count])   # one launch
            cq = ops.small_ct(ct)   # fp32-grade keys (split-bf16): see MaskHeadSegLevel.project_keys
            keys_buf = torch.empty(spec.mh_count, B, Ns, d, dtype=ops.act_dtype(cq), device=dev)
            L.gemm(M=Rk, N=d, K=d, A=mh_feats[:spec.mh_count], B=[mp.k_proj.weight.detach() for mp in mps],
                   Cs=[keys_buf[m] for m in range(spec.mh_count)], row_mask=valid, ct=cq, lda=d, ldb=d, ldc=d)
            keys = [keys_buf[m] for m in range(spec.mh_count)]
            inv_den = ops.mask_inv_den(masks[:spec.mh_count])
            ctx.mh_valid = valid

        # ---- spatial attention bias log(clamp(relu(W_l . pairwise_locs))) of every layer: one grouped launch (depends on
        # the layer's weights only, not on the query state, and is shared by the blocks that re-traverse the layers)
        sbias_all = None
        if spec.spatial:
            sbias_all = torch.empty(Ln, B, H, Nq, Nq, dtype=torch.float32, device=dev)
            fcs = [layers[i].self_attn.self_attn.pairwise_loc_fc for i in range(Ln)]
            for i0 in range(0, Ln, MAXG):
                n_ = min(MAXG, Ln - i0)
                arr = lambda ts: (C.c_void_p * len(ts))(*[L.ptr(t) for t in ts])
                L.check(L.lib().pq3d_spatial_bias_fwd_grouped(
                    L.ptr(pl), arr([fc.weight.detach() for fc in fcs[i0:i0 + n_]]),
                    arr([fc.bias.detach() for fc in fcs[i0:i0 + n_]]), arr([sbias_all[i0 + k] for k in range(n_)]), n_, B, H,
                    Nq, L.stream()), "pq3d_spatial_bias_fwd_grouped")
        tape: List[dict] = []
        pcls, pmask = [], []
        x = x0
        attn_mask = row_open = None
        q_next = None   # the next application's cross-attention queries when the chain launch of this one formed them
        for blk in range(spec.num_blocks):
            for i, layer in enumerate(layers):
                app = blk * Ln + i
                rec: Dict[str, object] = {"i": i, "x_in": x}
                # dropout sites of this layer application (None when dropout is off); memories are stacked along the
                # attention batch in groups of B -> drop_bmod=B gives memory j the site of slot j
                dr_ca = spec.drop(cas[i][0], app, ops.DROP_CA_ATTN, dev)
                dr_cr = spec.drop(cas[i][0], app, ops.DROP_CA_RES, dev)
                dr_sa = None if spec.spatial else spec.drop(layer.self_attn, app, ops.DROP_SA_ATTN, dev)
                dr_sr = spec.drop(layer.self_attn, app, ops.DROP_SA_RES, dev)
                dr_fi = spec.drop(layer.ffn, app, ops.DROP_FFN_INNER, dev)
                dr_fr = spec.drop(layer.ffn, app, ops.DROP_FFN_RES, dev)
                rec.update(dr_ca=dr_ca, dr_cr=dr_cr, dr_sa=dr_sa, dr_sr=dr_sr, dr_fi=dr_fi, dr_fr=dr_fr)
                if spec.mh is not None and not spec.skip_pred:
                    cls, mlog, amask = _mh_forward(spec, x, keys, inv_den, seg_pad, rec, app)
                    pcls.append(cls)
                    pmask.append(mlog)
                    attn_mask = offline_mask if spec.offline else amask
                elif spec.offline:
                    attn_mask = offline_mask
                mask_bits = None
                if spec.use_self_mask:
                    # row-open flags AND the mask as bit words (open rows cleared) from one pass over the bytes: the
                    # resident backward reads 1/8 of the mask and no byte tiles (attn_resident.hip MASK3 == 2)
                    if ct == BF16:
                        row_open, mask_bits = ops.mask_pack(attn_mask)
                    else:
                        row_open = ops.mask_row_all(attn_mask)
                rec["attn_mask"], rec["row_open"], rec["mask_bits"] = attn_mask, row_open, mask_bits
                # -- cross attention over the M memories: 4 launches
                if q_next is not None:   # formed by the previous layer application's chain launch (csrc/chain_ffn.hip, step 6)
                    q_all, q_next = q_next, None
                else:
                    q_all = torch.empty(M, B, Nq, d, dtype=torch.float32 if kv3 else ad, device=dev)
                    ws = [ca.multihead_attn.in_proj_weight.detach() for ca in cas[i]]
                    bsl = [ca.multihead_attn.in_proj_bias.detach() for ca in cas[i]]
                    L.gemm(M=R, N=d, K=d, A=[x] * M, A2=[qpos] * M, B=[w[:d] for w in ws], bias=[b[:d] for b in bsl],
                           Cs=[q_all[m] for m in range(M)], ct=cq, lda=d, ldb=d, ldc=d)
                o_all = torch.empty(M, B, Nq, d, dtype=ad, device=dev)
                lse = torch.empty(M * B, H, Nq, dtype=torch.float32, device=dev)
                o_f32 = None
                if kv3:
                    # split-bf16 cross-attention (csrc/attn_x3.hip): fp32 q in, fp32 o out (-> the split-bf16 out-projection); the
                    # bf16 copies it leaves of q and o are what the (single-bf16) backward reads -- a 'bf16'-mode tape
                    q_f32, q_all = q_all, torch.empty(M, B, Nq, d, dtype=ad, device=dev)
                    o_f32 = torch.empty(M, B, Nq, d, dtype=torch.float32, device=dev)
                    mkw = dict(mask=attn_mask, row_open=row_open, mask_bmod=B, mask_bits=mask_bits) if spec.use_self_mask \
                        else dict(kpm=kpm_all)
                    _attn(q_f32.view(M * B, Nq, d), KV[i, 0].view(M * B, Ns, d), KV[i, 1].view(M * B, Ns, d),
                          o_f32.view(M * B, Nq, d), lse, H, L.BF16X3, True, drop=dr_ca, drop_bmod=B,
                          planes=(KV_lo[i, 0], KV_lo[i, 1], q_all, o_all), **mkw)
                elif spec.use_self_mask:
                    _attn(q_all.view(M * B, Nq, d), KV[i, 0].view(M * B, Ns, d), KV[i, 1].view(M * B, Ns, d),
                          o_all.view(M * B, Nq, d), lse, H, ct, True, mask=attn_mask, row_open=row_open, mask_bmod=B,
                          drop=dr_ca, drop_bmod=B, mask_bits=mask_bits)
                else:
                    _attn(q_all.view(M * B, Nq, d), KV[i, 0].view(M * B, Ns, d), KV[i, 1].view(M * B, Ns, d),
                          o_all.view(M * B, Nq, d), lse, H, ct, True, kpm=kpm_all, drop=dr_ca, drop_bmod=B)
                sa = layer.self_attn
                if spec.spatial:
                    msa = sa.self_attn
                    Wl = [msa.w_qs.weight.detach(), msa.w_ks.weight.detach(), msa.w_vs.weight.detach()]
                    bl = [msa.w_qs.bias.detach(), msa.w_ks.bias.detach(), msa.w_vs.bias.detach()]
                    Wo, bo = msa.fc.weight.detach(), msa.fc.bias.detach()
                else:
                    w, b = sa.self_attn.in_proj_weight.detach(), sa.self_attn.in_proj_bias.detach()
                    Wl, bl = [w[:d], w[d:2 * d], w[2 * d:]], [b[:d], b[d:2 * d], b[2 * d:]]
                    Wo, bo = sa.self_attn.out_proj.weight.detach(), sa.self_attn.out_proj.bias.detach()
                # out-projections + merged post-norm + the self-attention's q / k / v projections: ONE launch when the shapes allow
                # (csrc/chain_ca.hip: same bits as the three launches below)
                chain_ca = _chain_on(dev) and ct == BF16 and cq == L.BF16X3 and not spec.prompt and dr_cr is None and \
                    ops.chain_ca_ok(R, d, M) and o_all.dtype == torch.bfloat16
                qkv = None
                if chain_ca:
                    flags = getattr(enc, "_chain_flags_ca", None)
                    if flags is None or flags.device != dev:
                        flags = enc._chain_flags_ca = ops.chain_flags(2048, dev)
                    op_all, x1, mean_c, rstd_c, qkv = ops.chain_ca_fwd(
                        o_f32 if kv3 else o_all, [ca.multihead_attn.out_proj.weight.detach() for ca in cas[i]],
                        [ca.multihead_attn.out_proj.bias.detach() for ca in cas[i]], x, [ca.norm.weight.detach() for ca in cas[i]],
                        [ca.norm.bias.detach() for ca in cas[i]], cas[i][0].norm.eps, coef[app] if coef is not None else None, Nq, qpos,
                        [t_.contiguous() for t_ in Wl], [t_.contiguous() for t_ in bl], flags)
                else:
                    op_all = torch.empty(M, B, Nq, d, dtype=torch.float32, device=dev)
                    L.gemm(M=R, N=d, K=d, A=[(o_f32 if kv3 else o_all)[m] for m in range(M)],
                           B=[ca.multihead_attn.out_proj.weight.detach() for ca in cas[i]],
                           bias=[ca.multihead_attn.out_proj.bias.detach() for ca in cas[i]],
                           Cs=[op_all[m] for m in range(M)], ct=cq if kv3 else ct, lda=d, ldb=d, ldc=d)
                    x1, mean_c, rstd_c = _ln_fwd(x, [op_all[m] for m in range(M)], [ca.norm.weight.detach() for ca in cas[i]],
                                                 [ca.norm.bias.detach() for ca in cas[i]], cas[i][0].norm.eps,
                                                 coef[app] if coef is not None else None, Nq, drop=dr_cr)
                rec.update(q_all=q_all, o_all=o_all, lse=lse, op_all=op_all, mean_c=mean_c, rstd_c=rstd_c, x1=x1)
                x1s = x1     # input of the self-attention sublayer
                if spec.prompt:
                    # -- sequential prompt cross-attention on the parallel block's output (CrossAttentionLayer.forward_post,
                    # query_encoder.py:288-307: q = x1 + query_pos, k = v = prompt, own LayerNorm): 4 launches
                    pc = pcas[i]
                    dr_pa = spec.drop(pc, app, ops.DROP_CA_ATTN, dev, m=4)   # sequential slot 4 + 0 (modules.QueryEncoderLayer)
                    dr_pr = spec.drop(pc, app, ops.DROP_CA_RES, dev, m=4)
                    wp, bpq = pc.multihead_attn.in_proj_weight.detach(), pc.multihead_attn.in_proj_bias.detach()
                    adp = torch.float32 if kv3 else ad
                    qp = torch.empty(B, Nq, d, dtype=adp, device=dev)
                    L.gemm(M=R, N=d, K=d, A=[x1], A2=[qpos], B=[wp[:d]], bias=[bpq[:d]], Cs=[qp], ct=cq, lda=d, ldb=d, ldc=d)
                    o_p = torch.empty(B, Nq, d, dtype=adp, device=dev)
                    lse_p = torch.empty(B, H, Nq, dtype=torch.float32, device=dev)
                    _attn(qp, (PKV_f if kv3 else PKV)[i, 0], (PKV_f if kv3 else PKV)[i, 1], o_p, lse_p, H, L.F32 if kv3 else ct, True,
                          kpm=prompt_kpm, drop=dr_pa)
                    opp = torch.empty(B, Nq, d, dtype=torch.float32, device=dev)
                    L.gemm(M=R, N=d, K=d, A=[o_p], B=[pc.multihead_attn.out_proj.weight.detach()],
                           bias=[pc.multihead_attn.out_proj.bias.detach()], Cs=[opp], ct=cq if kv3 else ct, lda=d, ldb=d, ldc=d)
                    if kv3:   # the backward's bf16 tape
                        qp, o_p = ops.cast_bf16([qp, o_p])
                    x1s, mean_p, rstd_p = _ln_fwd(x1, [opp], [pc.norm.weight.detach()], [pc.norm.bias.detach()], pc.norm.eps,
                                                  None, Nq, drop=dr_pr)
                    rec.update(qp=qp, o_p=o_p, lse_p=lse_p, opp=opp, mean_p=mean_p, rstd_p=rstd_p, dr_pa=dr_pa, dr_pr=dr_pr)
                rec["x1s"] = x1s
                # -- self attention: 5 launches (spatial) / 4
                # N_q x N_q scores per scene: projections at fp32 grade, attention core on the exact-f32 MFMA path
                if qkv is None:
                    qkv = torch.empty(3, B, Nq, d, dtype=torch.float32, device=dev)
                    L.gemm(M=R, N=d, K=d, A=[x1s] * 3, A2=[qpos, qpos, None], B=Wl, bias=bl, Cs=[qkv[0], qkv[1], qkv[2]], ct=cq,
                           lda=d, ldb=d, ldc=d)
                sbias = sbias_all[i] if spec.spatial else None   # layer-invariant across blocks: computed once above
                o_s = torch.empty(B, Nq, d, dtype=torch.float32, device=dev)
                lse_s = torch.empty(B, H, Nq, dtype=torch.float32, device=dev)
                ffn = layer.ffn
                F_ = ffn.linear1.out_features
                chain = _chain_on(dev) and cq == L.BF16X3 and spec.act == "relu" and dr_sr is None and dr_fi is None and dr_fr is None and \
                    ops.chain_ffn_ok(R, d, F_) and sa.norm.weight.shape[0] == d
                # the self-attention core as step 0 of that launch (one launch less per layer; same bits as pq3d_attn_fwd's kernel)
                sa_in = chain and dr_sa is None and ops.sa_ct(ct) == L.BF16X3 and ops.chain_sa_ok(Nq, H, d) and qkv.is_contiguous() and \
                    (sbias is None or (sbias.is_contiguous() and sbias.dtype == torch.float32)) and \
                    (qmask is None or (qmask.is_contiguous() and qmask.dtype == torch.bool))
                if not sa_in:
                    _attn(qkv[0], qkv[1], qkv[2], o_s, lse_s, H, ops.sa_ct(ct), False, kpm=qmask, bias=sbias, drop=dr_sa)
                if chain:
                    # the row-local tail of the layer in ONE launch (csrc/chain_ffn.hip: out-projection, post-norm, FFN, post-norm;
                    # 8 workgroups per 32-row tile handing rows over inside one XCD) -- the five launches' bits
                    flags = getattr(enc, "_chain_flags", None)
                    if flags is None or flags.device != dev or flags.numel() < ((R + 31) // 32) * 128:
                        flags = enc._chain_flags = ops.chain_flags(max(R, 2048), dev)
                    # ... and the NEXT application's cross-attention query projections (they read this application's output): the
                    # mask head in front of the next layer reads x3, not the queries, so nothing else moves
                    nextq = None
                    if app + 1 < spec.num_blocks * Ln and M <= 3 and ad == torch.bfloat16:
                        cn = cas[(i + 1) % Ln]
                        nextq = (qpos, [ca.multihead_attn.in_proj_weight.detach()[:d] for ca in cn],
                                 [ca.multihead_attn.in_proj_bias.detach()[:d] for ca in cn])
                    outs = ops.chain_ffn_fwd(
                        o_s, Wo, bo, x1s, sa.norm.weight.detach(), sa.norm.bias.detach(), sa.norm.eps,
                        ffn.linear1.weight.detach(), ffn.linear1.bias.detach(), ffn.linear2.weight.detach(), ffn.linear2.bias.detach(),
                        ffn.norm.weight.detach(), ffn.norm.bias.detach(), ffn.norm.eps, flags, nextq=nextq,
                        q_dtype=torch.float32 if kv3 else torch.bfloat16,
                        sa=(qkv[0], qkv[1], qkv[2], sbias, qmask, lse_s, 1.0 / math.sqrt(d // H)) if sa_in else None)
                    f, x2, mean_s, rstd_s, h, _zp, z, x3, mean_f, rstd_f = outs[:10]
                    q_next = outs[10] if nextq is not None else None
                    pre = None
                    rec.update(qkv=qkv, sbias=sbias, o_s=o_s, lse_s=lse_s, f=f, mean_s=mean_s, rstd_s=rstd_s, x2=x2)
                else:
                    f = torch.empty(B, Nq, d, dtype=torch.float32, device=dev)
                    L.gemm(M=R, N=d, K=d, A=[o_s], B=[Wo], bias=[bo], Cs=[f], ct=cq, lda=d, ldb=d, ldc=d)
                    x2, mean_s, rstd_s = _ln_fwd(x1s, [f], [sa.norm.weight.detach()], [sa.norm.bias.detach()], sa.norm.eps, None, Nq,
                                                 drop=dr_sr)
                    rec.update(qkv=qkv, sbias=sbias, o_s=o_s, lse_s=lse_s, f=f, mean_s=mean_s, rstd_s=rstd_s, x2=x2)
                    # -- FFN: 3 launches
                    ffn = layer.ffn
                    F_ = ffn.linear1.out_features
                    h = torch.empty(B, Nq, F_, dtype=ops.act_dtype(cq), device=dev)
                    pre = torch.empty_like(h) if spec.act == "gelu" else None
                    L.gemm(M=R, N=F_, K=d, A=[x2], B=[ffn.linear1.weight.detach()], bias=[ffn.linear1.bias.detach()], Cs=[h],
                           C2=[pre], ct=cq, lda=d, ldb=d, ldc=F_, act=spec.act, drop=dr_fi)
                    # linear2 has K = F = 2048 on only M/64 x d/64 = 52 tiles: a long serial k-loop on a fifth of the chip.  Its K
                    # range is split over KS groups of ONE grouped launch (no atomics: each group owns an output), and the
                    # LayerNorm kernel adds the partial sums (+ residual, + dropout of the summed branch) in a fixed order --
                    # deterministic, so the bit-exact padding-invariance / scene-independence properties hold.
                    KS = 4 if F_ % (4 * 64) == 0 else 1
                    zp = torch.empty(KS, B, Nq, d, dtype=torch.float32, device=dev)
                    Fk = F_ // KS
                    hv, w2 = h.view(R, F_), ffn.linear2.weight.detach()
                    L.gemm(M=R, N=d, K=Fk, A=[hv[:, k * Fk:(k + 1) * Fk] for k in range(KS)],
                           B=[w2[:, k * Fk:(k + 1) * Fk] for k in range(KS)],
                           bias=[ffn.linear2.bias.detach()] + [None] * (KS - 1), Cs=[zp[k] for k in range(KS)], ct=cq, lda=F_,
                           ldb=F_, ldc=d)
                    z = torch.empty(B, Nq, d, dtype=torch.float32, device=dev)   # sum of the partials, kept for the backward
                    x3, mean_f, rstd_f = _ln_fwd(x2, [zp[k] for k in range(KS)], [ffn.norm.weight.detach()], [ffn.norm.bias.detach()],
                                                 ffn.norm.eps, None, Nq, drop=dr_fr, sum_branches=True, osum=z)
                rec.update(h=h, pre=pre, z=z, mean_f=mean_f, rstd_f=rstd_f)
                tape.append(rec)
                x = x3
        final_rec = None
        if spec.mh is not None:
            # detached aliases: x is also RETURNED, i.e. it becomes a tensor whose grad_fn is this ctx -- kept here as the same
            # object it would close a reference cycle (ctx -> rec -> x -> grad_fn = ctx) that only Python's cycle collector
            # frees: the saved activations of every step would outlive it, and under GraphedQuery3D the captured forward's
            # autograd graph did (round 4: its capture-stream AccumulateGrad nodes cost config 4's 'autograd' mode 2 ms per step)
            final_rec = {"x_in": x.detach()}
            cls, mlog, _ = _mh_forward(spec, x, keys, inv_den, seg_pad, final_rec, spec.num_blocks * Ln)
            if spec.skip_pred:
                pcls, pmask = [], []
            pcls.append(cls)
            pmask.append(mlog)
        ctx.spec, ctx.tape, ctx.final_rec = spec, tape, final_rec
        ctx.KV, ctx.keys, ctx.inv_den, ctx.kpm_all = KV, keys, inv_den, kpm_all
        ctx.PKV, ctx.pcas, ctx.prompt, ctx.pmask = PKV, pcas, prompt if spec.prompt else None, prompt_kpm if spec.prompt else None
        ctx.cas, ctx.n_mh = cas, len(pcls)
        ctx.params = params
        ctx.save_for_backward(x0, qpos, qmask, pos, pl, seg_pad, coef, *feats, *masks)
        ctx.M, ctx.U, ctx.src, ctx.mh_src, ctx.mh_feats = M, U, src, mh_src, mh_feats
        return (x, *pcls, *pmask)

    @staticmethod
    def backward(ctx, dxf, *dheads):
        return _DecoderBackward(ctx, dxf, dheads).run()


def fused_decoder(enc, input_dict, pairwise_locs, mask_head=None, seg_fts_for_match=None, seg_masks=None,
                  offline_attn_masks=None, skip_prediction=False):
    """Run QueryMaskEncoder (+ MaskHeadSegLevel) through the fused executor.  Returns
    (query, predictions_class, predictions_mask) exactly like QueryMaskEncoder.forward followed by the final
    mask-head call.  Raises NotImplementedError for configurations it does not cover (callers fall back to the
    modular path)."""
    layer0 = enc.unified_encoder[0]
    if layer0.structure not in ("parallel", "mixed"):
        raise NotImplementedError("fused path covers structure='parallel' and 'mixed'")
    if any(getattr(m_, "normalize_before", False) or getattr(m_, "spatial_attn_fusion", "mul") != "mul" for m_ in enc.modules()):
        raise NotImplementedError("fused path: post-norm layers with spatial_attn_fusion='mul' (the shipped configurations)")
    training = enc.training
    mems = [m for m in layer0.memories if training or m not in layer0.drop_memories_test]
    prompt = pmask = None
    if layer0.structure == "mixed":
        # query_encoder.py:162-165: parallel over the scene memories, then sequential_ca(query, ['prompt']) -- the literal
        # list, so the prompt attends even when drop_memories_test names it
        mems = [m for m in mems if m != "prompt"]
        if "prompt" not in layer0.memory2ca or "prompt" not in input_dict:
            raise NotImplementedError("fused path: structure='mixed' needs a prompt memory")
        prompt, pmask, ppos = input_dict["prompt"][:3]
        if ppos is not None or pmask.ndim != 2 or isinstance(prompt, (list, tuple)):
            raise NotImplementedError("fused path: prompt memory with a position tensor / 3-D mask / multi-scale list")
    if not mems or any(m == "prompt" for m in mems):
        raise NotImplementedError("fused path needs scene memories (a prompt memory only under structure='mixed')")
    x0, qmask, qpos = input_dict["query"][:3]
    feats = [input_dict[m][0] for m in mems]
    masks = [input_dict[m][1] for m in mems]
    poss = [input_dict[m][2] for m in mems]
    if any(m.ndim != 2 for m in masks):
        raise NotImplementedError("fused path: pre-set 3-D masks")
    Ln_ = len(enc.unified_encoder)
    # unique source tensors: a multi-scale voxel memory (list: one [B, N_seg, d] tensor per layer, the last one for the
    # mask head -- pcd_mask3d_encoder.py:133-154, query_encoder.py:90-91) contributes one source per scale
    uniq, src = [], [[0] * len(mems) for _ in range(Ln_)]

    def uid(t):
        for k, q in enumerate(uniq):
            if q is t:
                return k
        uniq.append(t)
        return len(uniq) - 1
    for j, f in enumerate(feats):
        if isinstance(f, (list, tuple)):
            if len(f) < Ln_:
                raise NotImplementedError("fused path: a multi-scale memory needs one scale per layer")
            for i in range(Ln_):
                src[i][j] = uid(f[i])
        else:
            for i in range(Ln_):
                src[i][j] = uid(f)
    if any(p is not poss[0] for p in poss) or len({tuple(f.shape) for f in uniq}) != 1:
        raise NotImplementedError("fused path: memories must share one position tensor and one shape")
    if len(mems) * Ln_ * 2 > 4 * MAXG:
        raise NotImplementedError("fused path: too many (layer, memory) groups")
    mh_count, mh_src = 0, []
    if mask_head is not None:
        if seg_fts_for_match is None:
            raise NotImplementedError("fused path: mask head without seg_fts_for_match")
        mh_count = min(len(seg_fts_for_match), len(mask_head.mask_pred_list))   # mask_head.py:31: zip() truncates
        for k, sf in enumerate(seg_fts_for_match[:mh_count]):
            want = feats[k][-1] if (k < len(feats) and isinstance(feats[k], (list, tuple))) else (feats[k] if k < len(feats) else None)
            if sf[0] is not want:
                raise NotImplementedError("fused path: mask-head memories must be the leading scene memories")
            mh_src.append(uid(sf[0]))
    if enc.use_self_mask and mask_head is None:
        raise NotImplementedError("use_self_mask without a mask head")
    coef = None
    if training and layer0.memory_dropout > 0.0:
        # query_encoder.py:145-151: every layer application draws its own per-(scene, memory) keep mask -> [apps, M, B]
        from .modules import memory_keep_coef
        hook = getattr(enc, "memory_keep_hook", None)
        n_app = enc.num_blocks * len(enc.unified_encoder)
        B_ = x0.shape[0]
        coef = torch.stack([memory_keep_coef(B_, len(mems), layer0.memory_dropout, x0.device,
                                             hook(a, B_, len(mems), x0.device) if hook is not None else None)
                            for a in range(n_app)]).contiguous()
    ct = L.BF16 if layer0.compute in ("bf16", "bf16x3") else L.F32
    # compute mode 'bf16x3': the split-bf16 key/value side where its kernels cover the shape (128-row-tile plane GEMM: d % 128 == 0;
    # csrc/attn_x3.hip: d_h = 32 / 64, <= 256 queries), the exact-f32 kernels otherwise -- same accuracy contract
    kv3 = False
    if layer0.compute == "bf16x3":
        B_, Ns_, d_ = uniq[0].shape
        kv3 = (d_ % 128 == 0 and d_ in (32 * enc.num_heads, 64 * enc.num_heads) and x0.shape[1] <= 256 and B_ * Ns_ >= 128 and 2 * len(uniq) <= MAXG and
               len(mems) * Ln_ <= MAXG and all(f.dtype == torch.float32 for f in uniq) and
               (prompt is None or (prompt.dtype == torch.float32 and (prompt.shape[0] * prompt.shape[1] * d_) % 8 == 0 and
                                   (x0.shape[0] * x0.shape[1] * d_) % 8 == 0)))
        if not kv3:
            ct = L.F32
    drop_base, mh_drop = None, False
    if training:   # the caller (QueryMaskEncoder.forward) has opened the RNG epoch (modules.begin_dropout_step)
        layers_ = list(enc.unified_encoder)
        ps = [m.dropout_p for l_ in layers_ for m in (list(l_.cross_attn_list) + [l_.self_attn, l_.ffn])]
        if any(p_ > 0.0 for p_ in ps):
            if any(len({c.dropout_p for c in l_.cross_attn_list}) > 1 for l_ in layers_):
                raise NotImplementedError("fused path: one dropout probability per layer's cross-attention list")
            drop_base = enc._drop_base
        mh_drop = mask_head is not None and mask_head.dropout_p > 0.0
    spec = FusedSpec(enc, mask_head, mems, ct, layer0.ffn.activation, enc.use_self_mask, enc.num_blocks,
                     enc.spatial_selfattn, mh_count, offline_attn_masks is not None, skip_prediction, drop_base, mh_drop)
    st = input_dict.get("_stacked_scene_kpm")
    if st is not None and list(st[1]) == list(mems) and st[0].is_contiguous() and \
            all(masks[j].data_ptr() == st[0][j].data_ptr() for j in range(len(mems))):
        spec.stacked_kpm = st[0]
    spec.src, spec.mh_src, spec.n_src = src, mh_src, len(uniq)
    spec.kv3 = kv3
    spec.prompt = prompt is not None
    params = [p for p in enc.parameters()] + ([p for p in mask_head.parameters()] if mask_head is not None else [])
    outs = _FusedDecoder.apply(spec, x0, qpos, qmask, poss[0], pairwise_locs, seg_masks, offline_attn_masks, coef,
                               prompt, pmask, *uniq, *masks, *params)
    query = outs[0]
    n = (len(outs) - 1) // 2
    return query, list(outs[1:1 + n]), list(outs[1 + n:1 + 2 * n])
