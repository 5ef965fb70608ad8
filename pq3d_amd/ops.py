"""torch.autograd.Function wrappers around the C-ABI kernels.  PyTorch supplies device memory, streams and the
autograd tape only; every FLOP below runs in libpq3d_hip.so.  Shapes follow the reference's batch-first
[B, L, d] convention; masks are torch.bool with the PyTorch meaning True = ignore."""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional, Sequence

import torch
from torch.autograd import Function

from . import _lib as L
from ._lib import BF16, BF16X3, F32
from .profiler import timed

_empty = torch.empty
# PQ3D_DETERMINISTIC=1: reductions that have an order-independent form take it (today: bias-gradient column sums of long
# accumulating calls, include/pq3d_hip.h pq3d_colsum_grouped accumulate == 2).  Split-K weight gradients and the LayerNorm
# parameter gradients still add with fp32 atomics.
DETERMINISTIC = os.environ.get("PQ3D_DETERMINISTIC", "0") == "1"


def act_dtype(ct: int) -> torch.dtype:
    """Storage dtype of MFMA-operand activations (Q/K/V/O, FFN hidden) for a compute type.  Split-bf16 (BF16X3)
    products take fp32 operands (they are split into hi + lo bf16 parts inside the GEMM staging path)."""
    return torch.bfloat16 if ct == BF16 else torch.float32


def bwd_ct(ct: int) -> int:
    """Compute type of the BACKWARD products of an op whose forward ran at ``ct``: split-bf16 is a forward-accuracy
    device (it keeps ReLU kinks / mask thresholds / softmax inputs at fp32 grade); gradients are formed with single
    bf16 products (fp32 accumulate), whose ~1e-3 relative error is far inside the 2e-2 gradient tolerance."""
    return BF16 if ct == BF16X3 else ct


def small_ct(ct: int) -> int:
    """Compute type of the query-side (M = B*N_q rows) projections / FFN / head GEMMs for a module compute type:
    'bf16' -> split-bf16 (these launches are latency-bound, the extra MFMAs are free), 'fp32' -> exact f32."""
    return BF16X3 if ct == BF16 else ct


def sa_ct(ct: int) -> int:
    """Compute type of the decoder's self-attention core (fp32 q / k / v in both modes): 'bf16' mode -> fp32-grade split-bf16
    MFMA kernels (csrc/attn_sa.hip), 'fp32' mode -> exact fp32."""
    return BF16X3 if ct == BF16 else F32


def _c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return None if t is None else t.contiguous()


def _splitk(tiles: int, k: int, ct: int) -> int:
    nkt = max(1, k // (64 if ct == BF16 else 32))
    return max(1, min(nkt // 2 if nkt >= 2 else 1, 512 // max(tiles, 1), 64))


# ------------------------------------------------------------------------------------------------ dropout
class DropRNG:
    """Per-device dropout state (include/pq3d_hip.h "Dropout").  ``seed`` is the int64 device word the kernels hash;
    advance() bumps it ON THE DEVICE (capturable: a replayed HIP graph draws new masks every step) and snapshots it
    into ``cur`` -- the tensor dropout sites actually point at, so a backward pass that runs after a later forward
    still regenerates its own masks.  The initial value comes from torch's CPU generator (torch.manual_seed)."""

    def __init__(self, device):
        self.seed = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).to(device)
        self.cur = self.seed.clone()
        self.epoch = 0

    def advance(self) -> None:
        self.seed.add_(1)
        self.cur = self.seed.clone()
        self.epoch += 1

    def set_seed(self, value: int) -> None:
        self.seed.fill_(int(value))
        self.cur = self.seed.clone()
        self.epoch += 1


_DROP_RNG = {}


def drop_rng(device) -> DropRNG:
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    st = _DROP_RNG.get(device)
    if st is None:
        st = _DROP_RNG[device] = DropRNG(device)
    return st


# dropout-site numbering inside one decoder forward: (layer application, kind, memory) -> site id.  The fused and
# the modular executors (and the oracle-side mask generator in tests) all use this one function.
DROP_CA_ATTN, DROP_CA_RES, DROP_SA_ATTN, DROP_SA_RES, DROP_FFN_INNER, DROP_FFN_RES, DROP_MLP_HEAD, DROP_ENC_OUT = range(8)


def drop_site(base: int, app: int, kind: int, m: int = 0) -> int:
    return base + (app * 16 + kind) * 8 + m


def make_drop(p: float, site: int, device) -> Optional[L.Drop]:
    return L.Drop(p, site, drop_rng(device).cur) if p and p > 0.0 else None


def dropout_mask(rows: int, cols: int, drop: L.Drop) -> torch.Tensor:
    """keep-mask [rows, cols] (bool) of a dropout site -- what the fused kernels draw (tests, debugging)."""
    keep = _empty(rows, cols, dtype=torch.bool, device=drop.seed.device)
    dc = drop.c()
    L.check(L.lib().pq3d_dropout_mask(L.ptr(keep), rows, cols, C.byref(dc), L.stream()), "pq3d_dropout_mask")
    return keep


def _dropout_apply(x: torch.Tensor, drop: L.Drop, out_dtype=None, alpha: float = 1.0) -> torch.Tensor:
    x = x.contiguous()
    y = _empty(x.shape, dtype=out_dtype or x.dtype, device=x.device)
    cols = x.shape[-1]
    dc = drop.c()
    L.check(L.lib().pq3d_dropout_apply_scaled(L.ptr(x), L.dt_of(x), L.ptr(y), L.dt_of(y), x.numel() // cols, cols, C.byref(dc),
                                              float(alpha), L.stream()), "pq3d_dropout_apply")
    return y


class _Dropout(Function):
    @staticmethod
    def forward(ctx, x, drop, alpha=1.0):
        ctx.drop, ctx.alpha = drop, alpha
        return _dropout_apply(x, drop, alpha=alpha)

    @staticmethod
    def backward(ctx, dy):
        return _dropout_apply(dy, ctx.drop, alpha=ctx.alpha), None, None


def dropout(x: torch.Tensor, drop: Optional[L.Drop], alpha: float = 1.0) -> torch.Tensor:
    """alpha * nn.Dropout(x) over the last dim as the site's column axis (site = x viewed as [rows, x.shape[-1]])."""
    if drop is None:
        return x if alpha == 1.0 else x * alpha
    return _Dropout.apply(x, drop, float(alpha))


# ------------------------------------------------------------------------------------------------ small kernels
def split_planes(srcs, adds, his, los) -> None:
    """bf16 hi / lo planes of fp32 tensors (+ an optional addend): his[g] = bf16(v), los[g] = bf16(v - his[g]); his[g] may be None.
    The operand form of the split-bf16 key/value side (compute mode 'bf16x3')."""
    for i in range(0, len(srcs), L.MAXG):
        sl = slice(i, i + L.MAXG)
        n = len(srcs[sl])
        arr = lambda ts: (C.c_void_p * n)(*[L.ptr(t) for t in ts])
        cnt = (C.c_int64 * n)(*[t.numel() for t in srcs[sl]])
        L.check(L.lib().pq3d_split_planes(arr(srcs[sl]), arr(adds[sl]), arr(his[sl]), arr(los[sl]), cnt, n, L.stream()),
                "pq3d_split_planes")


def cast_bf16(ts):
    """bf16 copies of same-sized fp32 tensors in one launch (pq3d_add_cast): the bf16 tape of a split-bf16 forward."""
    outs = [_empty(t.shape, dtype=torch.bfloat16, device=t.device) for t in ts]
    n = ts[0].numel()
    assert all(t.numel() == n and t.dtype == torch.float32 and t.is_contiguous() for t in ts) and n % 8 == 0
    for i in range(0, len(ts), L.MAXG):
        k = len(ts[i:i + L.MAXG])
        arr = lambda xs: (C.c_void_p * k)(*[L.ptr(t) for t in xs])
        L.check(L.lib().pq3d_add_cast(arr(ts[i:i + k]), arr([None] * k), arr(outs[i:i + k]), k, BF16, n, L.stream()), "pq3d_add_cast")
    return outs


def colsum(x2d: torch.Tensor) -> torch.Tensor:
    R, N = x2d.shape
    out = _empty(N, dtype=torch.float32, device=x2d.device)
    L.check(L.lib().pq3d_colsum(L.ptr(x2d), L.dt_of(x2d), R, N, N, L.ptr(out), L.stream()), "pq3d_colsum")
    return out


def scale_rows(x: torch.Tensor, rows: int, out_dtype: torch.dtype, scale=None, zero_flag=None, keep_mask=None):
    y = _empty(x.shape, dtype=out_dtype, device=x.device)
    L.check(L.lib().pq3d_scale_rows(L.ptr(x), L.dt_of(x), L.ptr(y), L.dt_of(y), rows, x.numel() // max(rows, 1),
                                    L.ptr(scale), L.ptr(zero_flag), L.ptr(keep_mask), L.stream()), "pq3d_scale_rows")
    return y


def scale_rows_many(xs, rows: int, out_dtype: torch.dtype, scale=None, zero_flag=None, keep_mask=None) -> list:
    """[scale_rows(x, ...) for x in xs] for same-shape fp32 tensors sharing the scales / flags: one launch per 32 tensors."""
    xs = [x.contiguous() for x in xs]
    C_ = xs[0].numel() // max(rows, 1)
    if len(xs) < 2 or C_ % 8 or any(x.dtype != torch.float32 or x.shape != xs[0].shape or x.data_ptr() % 16 for x in xs):
        return [scale_rows(x, rows, out_dtype, scale, zero_flag, keep_mask) for x in xs]
    ys = _empty((len(xs),) + tuple(xs[0].shape), dtype=out_dtype, device=xs[0].device)
    for s0 in range(0, len(xs), L.MAXG):
        ch = range(s0, min(s0 + L.MAXG, len(xs)))
        xa = (C.c_void_p * len(ch))(*[L.ptr(xs[i]) for i in ch])
        ya = (C.c_void_p * len(ch))(*[L.ptr(ys[i]) for i in ch])
        L.check(L.lib().pq3d_scale_rows_grouped(xa, ya, len(ch), L.dt_of(ys), rows, C_, L.ptr(scale), L.ptr(zero_flag), L.ptr(keep_mask),
                                                L.stream()), "pq3d_scale_rows_grouped")
    return [ys[i] for i in range(len(xs))]


def act_bwd(dy: torch.Tensor, saved: torch.Tensor, act: str, out_dtype: torch.dtype) -> torch.Tensor:
    out = _empty(dy.shape, dtype=out_dtype, device=dy.device)
    L.check(L.lib().pq3d_act_bwd(L.ptr(dy), L.dt_of(dy), L.ptr(saved), L.dt_of(saved), L.ptr(out), L.dt_of(out),
                                 L.ACT[act], dy.numel(), L.stream()), "pq3d_act_bwd")
    return out


def mask_row_all(mask: torch.Tensor) -> torch.Tensor:
    """[B,Lq,Lk] bool -> [B,Lq] bool, True where the whole row is masked (query_encoder.py:83)."""
    mask = mask.contiguous()
    out = _empty(mask.shape[:-1], dtype=torch.bool, device=mask.device)
    L.check(L.lib().pq3d_mask_row_all(L.ptr(mask), L.ptr(out), out.numel(), mask.shape[-1], L.stream()),
            "pq3d_mask_row_all")
    return out


def mask_pack(mask: torch.Tensor):
    """[B,Lq,Lk] bool -> (row_open [B,Lq] bool as mask_row_all, bits [B,Lq,ceil(Lk/32)] int32: the mask as bit words with the
    open rows cleared) in one pass (pq3d_mask_pack); the resident attention backward reads the bits instead of the bytes."""
    mask = mask.contiguous()
    Lk = mask.shape[-1]
    out = _empty(mask.shape[:-1], dtype=torch.bool, device=mask.device)
    bits = _empty(*mask.shape[:-1], (Lk + 31) // 32, dtype=torch.int32, device=mask.device)
    L.check(L.lib().pq3d_mask_pack(L.ptr(mask), L.ptr(out), L.ptr(bits), out.numel(), Lk, L.stream()), "pq3d_mask_pack")
    return out, bits


def mask_inv_den(masks: Sequence[torch.Tensor]) -> torch.Tensor:
    masks = [m.contiguous() for m in masks]
    arr = (C.c_void_p * len(masks))(*[L.ptr(m) for m in masks])
    out = _empty(masks[0].shape, dtype=torch.float32, device=masks[0].device)
    L.check(L.lib().pq3d_mask_inv_den(arr, len(masks), out.numel(), L.ptr(out), L.stream()), "pq3d_mask_inv_den")
    return out


def pairwise_locs(centers: torch.Tensor, eps: float = 1e-10) -> torch.Tensor:
    """calc_pairwise_locs (modules/utils.py:38-87, 'center', spatial_dim=5).  centers [B,L,>=3] fp32."""
    assert centers.dtype == torch.float32 and centers.stride(-1) == 1
    B, Lq = centers.shape[:2]
    if centers.stride(0) != Lq * centers.stride(1):
        centers = centers.contiguous()
    out = _empty(B, Lq, Lq, 5, dtype=torch.float32, device=centers.device)
    L.check(L.lib().pq3d_pairwise_locs(L.ptr(centers), centers.stride(1), L.ptr(out), B, Lq, eps, L.stream()),
            "pq3d_pairwise_locs")
    return out


def fourier(xyz: torch.Tensor, cmin: torch.Tensor, cmax: torch.Tensor, gauss_B: torch.Tensor, out=None) -> torch.Tensor:
    """Fourier features [sin | cos] of normalised coordinates (position_embedding.py:127-156); no grad."""
    assert xyz.dtype == torch.float32 and xyz.stride(-1) == 1
    B, N = xyz.shape[:2]
    if xyz.stride(0) != N * xyz.stride(1):
        xyz = xyz.contiguous()
    half = gauss_B.shape[1]
    if out is None:
        out = _empty(B, N, 2 * half, dtype=torch.float32, device=xyz.device)
    L.check(L.lib().pq3d_fourier(L.ptr(xyz), xyz.stride(1), L.ptr(_c(cmin.float())), L.ptr(_c(cmax.float())),
                                 L.ptr(_c(gauss_B)), L.ptr(out), B, N, half, L.stream()), "pq3d_fourier")
    return out


def fourier_pair(xyz_a: torch.Tensor, xyz_b: torch.Tensor, cmin, cmax, gauss_B) -> torch.Tensor:
    """fourier() of two point sets of the same scenes in one launch: [B*Na + B*Nb, 2*half], set a's rows first."""
    def prep(x):
        assert x.dtype == torch.float32 and x.stride(-1) == 1
        return x if x.stride(0) == x.shape[1] * x.stride(1) else x.contiguous()
    xyz_a, xyz_b = prep(xyz_a), prep(xyz_b)
    B, Na, Nb, half = xyz_a.shape[0], xyz_a.shape[1], xyz_b.shape[1], gauss_B.shape[1]
    out = _empty(B * (Na + Nb), 2 * half, dtype=torch.float32, device=xyz_a.device)
    L.check(L.lib().pq3d_fourier_pair(L.ptr(xyz_a), xyz_a.stride(1), Na, L.ptr(xyz_b), xyz_b.stride(1), Nb,
                                      L.ptr(_c(cmin.float())), L.ptr(_c(cmax.float())), L.ptr(_c(gauss_B)), L.ptr(out), B, half,
                                      L.stream()), "pq3d_fourier_pair")
    return out


def _parr(ts):
    return (C.c_void_p * len(ts))(*[L.ptr(t) for t in ts])


def mask_not(masks: Sequence[torch.Tensor]) -> list:
    """[~m for m in masks] in ONE launch (bool masks of any shapes): the 'True = valid' -> 'True = ignore' inversions
    of Query3DUnified.forward (query3d_unified.py:113,139,143,148,155).  Equal-shape masks come back as views of one
    stacked buffer (what the fused executor takes as the memories-stacked key-padding mask)."""
    masks = [m.contiguous() for m in masks]
    total = sum(m.numel() for m in masks)
    buf = _empty(total, dtype=torch.bool, device=masks[0].device)
    outs, off = [], 0
    for m in masks:
        outs.append(buf[off:off + m.numel()].view(m.shape))
        off += m.numel()
    cnt = (C.c_int64 * len(masks))(*[m.numel() for m in masks])
    for s0 in range(0, len(masks), L.MAXG):
        e = min(len(masks), s0 + L.MAXG)
        L.check(L.lib().pq3d_mask_not(_parr(masks[s0:e]), _parr(outs[s0:e]), C.byref(cnt, 8 * s0), e - s0, L.stream()),
                "pq3d_mask_not")
    return outs


def zero_many(tensors: Sequence[torch.Tensor]) -> None:
    """Zero-fill several fp32 tensors with one launch (hipMemsetAsync nodes are not replay-safe on this stack: common.h)."""
    ts = [t for t in tensors if t is not None and t.numel()]
    if not ts:
        return
    assert all(t.dtype == torch.float32 and t.is_contiguous() for t in ts)
    cnt = (C.c_int64 * len(ts))(*[t.numel() for t in ts])
    L.check(L.lib().pq3d_zero_many(_parr(ts), cnt, len(ts), L.stream()), "pq3d_zero_many")


def copy_many(dsts: Sequence[torch.Tensor], srcs: Sequence[torch.Tensor]) -> None:
    """dst_i.copy_(src_i) for many small fp32 tensors in one launch (the gradient pack of parallel.FlatGradAllReducer)."""
    pairs = [(d_, s_.contiguous()) for d_, s_ in zip(dsts, srcs) if d_.numel()]
    if not pairs:
        return
    assert all(d_.dtype == torch.float32 and s_.dtype == torch.float32 and d_.is_contiguous() and d_.numel() == s_.numel()
               for d_, s_ in pairs)
    cnt = (C.c_int64 * len(pairs))(*[d_.numel() for d_, _ in pairs])
    L.check(L.lib().pq3d_copy_many(_parr([s_ for _, s_ in pairs]), _parr([d_ for d_, _ in pairs]), cnt, len(pairs), L.stream()),
            "pq3d_copy_many")


def sum_n(parts: Sequence[torch.Tensor]) -> torch.Tensor:
    """sum of same-shape fp32 tensors in a fixed order, one launch (replaces torch.stack(..).sum(0): cat + reduce)."""
    parts = [p.contiguous() for p in parts]
    if len(parts) == 1:
        return parts[0]
    out = None
    step = L.MAXG - 1
    for s0 in range(0, len(parts), step):
        chunk = ([out] if out is not None else []) + parts[s0:s0 + step]
        new = torch.empty_like(parts[0])
        L.check(L.lib().pq3d_sum_n(_parr(chunk), len(chunk), L.ptr(new), new.numel(), L.stream()), "pq3d_sum_n")
        out = new
    return out


def sum_pair(parts_a: Sequence[torch.Tensor], parts_b: Sequence[torch.Tensor]):
    """(sum_n(parts_a), sum_n(parts_b)) in one launch, the two results ADJACENT in one buffer (a first): a consumer that
    wants their row concatenation (_SplitRows.backward) finds it already formed."""
    parts_a, parts_b = [p.contiguous() for p in parts_a], [p.contiguous() for p in parts_b]
    if len(parts_a) > L.MAXG or len(parts_b) > L.MAXG or parts_a[0].numel() % 4 or parts_b[0].numel() % 4:
        return sum_n(parts_a), sum_n(parts_b)
    na, nb = parts_a[0].numel(), parts_b[0].numel()
    buf = torch.empty(na + nb, dtype=torch.float32, device=parts_a[0].device)
    oa, ob = buf[:na].view(parts_a[0].shape), buf[na:].view(parts_b[0].shape)
    L.check(L.lib().pq3d_sum_pair(_parr(parts_a), len(parts_a), L.ptr(oa), na, _parr(parts_b), len(parts_b), L.ptr(ob), nb,
                                  L.stream()), "pq3d_sum_pair")
    return oa, ob


_MEAN_WS = {}


class _MeanAll(Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x.float())
        out = _empty(1, dtype=torch.float32, device=x.device)
        ws = _MEAN_WS.get(x.device)
        if ws is None:     # arrival counter + per-block partials; the kernel leaves the counter at zero again
            ws = _MEAN_WS[x.device] = torch.zeros(1 + 256, dtype=torch.float32, device=x.device)
        L.check(L.lib().pq3d_mean_all(L.ptr(x), x.numel(), L.ptr(out), L.ptr(ws), L.stream()), "pq3d_mean_all")
        ctx.shape = x.shape
        return out.view(())

    @staticmethod
    def backward(ctx, g):
        n = 1
        for s_ in ctx.shape:
            n *= s_
        dx = _empty(ctx.shape, dtype=torch.float32, device=g.device)
        gg = _c(g.float().reshape(1))
        L.check(L.lib().pq3d_fill_scaled(L.ptr(dx), n, L.ptr(gg), 1.0 / n, L.stream()), "pq3d_fill_scaled")
        return dx


class _MeanMany(Function):
    @staticmethod
    def forward(ctx, modes, cmins, *xs):
        xs = [_c(x.float()) for x in xs]
        dev = xs[0].device
        n = len(xs)
        out = _empty(1, dtype=torch.float32, device=dev)
        key = (dev, "many")
        ws = _MEAN_WS.get(key)
        if ws is None:
            ws = _MEAN_WS[key] = torch.zeros(1 + 2 * L.MAXG + 128 * L.MAXG, dtype=torch.float32, device=dev)
        ctx.arr = ((C.c_int64 * n)(*[x.numel() for x in xs]), (C.c_int32 * n)(*modes), (C.c_float * n)(*cmins))
        L.check(L.lib().pq3d_mean_many(_parr(xs), ctx.arr[0], ctx.arr[1], ctx.arr[2], n, L.ptr(out), L.ptr(ws), L.stream()),
                "pq3d_mean_many")
        ctx.save_for_backward(*xs)
        return out.view(())

    @staticmethod
    def backward(ctx, g):
        xs = ctx.saved_tensors
        dxs = [torch.empty_like(x) for x in xs]
        gg = _c(g.float().reshape(1))
        L.check(L.lib().pq3d_mean_many_bwd(_parr(xs), _parr(dxs), ctx.arr[0], ctx.arr[1], ctx.arr[2], len(xs), L.ptr(gg),
                                           L.stream()), "pq3d_mean_many_bwd")
        return (None, None, *dxs)


def mean_many(xs: Sequence[torch.Tensor], modes: Sequence[str], clamp_min: float = 0.0) -> torch.Tensor:
    """sum_g mean(f_g(x_g)) as ONE launch forward and one backward; modes[g] in {'plain', 'clamp_min', 'finite'}
    ('clamp_min': x.clamp(min=clamp_min).mean(); 'finite': torch.where(isfinite(x), x, 0).mean()) -- the synthetic loss over
    the prediction layers without ~14 framework launches per layer."""
    code = {"plain": 0, "clamp_min": 1, "finite": 2}
    total = None
    for s0 in range(0, len(xs), L.MAXG):
        part = _MeanMany.apply([code[m] for m in modes[s0:s0 + L.MAXG]], [float(clamp_min)] * len(xs[s0:s0 + L.MAXG]),
                               *xs[s0:s0 + L.MAXG])
        total = part if total is None else total + part
    return total


def mean_all(x: torch.Tensor) -> torch.Tensor:
    """x.mean() as one deterministic launch forward, one fill backward (the synthetic 'mean(query)' loss)."""
    return _MeanAll.apply(x)


# ------------------------------------------------------------------------------------------------ linear
# ---- deferred weight gradients of a whole-pass gradient arena ------------------------------------------------------------
# Inside ``with ops.grad_arena(...)`` every dW = g^T (x [+ x2]) of an arena-aware linear layer only feeds its slot of the flat
# gradient buffer, so nothing waits for it: the products are queued per (shape, dtype) bucket and launched as a few GROUPED
# split-K launches when the pass ends (grad_arena.__exit__) instead of one launch per layer -- the caption body's 46 launches of
# [512 x 512] / [2048 x 512] products over 512 rows at config 5 are latency chains of ~15 us each.  The decoder's own backward
# has done the same for its layers since round 1 (fused._DwQueue); this is the same idea for everything that goes through
# ops.linear / ops.linear_group around it.
class _DwDeferred:
    buckets = {}     # (N, K, R, g dtype, x dtype, has x2, has bias, ct) -> lists
    nbytes = 0       # operand bytes the queue keeps alive; above _DW_DEFER_CAP the queue is flushed on the spot


_DW_DEFER_CAP = 256 << 20   # the deferral exists for the latency-bound small-R products; big operands are not held for long


_DW_DEFER = True    # module switch for A/B measurements (tools/probes/bench_nodefer.py)


def _dw_can_defer(g, x, x2, N, K) -> bool:
    if not _DW_DEFER or not _Arena.whole_pass or _Arena.mode is None:
        return False
    return not ((x2 is not None and (x2.dtype != torch.float32 or x.dtype != torch.float32)) or N % 8 or K % 8 or
                g.data_ptr() % 16 or x.data_ptr() % 16 or (x2 is not None and x2.data_ptr() % 16) or
                not g.is_contiguous() or not x.is_contiguous())


_DW_NO_DEFER = set()    # id(param): never deferred (dw_defer_exclude)


def dw_defer_exclude(params, on: bool = True) -> None:
    """Exclude parameters from weight-gradient deferral.  REQUIRED for parameters whose gradient is read by a hook Python cannot see:
    torch DDP's reducer registers its bucket hooks in C++ on the AccumulateGrad node (grad_accumulator->add_post_hook), which neither
    Tensor._backward_hooks nor _post_accumulate_grad_hooks shows -- under a whole-pass gradient arena such a hook would read a slot
    whose queued product only runs when the pass ends.  (The package's own data-parallel path, parallel.FlatGradAllReducer, flushes
    the queue before a bucket leaves and needs no exclusion.)"""
    for p in params:
        (_DW_NO_DEFER.add if on else _DW_NO_DEFER.discard)(id(p))


def _param_has_hooks(q) -> bool:
    """Python-visible gradient hooks (tensor hooks, post-accumulate-grad hooks, Python hooks on the AccumulateGrad node) or an
    explicit exclusion (dw_defer_exclude: C++-side hooks such as DDP's cannot be detected from here)."""
    ent = _Arena.by_ptr.get(q)
    if ent is None:
        return False
    p_ = ent[0]
    if id(p_) in _DW_NO_DEFER or getattr(p_, "_backward_hooks", None) or getattr(p_, "_post_accumulate_grad_hooks", None):
        return True
    try:   # Python hooks registered on the parameter's AccumulateGrad node itself
        acc = p_.view_as(p_).grad_fn.next_functions[0][0]
        return bool(getattr(acc, "_post_hooks", None)) or bool(getattr(acc, "_pre_hooks", None))
    except Exception:  # noqa: BLE001
        return False


def _dw_defer(g, x, x2, dw, db, N, K, R, ct, pptrs=()) -> bool:
    """Queue dw[N, K] += g^T (x [+ x2]) (and db[N] += column sums of g); False when the pass has no whole-pass arena.
    A parameter with Python-visible gradient hooks, or one listed through dw_defer_exclude (torch DDP: its C++ bucket hooks are
    invisible from Python), is never deferred: its hook would read the slot before the queued product has run."""
    if not _dw_can_defer(g, x, x2, N, K) or any(_param_has_hooks(q) for q in pptrs if q is not None):
        return False
    _DwDeferred.nbytes += g.numel() * g.element_size() + x.numel() * x.element_size() + \
        (x2.numel() * x2.element_size() if x2 is not None else 0)
    key = (N, K, R, g.dtype, x.dtype, x2 is not None, db is not None, ct)
    b = _DwDeferred.buckets.setdefault(key, ([], [], [], [], []))
    # the queue keeps its OWN view objects of the slots: AccumulateGrad adopts the returned gradient without a copy only when
    # nobody else references that tensor object (arena_take) -- a second reference would turn .grad into a clone of zeros
    b[0].append(g); b[1].append(x); b[2].append(x2); b[3].append(dw.view(N, K)); b[4].append(db.view(-1) if db is not None else None)
    if _DwDeferred.nbytes > _DW_DEFER_CAP:
        dw_deferred_flush()
    return True


_TT_MULTI = os.environ.get("PQ3D_TT_MULTI", "1") != "0"   # A/B switch (tools/probes): one launch for a whole flush
# ... also for long reductions over few tiles (the encoders' [256 x 256] over 8192 rows)?  Measured slower (k-slices x 256 x 128
# tiles = 3x the atomics of the 64 x 64 split: c4 +1.0 %, c2 +0.2 .. 1.7 % on the same box): off
_TT_LONG = os.environ.get("PQ3D_TT_MULTI_LONG", "0") != "0"


def dw_long_path(N: int, K: int, R: int, count: int, ct: int) -> bool:
    """dw_operands' condition: a LONG reduction with enough 128 x 128 output tiles to fill the chip -> operands rounded to bf16
    once, then the 128 x 128-tile bf16 kernel (gemm_tt128).  Everything else is pq3d_gemm_tt_multi's."""
    return ct == BF16 and R >= 2048 and R % 64 == 0 and N % 128 == 0 and K % 128 == 0 and (N // 128) * (K // 128) * count >= 64


def tt_multi_ok(g, x, x2, dw, db, N: int, K: int, R: int) -> bool:
    """Can dW[N, K] += g^T (x [+ x2]) (+ db[N] += colsum g) join the one-launch flush (pq3d_gemm_tt_multi)?  (Callers keep long
    reductions over many tiles -- dw_long_path -- on the 128 x 128-tile bf16 kernel.)"""
    if not _TT_MULTI or (R >= 2048 and not _TT_LONG) or R < 1 or N % 8 or K % 8 or N < 8 or K < 8 or R * max(N, K) >= (1 << 31):
        return False
    for t in (g, x):
        if t.dtype not in (torch.float32, torch.bfloat16) or not t.is_contiguous() or t.data_ptr() % 16:
            return False
    if x2 is not None and (x2.dtype != torch.float32 or x.dtype != torch.float32 or not x2.is_contiguous() or x2.data_ptr() % 16):
        return False
    if dw.dtype != torch.float32 or not dw.is_contiguous() or (db is not None and (db.dtype != torch.float32 or not db.is_contiguous())):
        return False
    return g.numel() == R * N and x.numel() == R * K


def tt_multi_pays(problems) -> bool:
    """The one-launch flush wins through its 256 x 128 tiles (2.5-3x fewer operand re-reads from L2) while those fit about one
    round of workgroups (<= 400 wide tiles: the decoder's flush at configs 2 / 4 / 5); a flush WITHOUT such a launch -- the
    caption body's ~660 wide tiles over 512 rows, which the library sends back to 64 x 64 tiles -- is faster as one
    gemm_wktt launch per (shape, dtype) bucket (config 5, same box: 6.21 vs 6.27 ms per step)."""
    wt = sum((dw.shape[-2] // 256) * (dw.shape[-1] // 128) for _g, _x, _x2, dw, _db in problems
             if dw.shape[-2] % 256 == 0 and dw.shape[-1] % 128 == 0)
    return 0 < wt <= 400


def tt_multi(problems) -> None:
    """problems: [(g [R,N], x [R,K], x2 or None, dw [N,K], db [N] or None)] -- every weight (and bias) gradient of a flush in
    ONE launch per 56 problems (csrc/gemm_ttmulti.hip) instead of one launch per (shape, dtype) bucket."""
    for s0 in range(0, len(problems), L.TT_MAX_PROBLEMS):
        ch = problems[s0:s0 + L.TT_MAX_PROBLEMS]
        arr = (L.TtProblem * len(ch))()
        fl = nb = 0.0
        for q, (g, x, x2, dw, db) in zip(arr, ch):
            N, K = dw.shape[-2], dw.shape[-1]
            R = g.numel() // N
            q.M, q.N, q.K, q.lda, q.ldb = N, K, R, N, K
            q.dtA, q.dtB = L.dt_of(g), L.dt_of(x)
            q.A, q.B, q.B2, q.C, q.colsum = L.ptr(g), L.ptr(x), L.ptr(x2), L.ptr(dw), L.ptr(db)
            fl += 2.0 * N * K * R
            nb += float(g.numel() * g.element_size() + x.numel() * x.element_size() + (x2.numel() * x2.element_size() if x2 is not None else 0)
                        + N * K * 4 + (N * 4 if db is not None else 0))   # compulsory: both operands once, the fp32 result (+ bias gradient)
        L.check(timed("pq3d_gemm_tt_multi", f"ttmulti{len(ch)}", fl, nb, L.lib().pq3d_gemm_tt_multi, arr, len(ch), L.stream()),
                "pq3d_gemm_tt_multi")


def dw_deferred_flush(run: bool = True) -> None:
    """Launch (run=False: drop) the queued weight-gradient products.  Called when the pass ends (grad_arena.__exit__), from
    the fused decoder's readiness reports, from FlatGradAllReducer.launch() / pack() -- no reader of a slot gets ahead of the
    queue -- and when the queue holds more than _DW_DEFER_CAP bytes of operands."""
    buckets, _DwDeferred.buckets, _DwDeferred.nbytes = _DwDeferred.buckets, {}, 0
    if not run:
        return
    multi, rest = [], {}
    for key, (gs, xs, x2s, dws, dbs) in buckets.items():
        N, K, R, ct = key[0], key[1], key[2], key[7]
        if ct == BF16 and not dw_long_path(N, K, R, len(gs), ct) and \
                all(tt_multi_ok(g, x, x2, dw, db, N, K, R) for g, x, x2, dw, db in zip(gs, xs, x2s, dws, dbs)):
            multi += list(zip(gs, xs, x2s, dws, dbs))
        else:
            rest[key] = (gs, xs, x2s, dws, dbs)
    if multi and tt_multi_pays(multi):
        tt_multi(multi)
    else:
        rest = buckets
    for (N, K, R, _gd, _xd, has2, hasb, ct), (gs, xs, x2s, dws, dbs) in rest.items():
        tiles = ((N + 63) // 64) * ((K + 63) // 64)
        ga, xa, x2a = dw_operands(gs, xs, x2s if has2 else None, N, K, R, ct)
        for i in range(0, len(ga), L.MAXG):
            n = len(ga[i:i + L.MAXG])
            nkt = max(1, R // (64 if ct == BF16 else 32))
            sk = max(1, min(nkt // 2 if nkt >= 2 else 1, max(1, 768 // max(tiles * n, 1)), 64))
            L.gemm(M=N, N=K, K=R, A=ga[i:i + L.MAXG], B=xa[i:i + L.MAXG], B2=x2a[i:i + L.MAXG] if x2a is not None else None,
                   Cs=dws[i:i + L.MAXG], ct=ct, lda=N, ldb=K, ldc=K, transA=True, transB=True, splitk=max(2, sk),
                   accumulate=True, colsum=dbs[i:i + L.MAXG] if hasb else None)


class _Linear(Function):
    @staticmethod
    def forward(ctx, x, w, b, x2, row_mask, ct, act, out_dtype, fill_flag, fill_value, drop=None, residual=None, masked_grad=None):
        ctx.masked_grad = masked_grad
        x, x2, w, fill_flag, residual = _c(x), _c(x2), _c(w), _c(fill_flag), _c(residual)
        # epilogue order (gemm_common.h): bias, activation, dropout, then "+ residual": y = residual + dropout(x w^T + b).
        # No activation together with a residual: the backward reads the activation's mask off the saved OUTPUT.
        assert residual is None or act is None, "residual add excludes an activation"
        K = x.shape[-1]
        R = x.numel() // K
        N = w.shape[0]
        y = _empty(*x.shape[:-1], N, dtype=out_dtype, device=x.device)
        pre = _empty(y.shape, dtype=out_dtype, device=x.device) if act == "gelu" else None
        rm = _c(row_mask)
        L.gemm(M=R, N=N, K=K, A=[x], A2=[x2], B=[w], bias=[b], Cs=[y], C2=[pre], row_mask=[rm], ct=ct,
               lda=K, ldb=K, ldc=N, act=act, row_fill_flag=fill_flag, row_fill=fill_value, drop=drop,
               aux=[residual] if residual is not None else None, act_grad="add" if residual is not None else None)
        ctx.has_res = residual is not None
        ctx.pptr = (w.data_ptr(), b.data_ptr() if b is not None else None)
        ctx.save_for_backward(x, x2, w, pre if act == "gelu" else (y if act == "relu" else None), rm, fill_flag)
        ctx.ct, ctx.act, ctx.has_b, ctx.drop = ct, act, b is not None, drop
        return y

    @staticmethod
    def backward(ctx, dy):
        x, x2, w, saved, rm, fill_flag = ctx.saved_tensors
        ct = bwd_ct(ctx.ct)
        N, K = w.shape
        R = x.numel() // K
        g = dy.contiguous()
        alpha = 1.0
        if ctx.drop is not None and ctx.act == "relu" and not ctx.has_b:
            # the saved output is post-dropout: [y > 0] already carries the keep mask, only the 1/(1-p) factor is left and
            # rides the two products below as their alpha (no mask-apply launch)
            g = act_bwd(g, saved, ctx.act, act_dtype(ct))
            alpha = 1.0 / (1.0 - ctx.drop.p)
        else:
            if ctx.drop is not None:   # dropout sits after the activation: undo it first (mask * 1/(1-p))
                mg = ctx.masked_grad
                if mg is not None and mg.get("of") is not None and mg["of"].data_ptr() == g.data_ptr() and \
                        mg["of"].shape == g.shape and mg["g"].dtype == g.dtype:
                    g = mg.pop("g")        # written by the consuming norm's backward kernel (pq3d_rmsnorm_bwd_res_drop)
                    mg.pop("of")
                else:
                    g = _dropout_apply(g, ctx.drop)
            if ctx.act in ("relu", "gelu"):
                g = act_bwd(g, saved, ctx.act, act_dtype(ct))
        if rm is not None or fill_flag is not None:
            g = scale_rows(g, R, g.dtype, keep_mask=rm, zero_flag=fill_flag)
        dx = dx2 = dw = db = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[3]:
            dx = _empty(x.shape, dtype=x.dtype, device=x.device)
            # a long reduction over few output tiles (LM head: 512 x 512 outputs, N = 32128) is split over K
            sk = _splitk(((R + 63) // 64) * ((K + 63) // 64), N, ct) if (dx.dtype == torch.float32 and N >= 8192) else 1
            if sk > 1:   # these shapes run on gemm_wk's 64 x 256 tiles: one round of workgroups over the chip (c5's LM head: 8 -> 16
                sk = max(sk, min(64, 256 // max(1, ((R + 63) // 64) * ((K + 255) // 256))))   # slices, 107 -> 81 us)
            L.gemm(M=R, N=K, K=N, A=[g], B=[w], Cs=[dx], ct=ct, lda=N, ldb=K, ldc=K, transB=True, splitk=sk, alpha=alpha)
            dx2 = dx if (x2 is not None and ctx.needs_input_grad[3]) else None
            if not ctx.needs_input_grad[0]:
                dx = None
        want_db = ctx.has_b and ctx.needs_input_grad[2]
        give_w = True
        if ctx.needs_input_grad[1]:
            tiles = ((N + 63) // 64) * ((K + 63) // 64)
            epl = 8 if ct == BF16 else 4
            # the bias gradient rides on the weight-gradient GEMM when its fast (aligned) path applies
            fuse = want_db and N % epl == 0 and K % epl == 0 and N >= epl and K >= epl and \
                g.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0 and (x2 is None or x2.dtype == torch.float32) and \
                (x2 is None or x.dtype == torch.float32)
            # the owner's gradient arena, when offered for this pass: accumulate straight into the slots (pre-zeroed)
            slot, give = arena_take([ctx.pptr[0]] + ([ctx.pptr[1]] if fuse else []), [N * K] + ([N] if fuse else []))
            if slot is not None:
                dw, give_w = slot[0].view(N, K), give
                if fuse:
                    db = slot[1]
                if not (alpha == 1.0 and _dw_defer(g, x, x2, dw, db if fuse else None, N, K, R, ct, pptrs=ctx.pptr)):   # else: queued until the pass ends
                    ga, xa, x2a = dw_operands([g], [x], [x2], N, K, R, ct)   # long reductions: bf16 operands, 128 x 128 tiles
                    L.gemm(M=N, N=K, K=R, A=ga, B=xa, B2=x2a, Cs=[dw], ct=ct, lda=N, ldb=K, ldc=K, transA=True, transB=True,
                           splitk=max(2, _splitk(tiles, R, ct)), colsum=[db] if fuse else None, accumulate=True, alpha=alpha)
            else:
                ga, xa, x2a = dw_operands([g], [x], [x2], N, K, R, ct)   # long reductions: bf16 operands, 128 x 128 tiles
                dw = _empty(N, K, dtype=torch.float32, device=x.device)
                if fuse:
                    db = _empty(N, dtype=torch.float32, device=x.device)
                L.gemm(M=N, N=K, K=R, A=ga, B=xa, B2=x2a, Cs=[dw], ct=ct, lda=N, ldb=K, ldc=K, transA=True,
                       transB=True, splitk=max(2, _splitk(tiles, R, ct)) if fuse else _splitk(tiles, R, ct),
                       colsum=[db] if fuse else None, alpha=alpha)
        if want_db and db is None:
            db = colsum(g.view(R, N))
        dres = dy.contiguous() if (ctx.has_res and ctx.needs_input_grad[11]) else None
        if not give_w:   # accumulated in place into slots autograd already holds
            dw = None
            if db is not None and want_db and fuse:
                db = None
        return dx, dw, db, dx2, None, None, None, None, None, None, None, dres, None


def dw_operands(gs, xs, x2s, N: int, K: int, R: int, ct: int):
    """Operands of weight-gradient products dW[N, K] += g^T (x [+ x2]) over a LONG reduction (R >= 2048 rows: the encoders'
    B * N_seg rows, every projection of the stage-2 shipped shape with its 128 x 80 object rows): fp32 operands are rounded to
    bf16 ONCE by one launch per operand shape -- the rounding the GEMM staging applies anyway, (x + x2) summed in fp32 first
    -- so that the product takes the 128 x 128-tile bf16 kernel (gemm_tt128_kernel: half the operand re-reads of the 64 x 64
    tile, 2 B per element instead of 4; config s2: 155 -> ~490 TFLOP/s on these launches).  Returns (gs, xs, x2s)."""
    # ... when the launch has enough 128 x 128 output tiles to fill the chip without a deep split-K (config 2's input
    # encoders: 3 groups of [256, 256] over 8192 rows = 12 tiles -- 14 us of rounding + 33 us against 29 us on the 64 x 64
    # chunk kernel, measured; config s2: 16-24 groups of [768, 768] = 576+ tiles)
    if ct != BF16 or R < 2048 or R % 64 or N % 128 or K % 128 or (N // 128) * (K // 128) * len(gs) < 64:
        return gs, xs, x2s
    x2s = list(x2s) if x2s is not None else [None] * len(xs)
    if all(t.dtype == torch.bfloat16 for t in list(gs) + list(xs)) and all(t is None for t in x2s):
        return gs, xs, None
    if any(t.dtype != torch.float32 for t in x2s if t is not None) or \
            any(t.dtype == torch.bfloat16 and t2 is not None for t, t2 in zip(xs, x2s)):
        return gs, xs, (x2s if any(t is not None for t in x2s) else None)
    cache, jobs = {}, {}

    def conv(t, t2):
        if t.dtype == torch.bfloat16:
            return t
        key = (t.data_ptr(), t2.data_ptr() if t2 is not None else 0, t.numel())
        o = cache.get(key)
        if o is None:
            if t.numel() % 8 or t.data_ptr() % 16 or (t2 is not None and t2.data_ptr() % 16) or not t.is_contiguous():
                return None
            o = cache[key] = torch.empty(t.shape, dtype=torch.bfloat16, device=t.device)
            jobs.setdefault(t.numel(), []).append((t, t2, o))
        return o
    g2 = [conv(t, None) for t in gs]
    x2 = [conv(t, t2) for t, t2 in zip(xs, x2s)]
    if any(t is None for t in g2 + x2):
        return gs, xs, (x2s if any(t is not None for t in x2s) else None)
    arr = lambda ts: (C.c_void_p * len(ts))(*[L.ptr(t) for t in ts])
    for n, lst in jobs.items():
        for s0 in range(0, len(lst), L.MAXG):
            ch = lst[s0:s0 + L.MAXG]
            L.check(L.lib().pq3d_add_cast(arr([a for a, _, _ in ch]), arr([b for _, b, _ in ch]), arr([o for _, _, o in ch]),
                                          len(ch), L.BF16, n, L.stream()), "pq3d_add_cast")
    return g2, x2, None


def linear(x, w, b=None, *, ct: int, x2=None, act: Optional[str] = None, out_dtype=torch.float32, row_mask=None,
           fill_flag=None, fill_value=0.0, drop: Optional[L.Drop] = None, residual=None, masked_grad: Optional[dict] = None):
    """y = act((x + x2) @ w.T + b); rows where row_mask == False are zeroed; rows where fill_flag == True are
    set to fill_value (masked_fill of whole rows); residual (same shape as y) is added in the GEMM epilogue.
    (F.linear call sites, see include/pq3d_hip.h)
    masked_grad: a hand-over slot shared with the rmsnorm that consumes y (``rmsnorm(..., grad_drop=(drop, slot))``): that
    norm's backward kernel also writes dropout_mask * dy / (1 - p), which this layer's backward then takes instead of
    launching a dropout of its own (checked by address: anything else -- e.g. a second consumer of y, whose gradients autograd
    sums into a new tensor -- falls back to the launch; tests/test_gpu_t5_head.py).  Unsupported: a tensor hook that rewrites the
    norm's input gradient IN PLACE (same address) -- the pre-masked copy was formed from the value before the hook ran."""
    return _Linear.apply(x, w, b, x2, row_mask, ct, act, out_dtype, fill_flag, float(fill_value), drop, residual, masked_grad)


class _LinearGroup(Function):
    """[x_g @ W_g^T] for G bias-free same-shape linears in grouped launches (T5 q/k/v projections; the cross-attention
    K/V projections of ALL decoder layers, which share the encoder tokens).  Backward: one K-concatenated product for
    the inputs that are the same tensor (d x = sum_g dy_g W_g), one grouped split-K product for all weight gradients."""

    @staticmethod
    def forward(ctx, ct, out_dtype, G, *t):
        xs, Ws = [_c(a) for a in t[:G]], [_c(w) for w in t[G:2 * G]]
        K = xs[0].shape[-1]
        R = xs[0].numel() // K
        N = Ws[0].shape[0]
        out = _empty(G, *xs[0].shape[:-1], N, dtype=out_dtype, device=xs[0].device)
        for s in range(0, G, L.MAXG):
            e = min(G, s + L.MAXG)
            L.gemm(M=R, N=N, K=K, A=xs[s:e], B=Ws[s:e], Cs=[out[g] for g in range(s, e)], ct=ct, lda=K, ldb=K, ldc=N)
        ctx.same_x = all(x.data_ptr() == xs[0].data_ptr() for x in xs)
        ctx.save_for_backward(*xs, *Ws)
        ctx.cfg = (ct, G)
        ctx.pptr = [w.data_ptr() for w in t[G:2 * G]]
        return tuple(out[g] for g in range(G))

    @staticmethod
    def backward(ctx, *dys):
        ct, G = ctx.cfg
        ct = bwd_ct(ct)
        xs, Ws = ctx.saved_tensors[:G], ctx.saved_tensors[G:]
        K = xs[0].shape[-1]
        R = xs[0].numel() // K
        N = Ws[0].shape[0]
        dev = xs[0].device
        ad = act_dtype(ct)
        gs = [(_c(g).to(ad) if g is not None else torch.zeros(R, N, dtype=ad, device=dev)) for g in dys]
        need_dx = any(ctx.needs_input_grad[3 + g] for g in range(G))
        dxs = [None] * G
        if need_dx and ctx.same_x and G <= L.MAXG:      # d x = sum_g dy_g W_g: one K-concatenated product
            dx = _empty(xs[0].shape, dtype=xs[0].dtype, device=dev)
            L.gemm(M=R, N=K, K=N, A=gs, B=list(Ws), Cs=[dx] + [None] * (G - 1), ct=ct, lda=N, ldb=K, ldc=K, transB=True,
                   kconcat=G)
            dxs[0] = dx          # the same tensor was passed G times: autograd sums the slots, so only one carries it
        elif need_dx:
            dxb = _empty(G, *xs[0].shape, dtype=xs[0].dtype, device=dev)
            for s in range(0, G, L.MAXG):
                e = min(G, s + L.MAXG)
                L.gemm(M=R, N=K, K=N, A=gs[s:e], B=list(Ws[s:e]), Cs=[dxb[g] for g in range(s, e)], ct=ct, lda=N, ldb=K,
                       ldc=K, transB=True)
            if ctx.same_x:
                dxs[0] = dxb.sum(0)
            else:
                dxs = [dxb[g] for g in range(G)]
        slot, give = arena_take(ctx.pptr, [N * K] * G)   # the owner's gradient arena, when offered for this pass (pre-zeroed)
        if slot is not None:
            dWs = [v.view(N, K) for v in slot]
        else:
            give = True
            dWb = torch.zeros(G, N, K, dtype=torch.float32, device=dev)
            dWs = [dWb[g] for g in range(G)]
        tiles = ((N + 63) // 64) * ((K + 63) // 64)
        if slot is not None and all(_dw_can_defer(gs[g], xs[g], None, N, K) for g in range(G)) and \
                not any(_param_has_hooks(q) for q in ctx.pptr):
            for g in range(G):
                _dw_defer(gs[g], xs[g], None, dWs[g], None, N, K, R, ct)
            return (None, None, None, *dxs, *(dWs if give else [None] * G))   # queued until the pass ends
        gs, xs_, _ = dw_operands(list(gs), list(xs), None, N, K, R, ct)
        for s in range(0, G, L.MAXG):
            e = min(G, s + L.MAXG)
            L.gemm(M=N, N=K, K=R, A=gs[s:e], B=list(xs_[s:e]), Cs=dWs[s:e], ct=ct, lda=N, ldb=K, ldc=K,
                   transA=True, transB=True, splitk=max(2, _splitk(tiles * (e - s), R, ct)), accumulate=True)
        return (None, None, None, *dxs, *(dWs if give else [None] * G))


def linear_group(xs, Ws, *, ct: int, out_dtype=torch.float32):
    """[x_g @ W_g^T for g] -- bias-free linears of one shape in grouped launches."""
    G = len(xs)
    return _LinearGroup.apply(ct, out_dtype, G, *xs, *Ws)


# ------------------------------------------------------------------------------------------------ attention
def _attn_desc(q, k, v, o, lse, H, ct, zero_attn, scale, kpm, mask, row_open, bias, drop=None, drop_bmod=0,
               bwd=False, mask_bits=None) -> L.AttnDesc:
    B, Lq, dm = q.shape
    Lk = k.shape[1]
    d = L.AttnDesc()
    d.B, d.H, d.Lq, d.Lk, d.dh = B, H, Lq, Lk, dm // H
    d.ct, d.dt, d.zero_attn, d.scale = ct, L.dt_of(q), int(zero_attn), scale
    for name, t in (("q", q), ("k", k), ("v", v), ("o", o)):
        assert t.stride(-1) == 1
        setattr(d, name + "_sb", t.stride(0)); setattr(d, name + "_sl", t.stride(1)); setattr(d, name + "_sh", dm // H)
    d.q, d.k, d.v, d.o, d.lse = L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(o), L.ptr(lse)
    d.kpm, d.mask, d.row_open, d.bias = L.ptr(kpm), L.ptr(mask), L.ptr(row_open), L.ptr(bias)
    d.mask_bits = L.ptr(mask_bits) if mask is not None else None
    L.set_drop(d.drop, drop)
    d.drop_bmod = drop_bmod
    # key split (not with dbias).  The factor depends on the key length ONLY, never on the batch: a scene's result
    # must not change with how scenes are batched or sharded over ranks (tests/test_gpu_fullsize.py).  Measured
    # (tools/probes/attn_bench.py + bench.py): c2 (16 key blocks) and c5 (32) are fastest with 2 splits, c4 (64) with 4
    # (forward 70 -> 57 us, step 5.56 -> 5.43 ms); 8 only adds combine traffic.
    nkb = (Lk + 63) // 64
    ks = (1 if nkb < 8 else 2 if nkb < 64 else 4) if bias is None else 1
    if not bwd and ct == BF16 and q.dtype == torch.bfloat16 and bias is None and mask is None and \
            Lq <= 128 <= Lk and dm // H == 32 and nkb <= 16:
        # the all-keys-resident forward (attn_resident.hip) holds up to 1024 keys per workgroup: config 2 needs no split
        # (and no combine launch).  Longer scenes keep the streaming kernel and its split (measured at config 5, 2048 keys:
        # resident with 2 splits 63-76 us vs streaming 58 us) -- a function of the key length only, as above
        ks = 1
    if bwd and ks > 1 and ct == BF16 and bias is None and Lq <= 128 and (dm // H) in (32, 64) and \
            (B * H >= 320 or (B * H <= 256 and Lk >= 512 and dm // H == 32)):
        # the all-queries-resident backward (attn_resident.hip) runs one workgroup per (scene, head, slice): once the
        # stacked batch alone fills the chip (config 5: 48 x 8) a second slice only adds dQ partials (85 -> 75 us).
        # Gradients are summed with atomics downstream anyway, so this may depend on the batch; the forward's may not.
        # At most one workgroup per CU (config 2: 24 x 8): the 8-wave variant of the kernel takes all keys of a (scene,
        # head) -- no dQ partials, no combine launch.
        ks = 1
    if bwd and ks > 2 and ct == BF16 and bias is None and 128 < Lq <= 256 and dm // H == 32 and \
            B * H * (ks // 2) <= 256 and Lk // (ks // 2) >= 512:
        ks //= 2      # the 8-wave resident backward (two query halves, config 4): half the key slices, one workgroup per CU
    if bwd and ct == BF16 and bias is None and Lq <= 128 and dm // H == 64 and nkb >= 16 and B * H * ks < 512:
        # d_h = 64 resident backward (4 waves per workgroup, the shipped stage-1 decoder: 12 x 12 (scene, head) pairs,
        # 2048 keys): at least two workgroups per CU -- measured 2 / 4 / 8 / 16 slices: 15.12 / 14.82 / 14.97 / 15.45 ms
        ks = min(8, -(-512 // (B * H)))
    if ks > 1:
        ws = _empty(ks * B * H * Lq * (dm // H + 2), dtype=torch.float32, device=q.device)
        d.ksplit, d.ws = ks, L.ptr(ws)
        d._ws_keepalive = ws
    return d


class _Attention(Function):
    @staticmethod
    def forward(ctx, q, k, v, bias, kpm, mask, row_open, H, zero_attn, scale, ct, drop=None):
        q, k, v, bias, kpm, mask, row_open = map(_c, (q, k, v, bias, kpm, mask, row_open))
        B, Lq, dm = q.shape
        o = _empty(q.shape, dtype=q.dtype, device=q.device)
        lse = _empty(B, H, Lq, dtype=torch.float32, device=q.device)
        d = _attn_desc(q, k, v, o, lse, H, ct, zero_attn, scale, kpm, mask, row_open, bias, drop)
        Lk = k.shape[1]
        fl = 4.0 * B * Lq * Lk * dm
        nb = (q.numel() * 2 + k.numel() * 2) * q.element_size()
        L.check(timed("pq3d_attn_fwd", f"B{B}H{H}Lq{Lq}Lk{Lk}dh{dm // H}ct{ct}", fl, nb, L.lib().pq3d_attn_fwd,
                      C.byref(d), L.stream()), "pq3d_attn_fwd")
        ctx.save_for_backward(q, k, v, o, lse, bias, kpm, mask, row_open)
        ctx.cfg = (H, zero_attn, scale, ct)
        ctx.drop = drop
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse, bias, kpm, mask, row_open = ctx.saved_tensors
        H, zero_attn, scale, ct = ctx.cfg
        do = do.contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        delta = torch.empty_like(lse)
        dbias = torch.empty_like(bias) if (bias is not None and ctx.needs_input_grad[3]) else None
        d = _attn_desc(q, k, v, o, lse, H, ct, zero_attn, scale, kpm, mask, row_open, bias, ctx.drop, bwd=True)
        d.dout, d.dq, d.dk, d.dv, d.delta, d.dbias = map(L.ptr, (do, dq, dk, dv, delta, dbias))
        B, Lq, dm = q.shape
        Lk = k.shape[1]
        # ALGORITHMIC flops (backward = 2 x forward = 8 B Lq Lk d); the two recompute kernels EXECUTE 14:
        # dQ kernel S, dP, dQ (6) + dK/dV kernel S, dP, dK, dV (8)
        fl = 8.0 * B * Lq * Lk * dm
        nb = (q.numel() * 3 + k.numel() * 4) * q.element_size()
        L.check(timed("pq3d_attn_bwd", f"B{B}H{H}Lq{Lq}Lk{Lk}dh{dm // H}ct{ct}", fl, nb, L.lib().pq3d_attn_bwd,
                      C.byref(d), L.stream()), "pq3d_attn_bwd")
        return dq, dk, dv, dbias, None, None, None, None, None, None, None, None


def attention(q, k, v, *, H: int, ct: int, scale: Optional[float] = None, zero_attn=False, kpm=None, mask=None,
              row_open=None, bias=None, drop: Optional[L.Drop] = None):
    """softmax(scale q.k^T + bias + masks [, zero key]) v over heads packed in the last dim (see pq3d_attn_fwd)."""
    if scale is None:
        scale = 1.0 / math.sqrt(q.shape[-1] // H)
    return _Attention.apply(q, k, v, bias, kpm, mask, row_open, H, bool(zero_attn), float(scale), ct, drop)


# ------------------------------------------------------------------------------------------------ add + layernorm
def _ln_desc(x, os_, gammas, betas, coef, eps, rows_per_scene, y, mean, rstd, drop=None) -> L.LnDesc:
    d = L.LnDesc()
    L.set_drop(d.drop, drop)
    dm = os_[0].shape[-1]
    d.R, d.d, d.M, d.rows_per_scene = os_[0].numel() // dm, dm, len(os_), rows_per_scene
    d.dt_x = L.dt_of(x) if x is not None else 0
    d.dt_o, d.dt_y, d.eps = L.dt_of(os_[0]), (L.dt_of(y) if y is not None else 0), eps
    d.x, d.coef, d.y, d.mean, d.rstd = L.ptr(x), L.ptr(coef), L.ptr(y), L.ptr(mean), L.ptr(rstd)
    for m in range(len(os_)):
        d.o[m] = L.ptr(os_[m])
    for m in range(len(gammas)):     # fewer than len(os_) in sum_branches mode (one LayerNorm over partial sums)
        d.gamma[m], d.beta[m] = L.ptr(gammas[m]), L.ptr(betas[m])
    return d


class _AddLN(Function):
    @staticmethod
    def forward(ctx, x, coef, eps, rows_per_scene, out_dtype, M, drop, sum_branches, *t):
        G = 1 if sum_branches else M      # sum_branches: the M inputs are partial sums of ONE branch -> one gamma / beta
        os_ = [_c(a) for a in t[:M]]
        gammas, betas = [_c(a) for a in t[M:M + G]], [_c(a) for a in t[M + G:M + 2 * G]]
        x, coef = _c(x), _c(coef)
        dm = os_[0].shape[-1]
        R = os_[0].numel() // dm
        y = _empty(os_[0].shape, dtype=out_dtype, device=os_[0].device)
        mean = _empty(M, R, dtype=torch.float32, device=y.device)
        rstd = torch.empty_like(mean)
        d = _ln_desc(x, os_, gammas, betas, coef, eps, rows_per_scene, y, mean, rstd, drop)
        d.sum_branches = int(sum_branches)
        nb = (M + 1 + (x is not None)) * R * dm * 4.0
        L.check(timed("pq3d_add_ln_fwd", f"R{R}d{dm}M{M}", 0.0, nb, L.lib().pq3d_add_ln_fwd, C.byref(d), L.stream()),
                "pq3d_add_ln_fwd")
        ctx.save_for_backward(x, coef, mean, rstd, *os_, *gammas, *betas)
        ctx.cfg = (eps, rows_per_scene, M, G, bool(sum_branches))
        ctx.drop = drop
        return y

    @staticmethod
    def backward(ctx, dy):
        eps, rows_per_scene, M, G, sum_branches = ctx.cfg
        x, coef, mean, rstd = ctx.saved_tensors[:4]
        t = ctx.saved_tensors[4:]
        os_, gammas, betas = t[:M], t[M:M + G], t[M + G:M + 2 * G]
        dy = dy.contiguous().float()
        dev = dy.device
        dx = _empty(os_[0].shape, dtype=torch.float32, device=dev) if x is not None else None
        d_os = [_empty(o.shape, dtype=torch.float32, device=dev) for o in os_[:G]]
        dgs = [torch.empty_like(g) for g in gammas]
        dbs = [torch.empty_like(b) for b in betas]
        d = _ln_desc(x, os_, gammas, betas, coef, eps, rows_per_scene, None, mean, rstd, ctx.drop)
        d.dy, d.dx, d.sum_branches = L.ptr(dy), L.ptr(dx), int(sum_branches)
        for m in range(G):
            d.d_o[m], d.dgamma[m], d.dbeta[m] = L.ptr(d_os[m]), L.ptr(dgs[m]), L.ptr(dbs[m])
        R, dm = os_[0].numel() // os_[0].shape[-1], os_[0].shape[-1]
        nb = (3 * M + 1 + (x is not None)) * R * dm * 4.0
        L.check(timed("pq3d_add_ln_bwd", f"R{R}d{dm}M{M}", 0.0, nb, L.lib().pq3d_add_ln_bwd, C.byref(d), L.stream()),
                "pq3d_add_ln_bwd")
        if x is not None and x.dtype != torch.float32:
            dx = dx.to(x.dtype)
        if sum_branches:      # every partial sum receives the gradient of the sum
            d_os = [d_os[0]] * M
        d_os = [g if g.dtype == o.dtype else g.to(o.dtype) for g, o in zip(d_os, os_)]
        return (dx, None, None, None, None, None, None, None, *d_os, *dgs, *dbs)


def add_layernorm(x, os_: Sequence[torch.Tensor], gammas, betas, *, eps=1e-5, coef=None, rows_per_scene=None,
                  out_dtype=torch.float32, drop: Optional[L.Drop] = None, sum_branches: bool = False):
    """y = sum_m coef[m, scene] * LN_m(x + dropout_m(o_m))   (coef None -> mean over the M branches; x may be None;
    branch m draws dropout site drop.site + m).  sum_branches: os_ are partial sums of ONE branch (a K-split GEMM):
    y = LN(x + dropout(sum_m o_m)) with gammas[0] / betas[0]."""
    M = len(os_)
    if rows_per_scene is None:
        rows_per_scene = os_[0].shape[-2] if os_[0].dim() >= 2 else 1
    return _AddLN.apply(x, coef, float(eps), int(rows_per_scene), out_dtype, M, drop, bool(sum_branches), *os_, *gammas,
                        *betas)


# ------------------------------------------------------------------------------------------------ mask logits
class _MaskLogits(Function):
    @staticmethod
    def forward(ctx, inv_den, seg_pad, ct, M, *kq):
        ks, qs = [_c(a) for a in kq[:M]], [_c(a) for a in kq[M:]]
        B, Ns, dm = ks[0].shape
        Nq = qs[0].shape[1]
        logits = _empty(B, Ns, Nq, dtype=torch.float32, device=ks[0].device)
        amask = _empty(B, Nq, Ns, dtype=torch.bool, device=ks[0].device)
        L.gemm(M=Ns, N=Nq, K=dm, A=ks, B=qs, Cs=[logits] + [None] * (M - 1), ct=ct, lda=dm, ldb=dm, ldc=Nq, batch=B,
               strideA=Ns * dm, strideB=Nq * dm, strideC=Ns * Nq, kconcat=True, row_scale=inv_den,
               row_fill_flag=seg_pad, row_fill=-1e6, mask_out=amask)
        ctx.save_for_backward(inv_den, seg_pad, *ks, *qs)
        ctx.cfg = (ct, M)
        ctx.mark_non_differentiable(amask)
        return logits, amask

    @staticmethod
    def backward(ctx, dl, _dmask):
        ct, M = ctx.cfg
        ct = bwd_ct(ct)
        inv_den, seg_pad = ctx.saved_tensors[:2]
        ks, qs = ctx.saved_tensors[2:2 + M], ctx.saved_tensors[2 + M:]
        B, Ns, dm = ks[0].shape
        Nq = qs[0].shape[1]
        g = scale_rows(dl.contiguous(), B * Ns, act_dtype(ct), scale=inv_den, zero_flag=seg_pad)
        dks = [torch.empty_like(k) for k in ks]
        dqs = [torch.empty_like(q) for q in qs]
        L.gemm(M=Ns, N=dm, K=Nq, A=[g] * M, B=list(qs), Cs=dks, ct=ct, lda=Nq, ldb=dm, ldc=dm, transB=True, batch=B,
               strideA=Ns * Nq, strideB=Nq * dm, strideC=Ns * dm)
        L.gemm(M=Nq, N=dm, K=Ns, A=[g] * M, B=list(ks), Cs=dqs, ct=ct, lda=Nq, ldb=dm, ldc=dm, transA=True,
               transB=True, batch=B, strideA=Ns * Nq, strideB=Ns * dm, strideC=Nq * dm)
        return (None, None, None, None, *dks, *dqs)


def mask_logits(ks: Sequence[torch.Tensor], qs: Sequence[torch.Tensor], inv_den, seg_pad, *, ct: int):
    """mask_head.py:30-43: logits[b,s,q] = pad ? -1e6 : inv_den[b,s] * sum_m k_m[b,s,:].q_m[b,q,:];
    attn_mask[b,q,s] = sigmoid(logits) < 0.5.  k_m rows of invalid segments must already be zero."""
    return _MaskLogits.apply(inv_den, seg_pad, ct, len(ks), *ks, *qs)


# ------------------------------------------------------------------------------------------------ spatial bias
class _SpatialBias(Function):
    @staticmethod
    def forward(ctx, pl, W, bw):
        pl, W, bw = _c(pl), _c(W), _c(bw)
        B, Lq = pl.shape[:2]
        H = W.shape[0]
        bias = _empty(B, H, Lq, Lq, dtype=torch.float32, device=pl.device)
        L.check(L.lib().pq3d_spatial_bias_fwd(L.ptr(pl), L.ptr(W), L.ptr(bw), L.ptr(bias), B, H, Lq, L.stream()),
                "pq3d_spatial_bias_fwd")
        ctx.save_for_backward(pl, W, bw)
        return bias

    @staticmethod
    def backward(ctx, dbias):
        pl, W, bw = ctx.saved_tensors
        B, Lq = pl.shape[:2]
        dW, dbw = torch.empty_like(W), torch.empty_like(bw)
        L.check(L.lib().pq3d_spatial_bias_bwd(L.ptr(pl), L.ptr(W), L.ptr(bw), L.ptr(dbias.contiguous()), L.ptr(dW),
                                              L.ptr(dbw), B, W.shape[0], Lq, L.stream()), "pq3d_spatial_bias_bwd")
        return None, dW, dbw


def spatial_bias(pl, W, bw):
    """log(clamp(relu(pairwise_loc_fc(pl)), 1e-6)) laid out [B,H,L,L] (transformers.py:196-200,226)."""
    return _SpatialBias.apply(pl, W, bw)


# ------------------------------------------------------------------------------------------------ misc differentiable
class _GateMix(Function):
    @staticmethod
    def forward(ctx, q, u, g):
        q, u, g = _c(q), _c(u), _c(g)
        y = torch.empty_like(q)
        L.check(L.lib().pq3d_gate_mix_fwd(L.ptr(q), L.ptr(u), L.ptr(g), L.ptr(y), q.numel(), L.stream()),
                "pq3d_gate_mix_fwd")
        ctx.save_for_backward(q, u, g)
        return y

    @staticmethod
    def backward(ctx, dy):
        q, u, g = ctx.saved_tensors
        dq, du, dg = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
        L.check(L.lib().pq3d_gate_mix_bwd(L.ptr(q), L.ptr(u), L.ptr(g), L.ptr(dy.contiguous()), L.ptr(dq), L.ptr(du),
                                          L.ptr(dg), q.numel(), L.stream()), "pq3d_gate_mix_bwd")
        return dq, du, dg


def gate_mix(q, u, g):
    """(1 - sigmoid(g)) * q + sigmoid(g) * u   (query_encoder.py:167-170)."""
    return _GateMix.apply(q, u, g)


class _FillCols(Function):
    @staticmethod
    def forward(ctx, x, cols, value):
        x = _c(x)
        y = torch.empty_like(x)
        C_ = x.shape[-1]
        L.check(L.lib().pq3d_fill_cols(L.ptr(x), L.ptr(y), x.numel() // C_, C_, L.ptr(cols), cols.numel(), value,
                                       L.stream()), "pq3d_fill_cols")
        ctx.save_for_backward(cols)
        return y

    @staticmethod
    def backward(ctx, dy):
        (cols,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        C_ = dy.shape[-1]
        L.check(L.lib().pq3d_fill_cols(L.ptr(dy), L.ptr(dx), dy.numel() // C_, C_, L.ptr(cols), cols.numel(), 0.0,
                                       L.stream()), "pq3d_fill_cols")
        return dx, None, None


def fill_cols(x, cols: torch.Tensor, value: float):
    """x[..., cols] = value, out of place (mask_head.py:28)."""
    return _FillCols.apply(x, cols, float(value))


class SegmentPlan:
    """The voxel -> segment grouping of a batch, sorted ONCE on the device (pq3d_segment_plan): serves every reduction over
    it -- the 5 feature levels of PCDMask3DSegLevelEncoder (pcd_mask3d_encoder.py:142-150) forward, and through
    ``child(parent)`` (the grouping of the fine voxels by their coarse ancestor) the gradient of an up-sampled level.
    ``index`` [N] int64 (ids outside [0, dim_size) are dropped), batched scenes = ids offset by b * max_seg."""

    def __init__(self, index: torch.Tensor, dim_size: int):
        assert index.dtype == torch.int64 and index.dim() == 1
        self.index, self.N, self.S = _c(index), int(index.numel()), int(dim_size)
        nbytes = int(L.lib().pq3d_segment_plan_bytes(self.N, self.S))
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device=index.device)
        L.check(L.lib().pq3d_segment_plan(L.ptr(self.index), self.N, self.S, L.ptr(self.buf), nbytes, L.stream()),
                "pq3d_segment_plan")
        self._ws = {}
        self._children = {}

    def ws(self, C_: int) -> torch.Tensor:
        if C_ not in self._ws:
            self._ws[C_] = torch.empty(int(L.lib().pq3d_segment_ws_bytes(self.N, self.S, C_)), dtype=torch.uint8,
                                       device=self.buf.device)
        return self._ws[C_]

    def child(self, parent: torch.Tensor, n_coarse: int) -> "SegmentPlan":
        """Plan of the same voxels grouped by ``parent`` (rows of a coarse level); cached per parent tensor."""
        key = (parent.data_ptr(), int(n_coarse))
        if key not in self._children:
            self._children[key] = (SegmentPlan(parent, n_coarse), parent)    # keeps `parent` alive: the key is its address
        return self._children[key][0]

    def reduce(self, src, gather, row_div, C_, mean, want_count=True):
        out = _empty(self.S, C_, dtype=torch.float32, device=src.device)
        count = _empty(self.S, dtype=torch.float32, device=src.device) if want_count else None
        ws = self.ws(C_)
        # algorithmic bytes: N*C*4 (one row per voxel, as if the up-sampled level were read) + N*8 (ids) +
        # S*C*4 (result)
        nb = self.N * C_ * 4.0 + self.N * 8.0 + self.S * C_ * 4.0
        L.check(timed("pq3d_segment_reduce", f"N{self.N}S{self.S}C{C_}{'g' if gather is not None else ''}", 0.0, nb,
                      L.lib().pq3d_segment_reduce, L.ptr(src), src.shape[0], L.ptr(gather), L.ptr(row_div), L.ptr(self.buf),
                      self.N, self.S, C_, int(mean), L.ptr(out), L.ptr(count), L.ptr(ws), ws.numel(), L.stream()),
                "pq3d_segment_reduce")
        return out, count


def segment_gather(table: torch.Tensor, index: torch.Tensor, count: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[v,:] = table[index[v],:] (/ max(count[index[v]], 1)): the evaluator's mask[voxel2segment]
    (evaluator/instseg_eval.py:101,272-281) and the gradient of the segment mean; ids outside the table give zero rows.
    No autograd (an inference / backward helper)."""
    assert table.dtype == torch.float32 and index.dtype == torch.int64 and table.dim() == 2 and index.dim() == 1
    table, index = _c(table), _c(index)
    out = _empty(index.numel(), table.shape[1], dtype=torch.float32, device=table.device)
    N, S, C_ = index.numel(), table.shape[0], table.shape[1]
    L.check(timed("pq3d_segment_gather", f"N{N}S{S}C{C_}", 0.0, N * C_ * 4.0 + N * 8.0 + S * C_ * 4.0,
                  L.lib().pq3d_segment_gather, L.ptr(table), L.ptr(index), L.ptr(count), L.ptr(out), N, S, C_, L.stream()),
            "pq3d_segment_gather")
    return out


class _ScatterMean(Function):
    @staticmethod
    def forward(ctx, src, plan):
        src = _c(src)
        out, count = plan.reduce(src, None, None, src.shape[1], True)
        ctx.plan, ctx.count = plan, count
        return out

    @staticmethod
    def backward(ctx, dout):
        plan = ctx.plan
        return segment_gather(dout.contiguous(), plan.index, ctx.count), None


def scatter_mean(src: torch.Tensor, index: torch.Tensor, dim_size: int, plan: Optional[SegmentPlan] = None) -> torch.Tensor:
    """torch_scatter.scatter_mean(src, index, dim=0, dim_size) for [N,C] fp32 voxel features
    (pcd_mask3d_encoder.py:149).  ``plan``: a SegmentPlan of (index, dim_size) built once per batch and shared by every
    level; built here when absent."""
    assert src.dtype == torch.float32 and index.dtype == torch.int64 and src.dim() == 2
    assert src.shape[0] == index.numel()
    if plan is None:
        plan = SegmentPlan(index, int(dim_size))
    assert plan.N == index.numel() and plan.S == int(dim_size)
    return _ScatterMean.apply(src, plan)


class _UpsampleScatterMean(Function):
    @staticmethod
    def forward(ctx, src, parent, plan):
        src, parent = _c(src), _c(parent)
        out, count = plan.reduce(src, parent, None, src.shape[1], True)
        ctx.plan, ctx.parent, ctx.count, ctx.nc = plan, parent, count, src.shape[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        # dsrc[p,:] = sum over the fine voxels v below coarse row p of dout[index[v],:] / count[index[v]]: the same
        # reduction, grouped by the parent instead of the segment (no atomics: deterministic gradients)
        plan, dout = ctx.plan, dout.contiguous()
        pplan = plan.child(ctx.parent, ctx.nc)
        dsrc, _ = pplan.reduce(dout, plan.index, ctx.count, dout.shape[1], False, want_count=False)
        return dsrc, None, None


def upsample_scatter_mean(src: torch.Tensor, parent: torch.Tensor, index: torch.Tensor, dim_size: int,
                          plan: Optional[SegmentPlan] = None) -> torch.Tensor:
    """scatter_mean(upsample(src), index): ``src`` [Nc, C] are the features of a COARSE voxel level, ``parent`` [N] the
    coarse row of every full-resolution voxel (the composition of the stride-2 pooling maps, see
    ``compose_parents`` / ``parents_from_coords``), ``index`` [N] its segment (pcd_mask3d_encoder.py:133-152 without
    the up-sampled intermediate).  Voxels whose parent or segment is out of range contribute nothing and are not counted."""
    assert src.dtype == torch.float32 and parent.dtype == torch.int64 and index.dtype == torch.int64 and src.dim() == 2
    assert parent.numel() == index.numel()
    if plan is None:
        plan = SegmentPlan(index, int(dim_size))
    assert plan.N == index.numel() and plan.S == int(dim_size)
    return _UpsampleScatterMean.apply(src, parent, plan)


def compose_parents(maps) -> torch.Tensor:
    """Fine -> coarse row index through a chain of per-level parent maps [fine->l1, l1->l2, ...] (index composition in
    place of repeated MinkowskiPoolingTranspose)."""
    p = maps[0]
    for m in maps[1:]:
        p = m[p]
    return p


def parents_from_coords(fine: torch.Tensor, coarse: torch.Tensor, stride: int) -> torch.Tensor:
    """Row of ``coarse`` [Nc, 1+3] (batch, x, y, z; tensor stride ``stride``) that holds each voxel of ``fine`` [N, 1+3]:
    the coarse voxel at floor(xyz / stride) * stride in the same batch item -- the coordinate map a stride-2^k
    MinkowskiEngine pooling / transposed pooling pair uses.  -1 where the coarse level has no such voxel."""
    def key(c):
        c = c.long()
        q = torch.div(c[:, 1:], stride, rounding_mode="floor") + (1 << 19)
        return ((c[:, 0] << 60) | (q[:, 0] << 40) | (q[:, 1] << 20) | q[:, 2])
    kc, kf = key(coarse), key(fine)
    order = torch.argsort(kc)
    pos = torch.searchsorted(kc[order], kf).clamp_(max=kc.numel() - 1)
    hit = order[pos]
    return torch.where(kc[hit] == kf, hit, torch.full_like(hit, -1))


# ------------------------------------------------------------------------------------------------ grouped Linear + LN
class _LinearLNGroup(Function):
    """G independent nn.Sequential(Linear, LayerNorm) encoders of identical shape in 2 launches forward
    (grouped GEMM, independent-branch LayerNorm) and 2 backward (LayerNorm backward, grouped weight-gradient GEMM with
    fused bias gradients) -- ObjectEncoder.input_feat_proj of every scene memory (object_encoder.py:34,71)."""

    @staticmethod
    def forward(ctx, ct, eps, G, need_dx, *t):
        xs = [_c(a) for a in t[:G]]
        Ws, bs, gam, bet = t[G:2 * G], t[2 * G:3 * G], t[3 * G:4 * G], t[4 * G:5 * G]
        K = xs[0].shape[-1]
        R = xs[0].numel() // K
        N = Ws[0].shape[0]
        dev = xs[0].device
        lin = _empty(G, *xs[0].shape[:-1], N, dtype=torch.float32, device=dev)
        L.gemm(M=R, N=N, K=K, A=xs, B=[_c(w) for w in Ws], bias=list(bs), Cs=[lin[g] for g in range(G)], ct=ct, lda=K,
               ldb=K, ldc=N)
        ys = _empty(G, *xs[0].shape[:-1], N, dtype=torch.float32, device=dev)
        mean = _empty(G, R, dtype=torch.float32, device=dev)
        rstd = torch.empty_like(mean)
        d = _ln_desc(None, [lin[g] for g in range(G)], [_c(a) for a in gam], [_c(a) for a in bet], None, eps, R, None,
                     mean, rstd)
        d.independent = 1
        d.dt_y = F32
        for g in range(G):
            d.ys[g] = L.ptr(ys[g])
        L.check(timed("pq3d_add_ln_fwd", f"R{R}d{N}M{G}i", 0.0, 2.0 * G * R * N * 4, L.lib().pq3d_add_ln_fwd, C.byref(d),
                      L.stream()), "pq3d_add_ln_fwd")
        ctx.save_for_backward(lin, mean, rstd, *xs, *Ws, *gam, *bet)
        ctx.cfg = (ct, eps, G, need_dx)
        ctx.pptr = [a.data_ptr() for a in (*Ws, *bs, *gam, *bet)]   # order of the returned parameter gradients
        return tuple(ys[g] for g in range(G))

    @staticmethod
    def backward(ctx, *dys):
        ct, eps, G, need_dx = ctx.cfg
        ct = bwd_ct(ct)
        lin, mean, rstd = ctx.saved_tensors[:3]
        t = ctx.saved_tensors[3:]
        xs, Ws, gam, bet = t[:G], t[G:2 * G], t[2 * G:3 * G], t[3 * G:4 * G]
        K = xs[0].shape[-1]
        R = xs[0].numel() // K
        N = Ws[0].shape[0]
        dev = lin.device
        dys = [(_c(g).float() if g is not None else torch.zeros(lin.shape[1:], device=dev)) for g in dys]
        dlin = _empty(lin.shape, dtype=torch.float32, device=dev)
        # every atomics target of this backward (LayerNorm parameter gradients, split-K weight gradients, bias column
        # sums): the parameters' slots of the owner's gradient arena when the decoder's backward offered them for this
        # pass (zeroed by its one launch: no fill here, no pack copy later), else ONE zero-filled buffer of our own
        tiles = ((N + 63) // 64) * ((K + 63) // 64)
        epl = 8 if ct == BF16 else 4
        fuse = N % epl == 0 and K % epl == 0 and all(x.data_ptr() % 16 == 0 for x in xs)
        slot, give = arena_take(ctx.pptr, [N * K] * G + [N] * (3 * G)) if fuse else (None, False)
        if slot is not None:
            dWs, dbl, dgs, dbs = slot[:G], slot[G:2 * G], slot[2 * G:3 * G], slot[3 * G:]
            dWs = [w.view(N, K) for w in dWs]
        else:
            give = True
            zb = torch.zeros(G * (3 * N + N * K), dtype=torch.float32, device=dev)
            dgs = [zb[g * N:(g + 1) * N] for g in range(G)]
            dbs = [zb[(G + g) * N:(G + g + 1) * N] for g in range(G)]
            dbl = [zb[(2 * G + g) * N:(2 * G + g + 1) * N] for g in range(G)]
            dWs = [zb[3 * G * N + g * N * K:3 * G * N + (g + 1) * N * K].view(N, K) for g in range(G)]
        d = _ln_desc(None, [lin[g] for g in range(G)], list(gam), list(bet), None, eps, R, None, mean, rstd)
        d.independent = 1
        d.accumulate = 1
        for g in range(G):
            d.dys[g], d.d_o[g], d.dgamma[g], d.dbeta[g] = L.ptr(dys[g]), L.ptr(dlin[g]), L.ptr(dgs[g]), L.ptr(dbs[g])
        L.check(timed("pq3d_add_ln_bwd", f"R{R}d{N}M{G}i", 0.0, 3.0 * G * R * N * 4, L.lib().pq3d_add_ln_bwd, C.byref(d),
                      L.stream()), "pq3d_add_ln_bwd")
        if fuse and ct == BF16 and not dw_long_path(N, K, R, G, ct) and \
                all(tt_multi_ok(dlin[g], xs[g], None, dWs[g], dbl[g], N, K, R) for g in range(G)):
            # short reductions only (tt_multi_ok refuses R >= 2048 unless PQ3D_TT_MULTI_LONG=1: config 2's 3 encoders of [256 x 256]
            # over 8192 rows were measured slower on wide tiles + k-slices): the encoders of small batches / short memories --
            # one launch with the fused bias gradient (tests/test_gpu_ops.py::test_linear_ln_group_backward_tt_multi_path)
            tt_multi([(dlin[g], xs[g], None, dWs[g], dbl[g]) for g in range(G)])
        else:
            ga_, xa_, _ = dw_operands([dlin[g] for g in range(G)], list(xs), None, N, K, R, ct)
            L.gemm(M=N, N=K, K=R, A=ga_, B=xa_, Cs=dWs, ct=ct, lda=N, ldb=K, ldc=K, transA=True,
                   transB=True, splitk=max(2, _splitk(tiles * G, R, ct)), colsum=dbl if fuse else None, accumulate=True)
        if not fuse:
            dbl = [colsum(dlin[g].view(R, N)) for g in range(G)]
        dxs = [None] * G
        if need_dx:
            dxb = _empty(G, *xs[0].shape, dtype=xs[0].dtype, device=dev)
            L.gemm(M=R, N=K, K=N, A=[dlin[g] for g in range(G)], B=list(Ws), Cs=[dxb[g] for g in range(G)], ct=ct,
                   lda=N, ldb=K, ldc=K, transB=True)
            dxs = [dxb[g] for g in range(G)]
        if not give:   # accumulated in place into slots autograd already holds (second micro-batch / second use)
            return (None, None, None, None, *dxs, *([None] * (4 * G)))
        return (None, None, None, None, *dxs, *dWs, *dbl, *dgs, *dbs)


def linear_ln_group(xs, Ws, bs, gammas, betas, *, ct: int, eps: float = 1e-5):
    """[LayerNorm_g(x_g @ W_g^T + b_g)] for G same-shape encoders in grouped launches."""
    G = len(xs)
    need_dx = any(x.requires_grad for x in xs)
    return _LinearLNGroup.apply(ct, float(eps), G, need_dx, *xs, *Ws, *bs, *gammas, *betas)


# ---- gradient arena offered for one backward pass ----------------------------------------------------------------------
# The fused decoder's backward zero-fills the owner's flat gradient buffers with its one zero launch and then OFFERS the
# slots of the parameters outside the decoder (the input encoders) to the backward functions that run after it in the same
# pass: they accumulate straight into the slots (no zero-fill launch of their own, no pack copy afterwards) and hand
# autograd views of them.  The offer ends with the pass (engine callback), so a later backward that does not start with
# the decoder never sees stale zeroing.  mode "fresh": slots were just zeroed, return the views (autograd adopts them);
# "accumulate": .grad already aliases the slots (second micro-batch), add in place and return None.
# Process-global state (the autograd engine runs the backward functions on its own thread, so thread-local would not reach
# them): ONE backward pass at a time may use an arena -- two models stepping concurrently from different threads must not
# both be given one.
class _ArenaState:
    """The gradient arena of the backward pass in flight, as ONE object with three transitions -- begin() (a whole-pass
    grad_arena context or the decoder's offer), end() (the pass is over: everything per-pass is dropped) and the accessors
    below -- instead of loose class attributes that every call site had to clear by hand."""

    def __init__(self):
        self.zeroed_ptrs = set()   # flat buffers the last fresh pass zero-filled (read by the gradient pack that follows it)
        self.end()

    def begin(self, mode, by_ptr, whole_pass, pending=None):
        self.mode = mode                 # None | "fresh" | "accumulate"
        self.by_ptr = by_ptr             # parameter data_ptr -> (parameter, flat buffer, element offset, numel) of its slot
        self.written = set()             # slots some function of this pass already returned (a second use adds in place)
        self.whole_pass = whole_pass     # offered by grad_arena() around the whole backward (the decoder then neither zeroes nor offers)
        self.pending = pending           # fresh whole-pass arena: buffers still to be zeroed -- by the FIRST consumer

    def end(self, keep_zeroed: bool = True):
        self.mode, self.by_ptr, self.written, self.whole_pass, self.pending = None, {}, set(), False, None
        self.multi = set()               # slots that took an in-place second use in a FRESH pass: verified when the pass ends
        if not keep_zeroed:
            self.zeroed_ptrs = set()


_Arena = _ArenaState()


def arena_flush_zero(extra=()) -> bool:
    """Zero the whole-pass arena's buffers if that is still pending (one launch, ``extra`` buffers of the caller included);
    False when there was nothing pending (the caller zeroes its own buffers itself)."""
    if _Arena.pending is None:
        return False
    bufs, _Arena.pending = _Arena.pending, None
    zero_many(list(bufs) + [t for t in extra if t is not None])
    _Arena.zeroed_ptrs = {b.data_ptr() for b in bufs}
    return True


def arena_zeroed_buffers(consume: bool = True) -> set:
    """data_ptrs of the flat buffers the last fresh whole-pass arena zero-filled: the gradient pack right after that pass need
    not zero the slots of parameters that received no gradient.  consume: forget them (one pack per pass)."""
    z = _Arena.zeroed_ptrs
    if consume:
        _Arena.zeroed_ptrs = set()
    return z


def arena_verify(returned=None) -> None:
    """A fresh pass hands the FIRST arena-aware use of a parameter a view of its slot and lets later arena-aware uses (a tied
    weight) add into the slot in place.  That is only correct while autograd keeps that view as the gradient: a gradient
    for the same parameter from a function that is NOT arena-aware makes autograd sum out of place, .grad becomes a tensor
    of its own and the later in-place additions are lost.  Autograd gives no guarantee here, so the hand-out is CHECKED
    when the pass ends: every slot that took an in-place second use must still be what .grad aliases.  ``returned``:
    {parameter data_ptr: gradient torch.autograd.grad() returned} for callers that do not go through .backward()
    (GraphedQuery3D): checked instead of .grad."""
    bad = []
    for q in _Arena.multi:
        ent = _Arena.by_ptr.get(q)
        if ent is None:
            continue
        p_, fl_, o_, _n = ent
        g_ = returned.get(q, None) if returned is not None else p_.grad
        if returned is None and g_ is None:
            continue      # torch.autograd.grad(): .grad is not written -- the caller verifies what it got back (returned=...)
        if g_ is None or g_.data_ptr() != fl_.data_ptr() + 4 * o_:
            bad.append(tuple(p_.shape))
    _Arena.multi = set()
    if bad:
        raise RuntimeError(f"gradient arena: {len(bad)} parameter(s) (shapes {bad[:4]}) received gradients both in place "
                           "through their arena slot (a tied weight's second use) and through a function that does not use "
                           "the arena; autograd summed them out of place and the in-place part is lost -- run this backward "
                           "without ops.grad_arena / enc.grad_arena, or route every use of the parameter through pq3d_amd.ops")


class grad_arena:
    """``with ops.grad_arena(slots, buffers): loss.backward()`` -- the owner of the flat gradient buffers (a
    FlatGradAllReducer / TrainStep) offers every parameter's slot for the WHOLE backward pass: one zero launch up front
    (none when every .grad still aliases its slot: accumulation over micro-batches), then every arena-aware backward function
    -- the heads that run BEFORE the decoder's backward (the caption body), the decoder, the input encoders after it --
    accumulates in place.  Without this context the decoder's backward makes the offer itself (arena_offer), which only
    reaches the functions that run after it.

    .grad IS NOT FINAL UNTIL THE CONTEXT EXITS: weight gradients of arena-aware linear layers are queued (_DwDeferred) and
    launched grouped at __exit__ (earlier only at the fused decoder's readiness reports, FlatGradAllReducer.launch() /
    pack(), or when the queue exceeds _DW_DEFER_CAP); a reader of p.grad inside the pass sees zeros or partial sums.
    Parameters with gradient hooks registered are excluded from the deferral for that reason."""

    def __init__(self, slots, buffers, pack_follows: bool = False):
        # pack_follows: the owner calls FlatGradAllReducer.pack() on these buffers right after the pass; only then is the
        # "already zero-filled" note (arena_zeroed_buffers) kept past the end of the context
        self.slots, self.buffers, self.pack_follows = slots, list(buffers), pack_follows

    def verify_returned(self, params, grads) -> None:
        """For torch.autograd.grad() callers, after the context: the gradients returned for tied parameters that took an
        in-place second use must alias their slots."""
        _Arena.multi, _Arena.by_ptr = set(self.multi), dict(self.by_ptr)
        try:
            arena_verify({p.data_ptr(): g for p, g in zip(params, grads)})
        finally:
            _Arena.end()

    def __enter__(self):
        _Arena.zeroed_ptrs, _Arena.multi = set(), set()
        zeroed = {b.data_ptr() for b in self.buffers}
        ents = {}
        for i_, q_ in getattr(self.slots, "params", {}).items():
            fl_, o_, n_ = self.slots[i_]
            if q_.requires_grad and fl_.data_ptr() in zeroed:
                ents[q_.data_ptr()] = (q_, fl_, o_, n_)
        alias = [q_.grad is not None and q_.grad.data_ptr() == f_.data_ptr() + 4 * o_ for q_, f_, o_, n_ in ents.values()]
        # a later micro-batch of an accumulating step: the in-place gradients of the previous one are still adopted as .grad
        # (functions only add into slots whose .grad aliases them; gradients that live in tensors of their own keep
        # accumulating through autograd and are packed afterwards).  No aliasing .grad anywhere: a fresh step, one zero launch
        mode = "accumulate" if any(alias) else "fresh"
        # fresh: the buffers are zeroed by the first consumer of the pass (with its own scratch: one launch)
        _Arena.begin(mode, ents, True, pending=list(self.buffers) if mode == "fresh" else None)
        return self

    def __exit__(self, *exc):
        self.multi, self.by_ptr = set(_Arena.multi), dict(_Arena.by_ptr)
        try:
            dw_deferred_flush(run=exc[0] is None)   # the queued weight-gradient products of the pass: a few grouped launches
            if exc[0] is None:
                arena_flush_zero()    # nobody consumed it: the owner still expects zeroed buffers
                arena_verify()
        finally:
            _Arena.end(keep_zeroed=self.pack_follows)
        return False


def arena_offer(views_by_ptr, mode):
    _Arena.begin(mode, views_by_ptr, False)

    def _end():
        if not _Arena.whole_pass:
            try:
                arena_verify()
            finally:
                _Arena.end()
    torch.autograd.Variable._execution_engine.queue_callback(_end)


def arena_take(ptrs, numels=None):
    """Slot views for the parameters at ``ptrs`` (all of them or None) and whether to return them to autograd.  ``numels``:
    the element counts the caller is about to write -- a tensor that only STARTS where a parameter starts (the q rows of a
    stacked in_proj_weight passed as a slice) is not that parameter."""
    if _Arena.mode is None or any(q not in _Arena.by_ptr for q in ptrs):
        return None, False
    if numels is not None and any(_Arena.by_ptr[q][3] != n for q, n in zip(ptrs, numels)):
        return None, False
    arena_flush_zero()
    # fresh view objects per call: AccumulateGrad adopts a gradient without a copy only when nobody else references it
    ent = [(p, fl[o:o + n].view(p.shape)) for p, fl, o, n in (_Arena.by_ptr[q] for q in ptrs)]
    if _Arena.mode == "fresh":
        seen = [q in _Arena.written for q in ptrs]
        if any(seen) and not all(seen):
            return None, False
        give = not any(seen)
        if not give:
            _Arena.multi.update(ptrs)
    else:   # accumulate: only slots autograd already holds as .grad (anything else still has last step's data in it)
        if not all(p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in ent):
            return None, False
        give = False
    _Arena.written.update(ptrs)
    return [v for _, v in ent], give


class _SplitRows(Function):
    """(y[:n], y[n:]) of a row matrix whose backward is ONE concatenation -- autograd's two slice backwards would each
    zero-fill a full-size buffer, copy their part in, and then add the two (5 launches)."""

    @staticmethod
    def forward(ctx, y, n):
        ctx.n, ctx.shape = n, tuple(y.shape)
        return y[:n], y[n:]

    @staticmethod
    def backward(ctx, da, db):
        n, (rows, d_) = ctx.n, ctx.shape
        if da is None and db is None:
            return None, None
        if (da is not None and db is not None and da.is_contiguous() and db.is_contiguous() and da.dtype == db.dtype
                and da.numel() == n * d_ and db.numel() == (rows - n) * d_
                and da.untyped_storage().data_ptr() == db.untyped_storage().data_ptr()
                and da.storage_offset() + da.numel() == db.storage_offset()):
            # the producer wrote the two gradients side by side in one buffer (ops.sum_pair): already concatenated
            return torch.as_strided(da, (rows, d_), (d_, 1), da.storage_offset()), None
        da = da.reshape(n, d_) if da is not None else db.new_zeros(n, d_)
        db = db.reshape(rows - n, d_) if db is not None else da.new_zeros(rows - n, d_)
        return torch.cat([da, db], 0), None


def split_rows(y, n: int):
    return _SplitRows.apply(y, int(n))


# ------------------------------------------------------------------------------------------------ T5 body pieces
class _RMSNorm(Function):
    """res=True: returns (y, x) -- the second output is x itself, for the sublayer's residual add: the gradients of BOTH uses
    of x then arrive in this backward and are summed inside its kernel (no add launch at the junction).
    grad_drop = (drop site, slot dict): the backward kernel also writes dropout_mask(site) * dx / (1 - p) into slot["g"]
    (slot["of"] = dx) for the projection that produced x (ops.linear(..., drop=site, masked_grad=slot))."""

    @staticmethod
    def forward(ctx, x, w, eps, res=False, grad_drop=None):
        ctx.res, ctx.grad_drop = bool(res), grad_drop
        x, w = _c(x).float(), _c(w).float()
        d_ = x.shape[-1]
        R = x.numel() // d_
        y = torch.empty_like(x)
        rstd = _empty(R, dtype=torch.float32, device=x.device)
        L.check(L.lib().pq3d_rmsnorm_fwd(L.ptr(x), L.ptr(w), L.ptr(y), L.ptr(rstd), R, d_, eps, L.stream()), "pq3d_rmsnorm_fwd")
        ctx.save_for_backward(x, w, rstd)
        ctx.pptr = w.data_ptr()
        return (y, x.view_as(x)) if res else y

    @staticmethod
    def backward(ctx, dy, dres=None):
        x, w, rstd = ctx.saved_tensors
        d_ = x.shape[-1]
        if dy is None:   # only the pass-through was used
            return (dres, None, None, None, None)
        dy = dy.contiguous().float()
        dres = dres.contiguous().float() if dres is not None else None
        dx = torch.empty_like(x)
        slot, give = arena_take([ctx.pptr], [w.numel()])   # the owner's gradient arena, when offered for this pass (pre-zeroed)
        dw = slot[0] if slot is not None else torch.empty_like(w)
        gd = ctx.grad_drop
        dxm, dc = None, None
        if gd is not None and gd[0] is not None and gd[0].p > 0.0:
            dxm, dc = torch.empty_like(x), gd[0].c()
        L.check(L.lib().pq3d_rmsnorm_bwd_res_drop(L.ptr(x), L.ptr(w), L.ptr(rstd), L.ptr(dy), L.ptr(dres), L.ptr(dx), L.ptr(dw),
                                                  x.numel() // d_, d_, 1 if slot is not None else 0,
                                                  C.byref(dc) if dc is not None else None, L.ptr(dxm), L.stream()),
                "pq3d_rmsnorm_bwd_res_drop")
        if dxm is not None:
            gd[1]["g"], gd[1]["of"] = dxm, dx
        return dx, (dw if (slot is None or give) else None), None, None, None


def rmsnorm(x, w, eps: float = 1e-6, grad_drop=None):
    """T5LayerNorm: x * rsqrt(mean(x^2) + eps) * w (fp32)."""
    return _RMSNorm.apply(x, w, float(eps), False, grad_drop)


def rmsnorm_res(x, w, eps: float = 1e-6, grad_drop=None):
    """(rmsnorm(x), x'): x' is x for the residual add of the pre-norm sublayer -- its gradient joins the norm's inside the
    norm's backward kernel."""
    return _RMSNorm.apply(x, w, float(eps), True, grad_drop)


class _Embedding(Function):
    @staticmethod
    def forward(ctx, table, ids, drop=None):
        table, ids = _c(table).float(), _c(ids).long()
        d_ = table.shape[1]
        out = _empty(*ids.shape, d_, dtype=torch.float32, device=table.device)
        if drop is not None:
            dc = drop.c()
            L.check(L.lib().pq3d_embedding_drop_fwd(L.ptr(table), L.ptr(ids), L.ptr(out), ids.numel(), d_, C.byref(dc), L.stream()),
                    "pq3d_embedding_drop_fwd")
        else:
            L.check(L.lib().pq3d_embedding_fwd(L.ptr(table), L.ptr(ids), L.ptr(out), ids.numel(), d_, L.stream()),
                    "pq3d_embedding_fwd")
        ctx.save_for_backward(ids)
        ctx.shape, ctx.drop = table.shape, drop
        ctx.pptr = table.data_ptr()
        return out

    @staticmethod
    def backward(ctx, dout):
        (ids,) = ctx.saved_tensors
        slot, give = arena_take([ctx.pptr], [ctx.shape[0] * ctx.shape[1]])
        dt = slot[0].view(ctx.shape) if slot is not None else torch.zeros(ctx.shape, dtype=torch.float32, device=dout.device)
        dout = dout.contiguous().float()
        if ctx.drop is not None:
            dc = ctx.drop.c()
            L.check(L.lib().pq3d_embedding_drop_bwd_acc(L.ptr(dout), L.ptr(ids), L.ptr(dt), ids.numel(), ctx.shape[1], C.byref(dc),
                                                        L.stream()), "pq3d_embedding_drop_bwd_acc")
        else:
            L.check(L.lib().pq3d_embedding_bwd_acc(L.ptr(dout), L.ptr(ids), L.ptr(dt), ids.numel(), ctx.shape[1], L.stream()),
                    "pq3d_embedding_bwd_acc")
        return (dt if (slot is None or give) else None), None, None


class _T5Prep(Function):
    """decoder_input_ids, the [B, H, T, T] self-attention bias (relative-position bias + causal -inf) and the encoder tokens'
    key-padding bytes in ONE launch (pq3d_t5_prep); backward: the bias table's gradient in one more (pq3d_t5_bias_bwd)."""

    @staticmethod
    def forward(ctx, rel, labels, buckets, enc_valid, start_id, pad_id, H):
        ctx.set_materialize_grads(False)    # no zero-filled gradients for the integer / boolean outputs
        rel, labels, buckets = _c(rel).float(), _c(labels).long(), _c(buckets).long()
        B, T = labels.shape
        dev = rel.device
        ids = _empty(B, T, dtype=torch.int64, device=dev)
        bias = _empty(B, H, T, T, dtype=torch.float32, device=dev)
        kpm, ev, N = None, None, 0
        if enc_valid is not None:
            ev = _c(enc_valid)
            assert ev.dtype == torch.bool and ev.shape[0] == B
            N = ev.shape[1]
            kpm = _empty(B, N, dtype=torch.bool, device=dev)
        L.check(L.lib().pq3d_t5_prep(L.ptr(labels), int(start_id), int(pad_id), L.ptr(rel), L.ptr(buckets), L.ptr(ev), L.ptr(ids),
                                     L.ptr(bias), L.ptr(kpm), B, T, H, N, L.stream()), "pq3d_t5_prep")
        ctx.save_for_backward(buckets)
        ctx.cfg = (B, T, H, rel.shape[0])
        ctx.pptr = rel.data_ptr()
        ctx.mark_non_differentiable(ids)
        if kpm is not None:
            ctx.mark_non_differentiable(kpm)
            return ids, bias, kpm
        return ids, bias

    @staticmethod
    def backward(ctx, _dids, dbias, _dkpm=None):
        (buckets,) = ctx.saved_tensors
        B, T, H, NB = ctx.cfg
        if dbias is None:
            return (None,) * 7
        slot, give = arena_take([ctx.pptr], [NB * H])
        drel = slot[0].view(NB, H) if slot is not None else _empty(NB, H, dtype=torch.float32, device=dbias.device)
        dbias = dbias.contiguous().float()
        L.check(L.lib().pq3d_t5_bias_bwd(L.ptr(dbias), L.ptr(buckets), L.ptr(drel), B, T, H, NB, 1 if slot is not None else 0,
                                         L.stream()), "pq3d_t5_bias_bwd")
        return (drel if (slot is None or give) else None), None, None, None, None, None, None


def t5_prep(rel, labels, buckets, enc_valid, start_id: int, pad_id: int, H: int):
    """(decoder_input_ids [B,T], self-attention bias [B,H,T,T], encoder key-padding bytes [B,N] or None): see _T5Prep."""
    out = _T5Prep.apply(rel, labels, buckets, enc_valid, int(start_id), int(pad_id), int(H))
    return (out[0], out[1], out[2]) if enc_valid is not None else (out[0], out[1], None)


class _Fanout(Function):
    """n aliases of one tensor whose gradients are summed by ONE launch (pq3d_sum_n, fixed order) instead of autograd's n - 1
    pairwise adds: a tensor read by every layer of a stack (T5's shared self-attention position bias)."""

    @staticmethod
    def forward(ctx, x, n):
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g for g in gs if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1 or any(g.dtype != torch.float32 or g.shape != gs[0].shape for g in gs):
            out = gs[0]
            for g in gs[1:]:
                out = out + g
            return out, None
        return sum_n(gs), None


def fanout(x: torch.Tensor, n: int):
    """[x] * n for a tensor that n consumers read; with a gradient, their n gradients are summed in one launch."""
    if n <= 1 or not (torch.is_grad_enabled() and x.requires_grad):
        return [x] * n
    return list(_Fanout.apply(x, n))


def embedding(table, ids, drop: Optional[L.Drop] = None):
    """table[ids] (nn.Embedding), optionally with the dropout of site ``drop`` over the [ids.numel(), d] rows fused in."""
    return _Embedding.apply(table, ids, drop if (drop is not None and drop.p > 0.0) else None)


# ---------------------------------------------------------------------------------------------------------------------
# Row-local chain of a decoder layer's forward in one launch (csrc/chain_ffn.hip): self-attention out-projection + post-norm
# + FFN + post-norm.  Same bits as the five launches it replaces (tests/test_gpu_chain.py).
def chain_flags(R: int, device) -> torch.Tensor:
    """Hand-off words of one call site (zeroed once; the kernel keeps them consistent from launch to launch)."""
    return torch.zeros(max((R + 31) // 32, 64) * 8 * 16, dtype=torch.int32, device=device)


_CHAIN_ERR = {}
_CHAIN_WS = {}


def _chain_ws(dev) -> torch.Tensor:
    """Scratch of the backward chains (the groups' LayerNorm parameter-gradient partials): written and read inside one launch."""
    w = _CHAIN_WS.get(dev)
    if w is None:
        w = _CHAIN_WS[dev] = torch.empty(32 * 8 * 1536, dtype=torch.float32, device=dev)
    return w


def chain_ffn_ok(R: int, d: int, F_: int) -> bool:
    return d == 256 and F_ == 2048 and 1 <= R <= 2048


def chain_sa_ok(Nq: int, H: int, d: int) -> bool:
    """The self-attention core can run as step 0 of chain_ffn_fwd (csrc/chain_ffn.hip): 8 heads of 32, <= 240 queries per scene."""
    return d == 256 and H == 8 and 1 <= Nq <= 240 and os.environ.get("PQ3D_CHAIN_SA", "1") != "0"


def chain_ffn_fwd(o_s, Wo, bo, x1s, g1, be1, eps1, W1, b1, W2, b2, g2, be2, eps2, flags, nextq=None, q_dtype=torch.bfloat16, sa=None):
    """Returns (f, x2, mean1, rstd1, h, zp, z, x3, mean2, rstd2[, q_next]); every tensor fp32, rows = o_s.numel() // 256.
    nextq = (qpos, [Wq_m], [bq_m]): also q_next[m] = (x3 + qpos) Wq_m^T + bq_m as bf16 (the next layer's cross-attention queries;
    q_dtype = torch.float32 for compute mode 'bf16x3').
    sa = (q, k, v [B, Nq, 256] fp32, bias [B, 8, Nq, Nq] or None, kpm [B, Nq] bool or None, lse [B, 8, Nq], scale): the launch first
    forms o_s (an OUTPUT then) and lse = the split-bf16 self-attention core of pq3d_attn_fwd, bit for bit."""
    d = o_s.shape[-1]
    R, F_ = o_s.numel() // d, W1.shape[0]
    dev = o_s.device
    e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    f, x2, x3, z = e(o_s.shape), e(o_s.shape), e(o_s.shape), e(o_s.shape)
    h, zp = e(*o_s.shape[:-1], F_), e(4, *o_s.shape)
    mean1, rstd1, mean2, rstd2 = e(1, R), e(1, R), e(1, R), e(1, R)
    err = _CHAIN_ERR.get(dev)
    if err is None:
        err = _CHAIN_ERR[dev] = torch.zeros(1, dtype=torch.int32, device=dev)
    c = L.ChainFfnDesc()
    c.R, c.d, c.F, c.eps1, c.eps2 = R, d, F_, eps1, eps2
    for n, t in (("o_s", o_s), ("Wo", Wo), ("bo", bo), ("x1s", x1s), ("g1", g1), ("be1", be1), ("f", f), ("x2", x2), ("mean1", mean1),
                 ("rstd1", rstd1), ("W1", W1), ("b1", b1), ("h", h), ("W2", W2), ("b2", b2), ("zp", zp), ("z", z), ("g2", g2),
                 ("be2", be2), ("x3", x3), ("mean2", mean2), ("rstd2", rstd2), ("flags", flags), ("err", err)):
        assert t.is_contiguous() and (n in ("flags", "err") or t.dtype == torch.float32), n
        setattr(c, n, L.ptr(t))
    if sa is not None:
        q_, k_, v_, bias_, kpm_, lse_, scale_ = sa
        Nq = q_.shape[-2]
        assert all(t.is_contiguous() and t.dtype == torch.float32 and t.shape == o_s.shape for t in (q_, k_, v_)) and R % Nq == 0
        assert lse_.is_contiguous() and lse_.dtype == torch.float32 and lse_.numel() == (R // Nq) * 8 * Nq
        c.sa_q, c.sa_k, c.sa_v, c.sa_lse, c.sa_nq, c.sa_scale = L.ptr(q_), L.ptr(k_), L.ptr(v_), L.ptr(lse_), Nq, float(scale_)
        if bias_ is not None:
            assert bias_.is_contiguous() and bias_.dtype == torch.float32 and bias_.numel() == (R // Nq) * 8 * Nq * Nq
            c.sa_bias = L.ptr(bias_)
        if kpm_ is not None:
            assert kpm_.is_contiguous() and kpm_.dtype in (torch.bool, torch.uint8) and kpm_.numel() == R
            c.sa_kpm = L.ptr(kpm_)
            c._kpm_keepalive = kpm_
    qn = None
    if nextq is not None:
        qpos, Wqs, bqs = nextq
        qn = torch.empty(len(Wqs), *o_s.shape, dtype=q_dtype, device=dev)
        assert qpos.is_contiguous() and qpos.dtype == torch.float32 and len(Wqs) <= 3
        c.nq, c.qpos, c.qout_f32 = len(Wqs), L.ptr(qpos), int(q_dtype == torch.float32)
        for m, (w_, b_) in enumerate(zip(Wqs, bqs)):
            assert w_.is_contiguous() and b_.is_contiguous() and w_.dtype == torch.float32
            c.Wq[m], c.bq[m], c.qout[m] = L.ptr(w_), L.ptr(b_), L.ptr(qn[m])
    from .profiler import timed
    fl = 2.0 * R * d * (d + 2 * F_)
    nb = 4.0 * (R * d * 9 + 2 * R * F_ + d * d + 2 * d * F_)
    if sa is not None:   # the self-attention core of step 0: scores + value contraction, q / k / v in, bias in
        fl += 4.0 * R * sa[0].shape[-2] * d
        nb += 4.0 * (3 * R * d + (sa[3].numel() if sa[3] is not None else 0))
    L.check(timed("pq3d_chain_ffn_fwd", f"R{R}d{d}F{F_}", fl, nb, L.lib().pq3d_chain_ffn_fwd, C.byref(c), L.stream()), "pq3d_chain_ffn_fwd")
    return (f, x2, mean1, rstd1, h, zp, z, x3, mean2, rstd2) + ((qn,) if qn is not None else ())


def chain_ca_ok(R: int, d: int, M: int) -> bool:
    return d == 256 and 1 <= M <= 3 and 1 <= R <= 2048


def chain_ca_fwd(o_all, Wos, bos, x, gammas, betas, eps, coef, rows_per_scene, qpos, Wqkv, bqkv, flags):
    """o_all [M, ..., d] bf16 (fp32: split-bf16 out-projections, compute mode 'bf16x3') -> (op_all [M, ..., d], x1, mean [M, R],
    rstd [M, R], qkv [3, ..., d]); everything else fp32."""
    M, d = o_all.shape[0], o_all.shape[-1]
    R = x.numel() // d
    dev = x.device
    e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    op_all, x1, qkv = e(M, *x.shape), e(x.shape), e(3, *x.shape)
    mean, rstd = e(M, R), e(M, R)
    err = _CHAIN_ERR.get(dev)
    if err is None:
        err = _CHAIN_ERR[dev] = torch.zeros(1, dtype=torch.int32, device=dev)
    c = L.ChainCaDesc()
    c.R, c.d, c.M, c.rows_per_scene, c.eps = R, d, M, rows_per_scene, eps
    assert o_all.dtype in (torch.bfloat16, torch.float32) and o_all.is_contiguous() and x.is_contiguous() and qpos.is_contiguous()
    c.o_f32 = int(o_all.dtype == torch.float32)
    assert x.dtype == torch.float32 and qpos.dtype == torch.float32
    for m in range(M):
        for n, t in (("o", o_all[m]), ("Wo", Wos[m]), ("bo", bos[m]), ("gamma", gammas[m]), ("beta", betas[m]), ("op", op_all[m])):
            assert t.is_contiguous() and (n == "o" or t.dtype == torch.float32), n
            getattr(c, n)[m] = L.ptr(t)
    for g in range(3):
        for n, t in (("Wqkv", Wqkv[g]), ("bqkv", bqkv[g]), ("qkv", qkv[g])):
            assert t.is_contiguous() and t.dtype == torch.float32, n
            getattr(c, n)[g] = L.ptr(t)
    if coef is not None:
        assert coef.is_contiguous() and coef.dtype == torch.float32
    c.x, c.coef, c.x1, c.mean, c.rstd, c.qpos, c.flags, c.err = map(L.ptr, (x, coef, x1, mean, rstd, qpos, flags, err))
    fl = 2.0 * R * d * d * (M + 3)
    nb = 4.0 * (R * d * (2 * M + 6) + (M + 3) * d * d) + 2.0 * M * R * d
    L.check(timed("pq3d_chain_ca_fwd", f"R{R}d{d}M{M}", fl, nb, L.lib().pq3d_chain_ca_fwd, C.byref(c), L.stream()), "pq3d_chain_ca_fwd")
    return op_all, x1, mean, rstd, qkv


def chain_mh_ok(d: int, hidden: int, C_: int, Mm: int, R: int) -> bool:
    return d == 256 and hidden == 256 and 1 <= C_ <= 256 and 0 <= Mm <= 3 and 1 <= R <= 2048


def chain_mh_fwd(x, W0, b0, gamma, beta, eps, W4, b4, colfill, fill, Wqs, bqs, flags):
    """Row-local part of a mask-head call in one launch: returns (h1, h2, mean [1, R], rstd [1, R], cls [..., C], qm [Mm, ..., d]);
    everything fp32.  colfill: int32 [C] column flags (non-zero -> `fill`) or None."""
    d = x.shape[-1]
    R, C_, Mm = x.numel() // d, W4.shape[0], len(Wqs)
    dev = x.device
    e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    h1, h2, cls, qm = e(x.shape), e(x.shape), e(*x.shape[:-1], C_), e(Mm, *x.shape)
    mean, rstd = e(1, R), e(1, R)
    err = _CHAIN_ERR.get(dev)
    if err is None:
        err = _CHAIN_ERR[dev] = torch.zeros(1, dtype=torch.int32, device=dev)
    c = L.ChainMhDesc()
    c.R, c.d, c.Mm, c.C, c.eps, c.fill = R, d, Mm, C_, eps, fill
    for n, t in (("x", x), ("W0", W0), ("b0", b0), ("gamma", gamma), ("beta", beta), ("W4", W4), ("b4", b4)):
        assert t.is_contiguous() and t.dtype == torch.float32, n
    assert colfill is None or (colfill.dtype == torch.int32 and colfill.numel() == C_ and colfill.is_contiguous())
    for m in range(Mm):
        for n, t in (("Wq", Wqs[m]), ("bq", bqs[m]), ("qm", qm[m])):
            assert t.is_contiguous() and t.dtype == torch.float32, n
            getattr(c, n)[m] = L.ptr(t)
    c.x, c.W0, c.b0, c.gamma, c.beta, c.W4, c.b4, c.colfill = map(L.ptr, (x, W0, b0, gamma, beta, W4, b4, colfill))
    c.h1, c.h2, c.mean, c.rstd, c.cls, c.flags, c.err = map(L.ptr, (h1, h2, mean, rstd, cls, flags, err))
    fl = 2.0 * R * d * (d * (1 + Mm) + C_)
    nb = 4.0 * (R * d * (3 + Mm) + R * C_ + (1 + Mm) * d * d + C_ * d)
    L.check(timed("pq3d_chain_mh_fwd", f"R{R}d{d}M{Mm}C{C_}", fl, nb, L.lib().pq3d_chain_mh_fwd, C.byref(c), L.stream()), "pq3d_chain_mh_fwd")
    return h1, h2, mean, rstd, cls, qm


def chain_mh_bwd(dc, colfill, W4, h1, mean, rstd, gamma, dgamma, dbeta, W0, cur, dqs, Wqs, flags, dcl_out=None, prev=None):
    """Backward of chain_mh_fwd's part in one launch.  dc [..., C] fp32; dqs: the Mm query-side gradients of the mask logits
    ([..., d] each, all fp32 or all bf16).  Returns (dcl, dpre bf16, out): dcl = dc with the flagged columns zeroed (dc itself
    without flags), dpre = d(linear 0 output), out = sum_m dq_m Wq_m + (dpre W0 + cur).  dgamma / dbeta are accumulated onto.
    prev = (dq_all [M, ..., d] bf16, [Wq_m], dxr, gq): cur is None and cur = sum_m dq_all_m Wq_m + dxr is formed in the launch (gq
    receives the sum without dxr)."""
    d = h1.shape[-1]
    R, C_, Mm = h1.numel() // d, W4.shape[0], len(dqs)
    dev = h1.device
    dcl = (dcl_out if dcl_out is not None else torch.empty_like(dc)) if colfill is not None else None
    assert dcl is None or (dcl.is_contiguous() and dcl.dtype == torch.float32 and dcl.shape == dc.shape)
    dh2, out = torch.empty_like(h1), torch.empty_like(h1)
    dpre = torch.empty(h1.shape, dtype=torch.bfloat16, device=dev)
    err = _CHAIN_ERR.get(dev)
    if err is None:
        err = _CHAIN_ERR[dev] = torch.zeros(1, dtype=torch.int32, device=dev)
    c = L.ChainMhBwdDesc()
    c.R, c.d, c.Mm, c.C = R, d, Mm, C_
    c.dq_f32 = int(Mm > 0 and dqs[0].dtype == torch.float32)
    if prev is not None:
        dq_all, Wqc, dxr, gq = prev
        assert cur is None and dq_all.dtype == torch.bfloat16 and dq_all.is_contiguous() and 1 <= dq_all.shape[0] <= 3
        c.nq = dq_all.shape[0]
        for m in range(c.nq):
            assert Wqc[m].is_contiguous() and Wqc[m].dtype == torch.float32
            c.dqc[m], c.Wqc[m] = L.ptr(dq_all[m]), L.ptr(Wqc[m])
        cur = dxr
        c.dxr, c.gq = L.ptr(dxr), L.ptr(gq)
        assert gq.is_contiguous() and gq.dtype == torch.float32
    for n, t in (("dc", dc), ("W4", W4), ("h1", h1), ("mean", mean), ("rstd", rstd), ("gamma", gamma), ("dgamma", dgamma),
                 ("dbeta", dbeta), ("W0", W0), ("cur", cur)):
        assert t.is_contiguous() and t.dtype == torch.float32, n
    assert colfill is None or (colfill.dtype == torch.int32 and colfill.numel() == C_ and colfill.is_contiguous())
    for m in range(Mm):
        assert dqs[m].is_contiguous() and dqs[m].dtype == dqs[0].dtype and dqs[m].dtype in (torch.float32, torch.bfloat16)
        assert Wqs[m].is_contiguous() and Wqs[m].dtype == torch.float32
        c.dq[m], c.Wq[m] = L.ptr(dqs[m]), L.ptr(Wqs[m])
    c.dc, c.colfill, c.dcl, c.W4, c.h1, c.mean, c.rstd, c.gamma, c.dgamma, c.dbeta = map(
        L.ptr, (dc, colfill, dcl, W4, h1, mean, rstd, gamma, dgamma, dbeta))
    c.dh2, c.dpre, c.W0, c.cur, c.out, c.flags, c.err, c.lnws = map(
        L.ptr, (dh2, dpre, W0, cur if prev is None else None, out, flags, err, _chain_ws(dev)))
    fl = 2.0 * R * d * (C_ + d * (1 + Mm))
    nb = 4.0 * (R * C_ * 2 + R * d * (5 + Mm) + C_ * d + (1 + Mm) * d * d) + 2.0 * R * d
    L.check(timed("pq3d_chain_mh_bwd", f"R{R}d{d}M{Mm}C{C_}", fl, nb, L.lib().pq3d_chain_mh_bwd, C.byref(c), L.stream()), "pq3d_chain_mh_bwd")
    return (dcl if dcl is not None else dc), dpre, out


def chain_ffn_bwd(dx, x2, z, g2, mean2, rstd2, dg2, db2, W2, h, W1, x1s, f, g1, mean1, rstd1, dg1, db1, flags, prev=None):
    """Backward of the FFN sublayer + the self-attention post-norm in one launch.  Returns (dy, dhp, df): dy = d z (= the
    residual-branch gradient of LN2), dhp = d(linear1 output) as bf16, df = d f (= the residual-branch gradient of LN1).
    dg2 / db2 / dg1 / db1 (arena views) are accumulated onto.
    prev = (dq_all [M, ..., d] bf16, [Wq_m], dxr, gq): dx is None and the upstream gradient sum_m dq_m Wq_m + dxr is formed in the
    launch (gq receives the sum without dxr); a fourth result, that upstream gradient, is returned."""
    like = dx if dx is not None else x2
    d = like.shape[-1]
    R, F_ = like.numel() // d, W1.shape[0]
    dev = like.device
    dy, df = torch.empty_like(like), torch.empty_like(like)
    dhp = torch.empty(*like.shape[:-1], F_, dtype=torch.bfloat16, device=dev)
    dxo = None
    if prev is not None:
        dq_all, Wqs, dxr, gq = prev
        dxo = torch.empty_like(like)
        dx = dxo   # (pointer only: not read by the kernel)
    part = torch.empty(4, R, d, dtype=torch.float32, device=dev)
    err = _CHAIN_ERR.get(dev)
    if err is None:
        err = _CHAIN_ERR[dev] = torch.zeros(1, dtype=torch.int32, device=dev)
    c = L.ChainFfnBwdDesc()
    c.R, c.d, c.F = R, d, F_
    for n, t in (("dx", dx), ("x2", x2), ("z", z), ("g2", g2), ("mean2", mean2), ("rstd2", rstd2), ("dg2", dg2), ("db2", db2), ("dy", dy),
                 ("W2", W2), ("h", h), ("dhp", dhp), ("W1", W1), ("part", part), ("x1s", x1s), ("f", f), ("g1", g1), ("mean1", mean1),
                 ("rstd1", rstd1), ("dg1", dg1), ("db1", db1), ("df", df), ("flags", flags), ("err", err), ("lnws", _chain_ws(dev))):
        assert t.is_contiguous() and (n in ("flags", "err", "dhp") or t.dtype == torch.float32), n
        setattr(c, n, L.ptr(t))
    if prev is not None:
        assert dq_all.dtype == torch.bfloat16 and dq_all.is_contiguous() and dxr.is_contiguous() and gq.is_contiguous() and len(Wqs) <= 3
        c.nq, c.dxr, c.gq, c.dxo = len(Wqs), L.ptr(dxr), L.ptr(gq), L.ptr(dxo)
        for m, w_ in enumerate(Wqs):
            assert w_.is_contiguous() and w_.dtype == torch.float32
            c.dq[m], c.Wq[m] = L.ptr(dq_all[m]), L.ptr(w_)
    fl = 2.0 * R * d * 2 * F_
    nb = 4.0 * (R * d * 12 + R * F_ + 2 * d * F_) + 2.0 * 2 * R * F_
    L.check(timed("pq3d_chain_ffn_bwd", f"R{R}d{d}F{F_}", fl, nb, L.lib().pq3d_chain_ffn_bwd, C.byref(c), L.stream()), "pq3d_chain_ffn_bwd")
    return (dy, dhp, df) if prev is None else (dy, dhp, df, dxo)


def chain_sa_bwd(dqkv, Wl, aux2, x, op_all, gammas, mean, rstd, coef, rows_per_scene, dgammas, dbetas, Wos, flags):
    """dqkv [3, ..., d] fp32 -> (g3 [3, ..., d], dop [M, ..., d], dxr, do_all [M, ..., d] bf16); d gamma / d beta accumulated."""
    d = x.shape[-1]
    R, M = x.numel() // d, op_all.shape[0]
    dev = x.device
    g3 = torch.empty(3, *x.shape, dtype=torch.float32, device=dev)
    dop = torch.empty(M, *x.shape, dtype=torch.float32, device=dev)
    dxr = torch.empty(x.shape, dtype=torch.float32, device=dev)
    do_all = torch.empty(M, *x.shape, dtype=torch.bfloat16, device=dev)
    err = _CHAIN_ERR.get(dev)
    if err is None:
        err = _CHAIN_ERR[dev] = torch.zeros(1, dtype=torch.int32, device=dev)
    c = L.ChainSaBwdDesc()
    c.R, c.d, c.M, c.rows_per_scene = R, d, M, rows_per_scene
    for g in range(3):
        for n, t in (("dqkv", dqkv[g]), ("Wl", Wl[g]), ("g3", g3[g])):
            assert t.is_contiguous() and t.dtype == torch.float32, n
            getattr(c, n)[g] = L.ptr(t)
    for m in range(M):
        for n, t in (("op", op_all[m]), ("gamma", gammas[m]), ("dop", dop[m]), ("dgamma", dgammas[m]), ("dbeta", dbetas[m]), ("Wo", Wos[m]),
                     ("do_all", do_all[m])):
            assert t.is_contiguous() and (n == "do_all" or t.dtype == torch.float32), n
            getattr(c, n)[m] = L.ptr(t)
    for t in (aux2, x, mean, rstd):
        assert t.is_contiguous() and t.dtype == torch.float32
    if coef is not None:
        assert coef.is_contiguous() and coef.dtype == torch.float32
    c.aux2, c.x, c.mean, c.rstd, c.coef, c.dxr, c.flags, c.err = map(L.ptr, (aux2, x, mean, rstd, coef, dxr, flags, err))
    c.lnws = L.ptr(_chain_ws(dev))
    fl = 2.0 * R * d * d * (3 + M)
    nb = 4.0 * (R * d * (8 + 3 * M) + (3 + M) * d * d) + 2.0 * M * R * d
    L.check(timed("pq3d_chain_sa_bwd", f"R{R}d{d}M{M}", fl, nb, L.lib().pq3d_chain_sa_bwd, C.byref(c), L.stream()), "pq3d_chain_sa_bwd")
    return g3, dop, dxr, do_all


def chain_error(device) -> bool:
    """True if a hand-off wait of any chain launch on `device` gave up (synchronises)."""
    err = _CHAIN_ERR.get(torch.device(device) if not isinstance(device, torch.device) else device)
    if err is None:   # keyed by the tensors' own device objects ('cuda:0'): accept an index-less spelling too
        err = next((e for d_, e in _CHAIN_ERR.items() if torch.device(device).type == d_.type and
                    (torch.device(device).index in (None, d_.index))), None)
    return bool(err is not None and int(err.item()) != 0)


class ChainHandoffError(L.Pq3dError):
    """A one-launch chain's in-kernel hand-off gave up (csrc/chain_common.h SPIN_LIMIT): its outputs are NOT valid."""


def chain_check(device=None) -> None:
    """Poll the chains' error word(s) where the host synchronises anyway (end of a training step, after a timed region, a loss
    .item()): a hand-off that ran past its bound means a member workgroup never became resident next to its group (CUs held by
    another stream's kernel, a CU-masked queue) and the launch went on with stale rows.  Raises ChainHandoffError, clears the word
    and turns the chains OFF for the rest of the process (the separate launches compute the same bits) -- never a silent result."""
    bad = []
    for dev, err in _CHAIN_ERR.items():
        if device is not None and torch.device(device).index not in (None, dev.index):
            continue
        if int(err.item()) != 0:
            err.zero_()
            bad.append(str(dev))
    if bad:
        from . import fused
        fused.set_chain(False)
        raise ChainHandoffError(f"a row-local chain launch on {', '.join(bad)} timed out in a hand-off: its results (and every "
                                "result since the last check) are invalid; chains are now disabled in this process "
                                "(fused.set_chain(False)) -- rerun the step")


_CHAIN_DEV_OK = {}


def chain_device_ok(device) -> bool:
    """Device gate of the chains (pq3d_chain_device_ok): gfx950, 256 CUs, 160 KB LDS and the measured round-robin 'XCD == (workgroup id + c) % 8'
    placement.  The measuring launch is skipped while the stream is capturing (property checks only; the probe's verdict is
    cached from the first eager call -- every capture in this package is preceded by eager warm-up steps)."""
    device = torch.device(device)
    key = device.index if device.index is not None else torch.cuda.current_device()
    v = _CHAIN_DEV_OK.get(key)
    if v is not None:
        return v
    capturing = torch.cuda.is_current_stream_capturing()
    with torch.cuda.device(key):
        ok = bool(L.lib().pq3d_chain_device_ok(0 if capturing else 1, L.stream()))
    if not capturing or not ok:
        _CHAIN_DEV_OK[key] = ok
        if not ok:
            import warnings
            warnings.warn(f"pq3d: the one-launch row-local chains are OFF on cuda:{key} (pq3d_chain_device_ok said no: not a 256-CU gfx950 "
                          "in SPX mode with the round-robin workgroup -> XCD placement); the separate launches compute the same bits, ~8 % slower", stacklevel=2)
    return ok
