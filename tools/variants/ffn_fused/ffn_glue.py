"""Host glue of the one-launch feed-forward variant (was in pq3d_amd/fused.py; needs pq3d_ffn_fwd from ffn.hip and the
FfnDesc ctypes structure below)."""
# the feed-forward sublayer's two products as one launch (csrc/ffn.hip).  Off by default: correct (tests/test_gpu_ops.py) but
# 34 us against 11 + 13 for the two grouped products at config 2 (same-box A/B, step +35 us) -- see the header of ffn.hip
FFN_FUSE = os.environ.get("PQ3D_FFN_FUSE", "0") != "0"


def ffn_fused_ok(cq, d, F_, x, w1, b1, w2, b2) -> bool:
    ts = [x, w1, b1, w2] + ([b2] if b2 is not None else [])
    return (cq == L.BF16X3 and d == 256 and F_ % 256 == 0 and F_ // 256 <= L.MAXG
            and all(t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() % 16 == 0 for t in ts))


def ffn_fwd(x, w1, b1, w2, b2, act, drop, want_pre):
    """pq3d_ffn_fwd: returns (h [.., F], pre or None, zp [F/256, .., d] partial sums of linear2)."""
    d = x.shape[-1]
    R, F_ = x.numel() // d, w1.shape[0]
    h = torch.empty(*x.shape[:-1], F_, dtype=torch.float32, device=x.device)
    pre = torch.empty_like(h) if want_pre else None
    zp = torch.empty(F_ // 256, *x.shape, dtype=torch.float32, device=x.device)
    q = L.FfnDesc()
    q.R, q.d, q.F, q.act = R, d, F_, L.ACT[act]
    q.x, q.w1, q.b1, q.w2, q.b2 = L.ptr(x), L.ptr(w1), L.ptr(b1), L.ptr(w2), L.ptr(b2)
    q.h, q.pre, q.zp = L.ptr(h), L.ptr(pre), L.ptr(zp)
    L.set_drop(q.drop, drop)
    L.check(timed("pq3d_ffn_fwd", f"R{R}d{d}F{F_}", 4.0 * R * d * F_, 4.0 * (2 * d * F_ + R * (2 * d + F_)), L.lib().pq3d_ffn_fwd,
                  C.byref(q), L.stream()), "pq3d_ffn_fwd")
    return h, pre, zp


