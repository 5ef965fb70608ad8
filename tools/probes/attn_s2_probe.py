"""Cross-attention backward / forward at the shipped stage-2 shape (128 scenes x 3 memories, 80 queries x 80 keys, 12 heads of
64): two-kernel path vs the all-queries-resident backward (probe builds with -DPQ3D_RES_MIN_LK=64 [-DPQ3D_RES_NQP3])."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pq3d_amd import _lib as L
from pq3d_amd import fused as F
dev = 'cuda'
lib = L.lib()


def timeit(fn, it=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


def case(name, B, Lq, Lk, H, d):
    q = torch.randn(B, Lq, d, device=dev).bfloat16(); k = torch.randn(B, Lk, d, device=dev).bfloat16()
    v = torch.randn(B, Lk, d, device=dev).bfloat16()
    o = torch.empty_like(q); lse = torch.empty(B, H, Lq, device=dev)
    do = torch.randn_like(o); delta = torch.empty_like(lse)
    vl = torch.randint(Lk // 2, Lk + 1, (B,)); vl[0] = Lk
    kw = dict(kpm=(torch.arange(Lk)[None] >= vl[:, None]).to(dev))
    fwd = lambda: F._attn(q, k, v, o, lse, H, L.BF16, True, **kw)
    fwd()
    res, tf = [], []
    for mode in (0, 1, 31):          # general two-kernel path | all-queries-resident backward | small cross-attention kernels
        lib.pq3d_attn_resident(mode)
        tf.append(timeit(fwd))
        dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v)
        bwd = lambda: F._attn(q, k, v, o, lse, H, L.BF16, True, bwd=(do, dq, dk, dv, delta, None), **kw)
        t = timeit(bwd)
        res.append((t, dq.float().clone(), dk.float().clone(), dv.float().clone()))
    lib.pq3d_attn_resident(63)
    errs = [float((res[0][i] - res[2][i]).abs().max() / res[0][i].abs().max()) for i in (1, 2, 3)]
    print(f"{name:34s} fwd general {tf[0]:7.1f} | small {tf[2]:7.1f} us || bwd two-kernel {res[0][0]:7.1f} | resident {res[1][0]:7.1f} | small {res[2][0]:7.1f} us | rel diff dq dk dv {errs[0]:.1e} {errs[1]:.1e} {errs[2]:.1e}", flush=True)


case("s2 cross B384 Lq80 Lk80 H12 dh64", 384, 80, 80, 12, 768)
case("s2 prompt B128 Lq80 Lk32 H12 dh64", 128, 80, 32, 12, 768)
case("c5p prompt B16 Lq100 Lk32 H8 dh32", 16, 100, 32, 8, 256)
case("c5 caption x-attn B16 Lq32 Lk100 H8 dh64", 16, 32, 100, 8, 512)
