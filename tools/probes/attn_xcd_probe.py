"""Cross-attention launches of the BASELINE configs alone (forward + backward through fused._attn, the step's own entry):
  c2: B 24 (3 memories x 8 scenes) Lq 100 Lk 1024, key padding;  c5: B 48 Lq 100 Lk 2048, key padding;
  c4: B 12 Lq 200 Lk 4096, 3-D self-mask + row-open flags.
Event-timed per launch over 4 rotating K / V buffers (as the step's layers read different buffers), then ONE calibration
copy (100 MiB in / out) for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes.  usage: attn_xcd_probe.py [iters] [configs]
A/B: PQ3D_LIB_PATH=pq3d_amd/libpq3d_hip_<variant>.so."""
import sys
sys.path.insert(0, '/root/repo')
import torch
from pq3d_amd import _lib as L
from pq3d_amd import fused as F
from pq3d_amd import ops
dev = 'cuda'
it = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfgs = sys.argv[2].split(',') if len(sys.argv) > 2 else ['c2', 'c4', 'c5']
H, d = 8, 256
NB = 4


def case(name, B, Lq, Lk, mask3=False):
    torch.manual_seed(0)
    q = torch.randn(B, Lq, d, device=dev).bfloat16()
    ks = [torch.randn(B, Lk, d, device=dev).bfloat16() for _ in range(NB)]
    vs = [torch.randn(B, Lk, d, device=dev).bfloat16() for _ in range(NB)]
    o = torch.empty_like(q); lse = torch.empty(B, H, Lq, device=dev)
    do = torch.randn_like(o); dq = torch.empty_like(q); dk = torch.empty_like(ks[0]); dv = torch.empty_like(vs[0])
    delta = torch.empty_like(lse)
    kw = {}
    if mask3:
        nb = B // 3
        m = torch.rand(nb, Lq, Lk, device=dev) < 0.5
        kw["mask"] = m
        import os
        if os.environ.get("PROBE_BITS", "1") == "1":
            kw["row_open"], kw["mask_bits"] = ops.mask_pack(m)
        else:
            kw["row_open"] = ops.mask_row_all(m)
        kw["mask_bmod"] = nb
    else:
        m = torch.zeros(B, Lk, dtype=torch.bool, device=dev)
        lens = torch.randint(Lk // 2, Lk + 1, (B,))
        for b in range(B):
            m[b, int(lens[b]):] = True
        m[0] = False
        kw["kpm"] = m

    def fwd(i):
        F._attn(q, ks[i % NB], vs[i % NB], o, lse, H, L.BF16, True, **kw)

    def bwd(i):
        F._attn(q, ks[i % NB], vs[i % NB], o, lse, H, L.BF16, True, bwd=(do, dq, dk, dv, delta, None), **kw)
    res = {}
    for nm, fn in (("fwd", fwd), ("bwd", bwd)):
        for i in range(3):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(it):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        res[nm] = e0.elapsed_time(e1) / it * 1e3
    kv_mb = 2 * B * Lk * d * 2 / 1e6
    print(f"RESULT {name} B{B} Lq{Lq} Lk{Lk} mask3={mask3}: fwd {res['fwd']:.1f} us  bwd {res['bwd']:.1f} us   (K+V {kv_mb:.1f} MB)")


if 'c2' in cfgs:
    case('c2', 24, 100, 1024)
if 'c5' in cfgs:
    case('c5', 48, 100, 2048)
if 'c4' in cfgs:
    case('c4', 12, 200, 4096, mask3=True)
n_cal = (100 << 20) // 4
csrc, cdst = torch.ones(n_cal, device=dev), torch.empty(n_cal, device=dev)
ops.copy_many([cdst], [csrc])
torch.cuda.synchronize()
