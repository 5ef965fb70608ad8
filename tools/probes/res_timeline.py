"""In-kernel timeline of the all-queries-resident cross-attention backward (csrc/attn_resident.hip) at the config-2 shape
(24 stacked scenes x 8 heads, 100 queries, 1024 keys, bf16): probe copy of the library with -DPQ3D_RES_TIMELINE, 300 launches
back to back, stamps of the last one.  usage (GPU box): python tools/probes/res_timeline.py"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
PROBE = os.path.join(ROOT, "pq3d_amd", "build", "libpq3d_probe_res.so")


def build_probe():
    from pq3d_amd import build as B
    objs = []
    for src in B.SOURCES:
        obj = os.path.join(B.HERE, "build", src + ".o")
        if src == "attn_resident.hip":
            obj = os.path.join(B.HERE, "build", "attn_resident_probe.o")
            subprocess.check_call(["/opt/rocm/bin/hipcc", *B.FLAGS, *B.EXTRA.get(src, []), "-DPQ3D_RES_TIMELINE", "-x", "hip", "-c",
                                   os.path.join(B.CSRC, src), "-o", obj])
        objs.append(obj)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", PROBE, *objs])


def main():
    import torch
    from pq3d_amd import _lib as L, fused
    B, H, Lq, Lk, d = 24, 8, 100, 1024, 256
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g).to(dev).bfloat16()
    q, k, v, go = r(B, Lq, d), r(B, Lk, d), r(B, Lk, d), r(B, Lq, d)
    kpm = torch.zeros(B, Lk, dtype=torch.bool, device=dev)
    kpm[:, 1000:] = True
    o, lse = torch.empty_like(q), torch.empty(B, H, Lq, device=dev)
    fused._attn(q, k, v, o, lse, H, L.BF16, True, kpm=kpm)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty(B, H, Lq, device=dev)
    ws = torch.zeros(16, dtype=torch.int64, device=dev)
    for it in range(300):
        dd = fused.ops._attn_desc(q, k, v, o, lse, H, L.BF16, True, 1.0 / 32 ** 0.5, kpm, None, None, None, None, 0, bwd=True)
        dd.dout, dd.dq, dd.dk, dd.dv, dd.delta, dd.dbias = map(L.ptr, (go, dq, dk, dv, delta, None))
        assert dd.ksplit <= 1, dd.ksplit
        dd.ws = L.ptr(ws)
        L.check(L.lib().pq3d_attn_bwd(L.C.byref(dd), L.stream()), "bwd")
    torch.cuda.synchronize()
    t = ws.cpu().tolist()
    print("stamps (10 ns ticks from entry):", [x - t[0] if x else None for x in t[:11]])
    names = ["stage Q/dO", "sync", "group 0", "group 1", "group 2", "group 3", "(group 4)", "loop exit", "sync", "dQ reduce + store"]
    prev = t[0]
    for n, x in zip(names, t[1:11]):
        if x:
            print(f"   {n}: {(x - prev) * 0.01:.2f} us")
            prev = x


if __name__ == "__main__":
    if os.environ.get("PQ3D_LIB_PATH") != PROBE:
        build_probe()
        os.environ["PQ3D_LIB_PATH"] = PROBE
        os.execv(sys.executable, [sys.executable] + sys.argv)
    main()
