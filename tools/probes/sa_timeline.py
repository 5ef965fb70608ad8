"""In-kernel timeline of the split-bf16 self-attention backward (csrc/attn_sa.hip) at the config-2 shape: builds a probe copy of
the library with -DPQ3D_SA_TIMELINE (wave 0 of workgroup (0, 0) stamps the 100 MHz clock at the phase boundaries), runs the
kernel with and without the folded out-projection and prints the phase durations.  usage (GPU box):
    python tools/probes/sa_timeline.py            # builds pq3d_amd/build/libpq3d_probe.so, then re-executes itself on it"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
PROBE = os.path.join(ROOT, "pq3d_amd", "build", "libpq3d_probe.so")


def build_probe():
    from pq3d_amd import build as B
    objs = []
    for src in B.SOURCES:
        obj = os.path.join(B.HERE, "build", src + ".o")
        if src == "attn_sa.hip":
            obj = os.path.join(B.HERE, "build", "attn_sa_probe.o")
            subprocess.check_call(["/opt/rocm/bin/hipcc", *B.FLAGS, *B.EXTRA.get(src, []), "-DPQ3D_SA_TIMELINE", "-x", "hip", "-c",
                                   os.path.join(B.CSRC, src), "-o", obj])
        objs.append(obj)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", PROBE, *objs])


def main():
    import torch
    from pq3d_amd import _lib as L, fused
    B, H, Lq, d = 8, 8, 100, 256
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    q, k, v, go, W = r(B, Lq, d), r(B, Lq, d), r(B, Lq, d), r(B, Lq, d), r(d, d) * 0.06
    bias = r(B, H, Lq, Lq)
    kpm = torch.zeros(B, Lq, dtype=torch.bool, device=dev)
    o, lse = torch.empty_like(q), torch.empty(B, H, Lq, device=dev)
    fused._attn(q, k, v, o, lse, H, L.BF16X3, False, kpm=kpm, bias=bias)
    names = ["addr+issue W", "stage Q(GKV)+kb", "sync W", "dO / delta", "sync", "phase A", "phase B", "stores"]
    for fold in (False, True):
        dqkv, delta, dsb = torch.empty(3, B, Lq, d, device=dev), torch.empty(B, H, Lq, device=dev), torch.empty_like(bias)
        ws = torch.zeros(16, dtype=torch.int64, device=dev)
        rows = []
        for it in range(300):   # back to back (clocks up); the stamps of the last launch are read
            dd = fused.ops._attn_desc(q, k, v, o, lse, H, L.BF16X3, False, 1.0 / 32 ** 0.5, kpm, None, None, bias, None, 0, bwd=True)
            dd.dout, dd.dq, dd.dk, dd.dv, dd.delta, dd.dbias = map(L.ptr, (None if fold else go, dqkv[0], dqkv[1], dqkv[2], delta, dsb))
            dd.ws = L.ptr(ws)
            if fold:
                dd.proj.mode, dd.proj.dm, dd.proj.x = 2, d, L.ptr(go)
                dd.proj.w[0] = L.ptr(W)
            L.check(L.lib().pq3d_attn_bwd(L.C.byref(dd), L.stream()), "bwd")
        torch.cuda.synchronize()
        t = ws.cpu().tolist()
        print("   phase A, first step: " + ", ".join(f"{n}: {(b_ - a_) * 0.01:.2f} us" for n, a_, b_ in zip(
            ["entry->bias ready", "S, dP products", "softmax arithmetic", "dQ products", "second step"], [t[4]] + t[8:12], t[8:13])))
        print(f"fold={fold}: stamps (10 ns ticks from kernel entry): {[x - t[0] for x in t]}")
        st = [t[0], t[1], t[2] if fold else t[1], t[3], t[4], t[5], t[6], t[7]]
        print("   " + ", ".join(f"{n}: {(b_ - a_) * 0.01:.2f} us" for n, a_, b_ in zip(
            ["stage", "sync W", "dO+delta", "sync", "phase A", "phase B", "stores"], st[:-1], st[1:])))


if __name__ == "__main__":
    if os.environ.get("PQ3D_LIB_PATH") != PROBE:
        build_probe()
        os.environ["PQ3D_LIB_PATH"] = PROBE
        os.execv(sys.executable, [sys.executable] + sys.argv)
    main()
