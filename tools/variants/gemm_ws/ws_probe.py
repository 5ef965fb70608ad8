"""Weight-stationary K/V projection (gemm_ws.hip) against the 128 x 128-tile kernel: same bits, time per launch.
usage (GPU box): python tools/probes/ws_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pq3d_amd import _lib as L

dev = "cuda"
# mode bits: 1 on, 2 fragment-major input, 4 nontemporal stores, 8 no stores (diagnostic), 16 no activation loads (diagnostic)
MODES = (0, 3)


def run(name, G, R, mems, relu=False, bias=True, reps=30):
    torch.manual_seed(0)
    xs = [torch.randn(R, 256, device=dev).bfloat16() for _ in range(mems)]
    ws = [(torch.randn(256, 256, device=dev) * 0.06).bfloat16() for _ in range(G)]
    bs = [torch.randn(256, device=dev) if bias else None for _ in range(G)]
    outs = {}
    R16 = (R + 15) // 16 * 16
    def blocked(x):   # [R, 256] -> fragment-major: [R/16][8 k-steps][4 lg][16 li][8]
        xp = torch.zeros(R16, 256, device=dev, dtype=x.dtype); xp[:R] = x
        return xp.view(R16 // 16, 16, 8, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(R16, 256)
    xb = [blocked(x) for x in xs]
    for on in MODES:
        L.lib().pq3d_gemm_ws(on)
        xs_ = xb if on & 2 else xs
        C = torch.full((G, R, 256), float("nan"), device=dev).bfloat16()
        f = lambda: L.gemm(M=R, N=256, K=256, A=[xs_[g % mems] for g in range(G)], B=ws, bias=bs if bias else None,
                           Cs=[C[g] for g in range(G)], ct=L.BF16, lda=256, ldb=256, ldc=256, act="relu" if relu else None)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        torch.cuda.synchronize()
        outs[on] = (C.clone(), e0.elapsed_time(e1) / reps * 1e3)
    L.lib().pq3d_gemm_ws(1)
    same = {m: torch.equal(outs[0][0].view(torch.int16), outs[m][0].view(torch.int16)) for m in MODES}
    ref = torch.stack([xs[g % mems].float() @ ws[g].float().t() + (bs[g] if bias else 0) for g in range(min(G, 3))])
    if relu:
        ref = ref.relu()
    err = (outs[3][0][:min(G, 3)].float() - ref).abs().max().item()
    byt = (G * R * 256 * 2 + mems * R * 256 * 2 + G * 256 * 256 * 2) / 1e6
    print(f"{name}: G {G} R {R} unique MB {byt:.0f} err {err:.2e} | " + "  ".join(f"m{m}: {outs[m][1]:.1f}us{'' if same[m] else '*'}" for m in MODES))


if __name__ == "__main__":
    run("c2", 24, 8192, 3)
    run("c4", 24, 16384, 3)
    run("c5", 32, 32768, 3)
    run("c2 relu nobias", 24, 8192, 3, relu=True, bias=False)
    run("ragged M", 24, 8192 - 72, 3)
    run("g8", 8, 16384, 1)
