"""In-kernel timeline of gemm_nt_astat_kernel (probe build -DPQ3D_ASTAT_TIMELINE): wave 0 (loader) and wave 4 (storer) of
the first few workgroups stamp the 100 MHz clock at kernel entry and, for the SECOND tile of their entry, before / after the barrier of
every slice, after the slice's MFMAs, after the store of the previous tile and after the epilogue.
    python tools/build_variant.py astat_tl -DPQ3D_ASTAT_TIMELINE && PQ3D_LIB_PATH=pq3d_amd/libpq3d_hip_astat_tl.so python tools/probes/astat_timeline.py"""
import ctypes as C, sys
sys.path.insert(0, '/root/repo')
import torch
from pq3d_amd import _lib as L
dev = 'cuda'
M, fams, per, N, K = 8192, 6, 4, 256, 256
As = [torch.randn(M, K, device=dev).bfloat16() for _ in range(fams)]
G = fams * per
W = [(torch.randn(N, K, device=dev) * 0.1).bfloat16() for _ in range(G)]
b = [torch.randn(N, device=dev) for _ in range(G)]
C_ = torch.empty(G, M, N, dtype=torch.bfloat16, device=dev)
def call():
    L.gemm(M=M, N=N, K=K, A=[As[g % fams] for g in range(G)], B=W, bias=b, Cs=[C_[g] for g in range(G)], ct=L.BF16, lda=K, ldb=K, ldc=N)
for _ in range(5):
    call()
torch.cuda.synchronize()
buf = (C.c_long * (64 * 16))()
assert L.lib().pq3d_astat_timeline(buf) == 0
t = list(buf)
names = ["entry"] + [f"s{sl}:{x}" for sl in range(4) for x in ("pre-bar", "post-bar", "mfma-done")] + ["epilogue-done", "sl0-store-done"]   # tile 1 of the entry
for wg in range(0, 32, 5):
    for role in (0, 1):
        row = t[(wg * 2 + role) * 16:(wg * 2 + role) * 16 + 15]
        if row[0] == 0:
            continue
        print(f"wg (y={wg // 4}, x={wg % 4}) {'loader wave 0' if role == 0 else 'store wave 4 '}: " +
              "  ".join(f"{n}={(v - row[0]) / 100:.2f}" for n, v in zip(names[1:], row[1:]) if v))
base = min(t[i * 16] for i in range(64) if t[i * 16])
print("kernel-entry spread of the sampled workgroups (us):", sorted(round((t[i * 16] - base) / 100, 2) for i in range(64) if t[i * 16])[:40])
