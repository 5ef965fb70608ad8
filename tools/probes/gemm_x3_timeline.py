"""Probe: in-kernel timeline (s_memtime stamps of workgroup 0) of the 800-row query-side GEMMs in split-bf16 mode."""
import ctypes, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
src = os.path.join(ROOT, "pq3d_amd", "csrc")
out = "/tmp/libpq3d_dbg.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-ffp-contract=off", "-DPQ3D_DEBUG_TIMING", "-I" + os.path.join(ROOT, "include"), "-shared", "-o", out,
                       *[os.path.join(src, f) for f in ("gemm.hip", "gemm128.hip", "api.cpp", "attention.hip", "norm.hip", "misc.hip", "attn_small.hip", "optim.hip", "loss.hip", "pointnet2.hip")],
                       os.path.join(src, "attn_resident.hip"), "-mllvm", "-amdgpu-mfma-vgpr-form"])
from pq3d_amd import _lib as L
L.LIB_PATH = out
lib = L.lib()
lib.pq3d_debug_read.argtypes = [ctypes.POINTER(ctypes.c_longlong)]
dev = "cuda"
for (M, N, K, ct) in [(800, 256, 256, L.BF16X3), (800, 768, 256, L.BF16X3), (800, 2048, 256, L.BF16X3), (800, 256, 2048, L.BF16X3), (800, 256, 256, L.BF16)]:
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev) * 0.05; Cc = torch.empty(M, N, device=dev)
    for rep in range(3):
        L.gemm(M=M, N=N, K=K, A=[A], B=[B], Cs=[Cc], ct=ct, lda=K, ldb=K, ldc=N)
        torch.cuda.synchronize()
        buf = (ctypes.c_longlong * 16)()
        lib.pq3d_debug_read(buf)
        v = list(buf)[:6]
    print(M, N, K, "ct", ct, "stamps (cycles from start): issue %d  first-put %d  first-mult %d  loop-end %d  epilogue-end %d" % tuple(v[i] - v[0] for i in range(1, 6)))
