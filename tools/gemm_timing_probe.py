"""Build a -DPQ3D_DEBUG_TIMING copy of the library and print s_memtime deltas inside the GEMM kernel."""
import ctypes, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src = os.path.join(ROOT, "pq3d_amd", "csrc")
out = "/tmp/libpq3d_dbg.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-ffp-contract=off", "-DPQ3D_DEBUG_TIMING", "-shared", "-o", out, *[os.path.join(src, f) for f in ("gemm.hip", "api.cpp", "attention.hip", "norm.hip", "misc.hip")]])
from pq3d_amd import _lib as L
L.LIB_PATH = out
lib = L.lib()
lib.pq3d_debug_read.argtypes = [ctypes.POINTER(ctypes.c_longlong)]
dev = "cuda"
for (M, N, K, da, db, a2) in [(800, 256, 2048, "f", "f", 0), (800, 256, 2048, "b", "b", 0), (800, 256, 2048, "f", "f", 1), (800, 256, 2048, "b", "f", 0), (64, 64, 2048, "b", "b", 0), (8192, 256, 2048, "b", "b", 0)]:
    dt = lambda c: torch.float32 if c == "f" else torch.bfloat16
    A = torch.randn(M, K, device=dev).to(dt(da)); B = (torch.randn(N, K, device=dev) * 0.05).to(dt(db)); Cc = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    A2 = torch.randn(M, K, device=dev) if a2 else None
    for rep in range(2):
        L.gemm(M=M, N=N, K=K, A=[A], A2=[A2], B=[B], Cs=[Cc], ct=L.BF16, lda=K, ldb=K, ldc=N)
        torch.cuda.synchronize()
        buf = (ctypes.c_longlong * 16)()
        lib.pq3d_debug_read(buf)
        v = list(buf)[:6]
    print(M, N, K, da, db, "A2" if a2 else "--", "deltas:", [v[i + 1] - v[i] for i in range(5)], "per-iter", (v[4] - v[3]) // 31, "total", v[5] - v[0])
