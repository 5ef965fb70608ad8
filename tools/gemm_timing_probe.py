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
for (M, N, K) in [(64, 64, 64), (800, 256, 256), (800, 256, 2048)]:
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev) * 0.05; Cc = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    bi = torch.zeros(N, device=dev)
    for rep in range(3):
        L.gemm(M=M, N=N, K=K, A=[A], B=[B], bias=[bi], Cs=[Cc], ct=L.BF16, lda=K, ldb=K, ldc=N)
        torch.cuda.synchronize()
        buf = (ctypes.c_longlong * 16)()
        lib.pq3d_debug_read(buf)
        v = list(buf)[:6]
        print(M, N, K, "rep", rep, "deltas (cycles @100MHz memtime?):", [v[i + 1] - v[i] for i in range(5)], "total", v[5] - v[0])
