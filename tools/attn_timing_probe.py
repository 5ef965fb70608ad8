import ctypes, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src = os.path.join(ROOT, "pq3d_amd", "csrc")
out = "/tmp/libpq3d_dbg.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-ffp-contract=off", "-DPQ3D_DEBUG_TIMING", "-shared", "-o", out, *[os.path.join(src, f) for f in ("gemm.hip", "api.cpp", "attention.hip", "norm.hip", "misc.hip")]])
from pq3d_amd import _lib as L
L.LIB_PATH = out
lib = L.lib()
from pq3d_amd import ops
lib.pq3d_attn_debug_read.argtypes = [ctypes.POINTER(ctypes.c_longlong)]
dev = "cuda"
B, H, Nq, Ns, d = 24, 8, 100, 1024, 256
q = torch.randn(B, Nq, d, device=dev).bfloat16(); k = torch.randn(B, Ns, d, device=dev).bfloat16(); v = torch.randn(B, Ns, d, device=dev).bfloat16()
kpm = torch.zeros(B, Ns, dtype=torch.bool, device=dev)
with torch.no_grad():
    for rep in range(3):
        o = ops.attention(q, k, v, H=H, ct=L.BF16, zero_attn=True, kpm=kpm)
        torch.cuda.synchronize()
        buf = (ctypes.c_longlong * 16)(); lib.pq3d_attn_debug_read(buf)
        print("fwd: stage", buf[0], "S", buf[1], "softmax", buf[2], "PV", buf[3], "loop total", buf[4], "(16 iterations)")
