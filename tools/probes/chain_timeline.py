"""In-kernel timeline of the one-launch chain (csrc/chain_ffn.hip, probe build with -DPQ3D_CHAIN_TL): workgroup 0's stamps at
the step boundaries of the last of 200 back-to-back launches.  usage: PQ3D_LIB_PATH=<probe .so> python tools/probes/chain_timeline.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pq3d_amd import ops
dev = torch.device("cuda")
B, Nq, d, F_ = 8, 100, 256, 2048
g = torch.Generator().manual_seed(0)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
args = (r(B, Nq, d), r(d, d, sc=0.06), r(d, sc=0.1), r(B, Nq, d), 1 + r(d, sc=0.1), r(d, sc=0.1), 1e-5,
        r(F_, d, sc=0.06), r(F_, sc=0.1), r(d, F_, sc=0.03), r(d, sc=0.1), 1 + r(d, sc=0.1), r(d, sc=0.1), 1e-5)
flags = ops.chain_flags(B * Nq, dev)
dev = args[0].device
ops._CHAIN_ERR[dev] = torch.zeros(64, dtype=torch.int32, device=dev)   # room for the stamps
for _ in range(200):
    ops.chain_ffn_fwd(*args, flags)
torch.cuda.synchronize()
t = ops._CHAIN_ERR[dev].view(torch.int64).cpu().tolist()[:10]
names = ["out-projection tile", "hand-off 1", "LayerNorm 1", "hand-off 2", "linear1 (4 k slabs)", "hand-off 3", "linear2 (4 k slabs)", "hand-off 4", "LayerNorm 2"]
for n, a, b in zip(names, t[:-1], t[1:]):
    print(f"  {n:28s} {(b - a) * 0.01:6.2f} us")
print(f"  total in-kernel {(t[9] - t[0]) * 0.01:.2f} us")
