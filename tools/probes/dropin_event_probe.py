"""GPU-side segment times of the drop-in step (events on the current stream, no host syncs inside the loop).
python tools/probes/dropin_event_probe.py <config> <mode>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from pq3d_amd.graphed import GraphedQuery3D
cfg, mode = sys.argv[1], sys.argv[2]
dev = torch.device("cuda", 0)
c = dict(bench.CONFIGS[cfg])
model, sd, dd_cpu = bench.build(c, "bf16", dev, seed=1234)
dd = {k: v.to(dev) for k, v in dd_cpu.items()}
model.train()
for m in model.modules():
    if hasattr(m, "dropout_p"):
        m.dropout_p = 0.0
gm = GraphedQuery3D(model, dd, mode=mode)
N = 20
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(N)]
def step(e=None):
    if e: e[0].record()
    model.zero_grad(set_to_none=True)
    out = gm(dd)
    if e: e[1].record()
    loss = bench.loss_fn(out, c["heads"])
    if e: e[2].record()
    loss.backward()
    if e: e[3].record()
for _ in range(5): step()
torch.cuda.synchronize()
for i in range(N): step(ev[i])
torch.cuda.synchronize()
seg = [sum(ev[i][k].elapsed_time(ev[i][k + 1]) for i in range(5, N)) / (N - 5) for k in range(3)]
gap = sum(ev[i][3].elapsed_time(ev[i + 1][0]) for i in range(5, N - 1)) / (N - 6)
print(cfg, mode, "GPU ms: forward %.3f  loss %.3f  backward %.3f  between steps %.3f" % (*seg, gap))
