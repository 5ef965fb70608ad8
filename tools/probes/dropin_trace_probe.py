"""Chrome trace of 4 drop-in steps (GraphedQuery3D) for offline gap analysis.  python tools/probes/dropin_trace_probe.py <config> <mode> <out.json>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from pq3d_amd.graphed import GraphedQuery3D
from torch.profiler import profile, ProfilerActivity
cfg, mode, outp = sys.argv[1], sys.argv[2], sys.argv[3]
dev = torch.device("cuda", 0)
c = dict(bench.CONFIGS[cfg])
model, sd, dd_cpu = bench.build(c, "bf16", dev, seed=1234)
dd = {k: v.to(dev) for k, v in dd_cpu.items()}
model.train()
for m in model.modules():
    if hasattr(m, "dropout_p"):
        m.dropout_p = 0.0
gm = GraphedQuery3D(model, dd, mode=mode)
def step():
    model.zero_grad(set_to_none=True)
    bench.loss_fn(gm(dd), c["heads"]).backward()
for _ in range(5): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(4): step()
    torch.cuda.synchronize()
prof.export_chrome_trace(outp)
