"""After one autograd-mode step of GraphedQuery3D: which parameters' .grad do NOT alias the flat buffers (AccumulateGrad
cloned instead of adopting the view)?    python tools/probes/dropin_alias_probe.py <config>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from pq3d_amd.graphed import GraphedQuery3D
cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
dev = torch.device("cuda", 0)
c = dict(bench.CONFIGS[cfg])
model, sd, dd_cpu = bench.build(c, "bf16", dev, seed=1234)
dd = {k: v.to(dev) for k, v in dd_cpu.items()}
model.train()
for m in model.modules():
    if hasattr(m, "dropout_p"):
        m.dropout_p = 0.0
gm = GraphedQuery3D(model, dd, mode="autograd")
for _ in range(2):
    model.zero_grad(set_to_none=True)
    bench.loss_fn(gm(dd), c["heads"]).backward()
names = {id(p): n for n, p in model.named_parameters()}
bad = [(names[id(p)], tuple(p.shape), p.is_contiguous(), None if p.grad is None else p.grad.is_contiguous()) for p in gm._params
       if id(p) not in gm._unused and (p.grad is None or p.grad.data_ptr() != gm._grad_view(p).data_ptr())]
print(cfg, "params", len(gm._params), "unused", len(gm._unused), "not aliasing", len(bad))
for b in bad[:40]:
    print("  ", b)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        model.zero_grad(set_to_none=True)
        bench.loss_fn(gm(dd), c["heads"]).backward()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=30, max_name_column_width=60))
