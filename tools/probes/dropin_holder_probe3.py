"""Do the parameters' AccumulateGrad nodes survive from one graph to the next?  Tags each node's metadata dict, drops the
Python references, and looks for the tag later.    python tools/probes/dropin_holder_probe3.py <config>"""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from pq3d_amd.graphed import GraphedQuery3D
cfg = sys.argv[1]
dev = torch.device("cuda", 0)
c = dict(bench.CONFIGS[cfg])
model, sd, dd_cpu = bench.build(c, "bf16", dev, seed=1234)
dd = {k: v.to(dev) for k, v in dd_cpu.items()}
model.train()
names = {id(p): n for n, p in model.named_parameters()}
params = [p for p in model.parameters() if p.requires_grad]
def tag(t):
    for p in params:
        n = p.expand_as(p).grad_fn.next_functions[0][0]
        n.metadata["tag"] = t
        del n
    gc.collect()
def survivors(t):
    out = []
    for p in params:
        n = p.expand_as(p).grad_fn.next_functions[0][0]
        if n.metadata.get("tag") == t:
            out.append(names[id(p)])
        del n
    gc.collect()
    return out
tag("pre")
s = survivors("pre")
print(cfg, "control (nothing holds them): survivors", len(s))
gm = GraphedQuery3D(model, dd, mode="autograd")
gc.collect()
tag("post-init")
s = survivors("post-init")
print(cfg, "after __init__, tagged then re-fetched: survivors", len(s), s[:6])
model.zero_grad(set_to_none=True)
loss = bench.loss_fn(gm(dd), c["heads"])
tag("in-step")          # nodes of the live graph
loss.backward()
del loss
gc.collect()
s = survivors("in-step")
print(cfg, "after the step's graph is gone: survivors", len(s), s[:10])
