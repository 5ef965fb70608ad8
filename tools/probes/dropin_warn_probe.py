"""Counts AccumulateGrad stream-mismatch warnings: during GraphedQuery3D.__init__ vs during steps; checks whether a
parameter's AccumulateGrad node survives between steps.  python tools/probes/dropin_warn_probe.py <config>"""
import os, sys, warnings
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
warnings.simplefilter("always")
with warnings.catch_warnings(record=True) as w0:
    warnings.simplefilter("always")
    gm = GraphedQuery3D(model, dd, mode="autograd")
print(cfg, "warnings in __init__:", sum("AccumulateGrad" in str(x.message) for x in w0))
p = gm._params[0]
def acc_node(p):
    return p.expand_as(p).grad_fn.next_functions[0][0]
n0 = acc_node(p); id0 = id(n0); del n0
for it in range(3):
    with warnings.catch_warnings(record=True) as w1:
        warnings.simplefilter("always")
        model.zero_grad(set_to_none=True)
        loss = bench.loss_fn(gm(dd), c["heads"])
        loss.backward()
        del loss
    print(cfg, "step", it, "stream-mismatch warnings:", sum("AccumulateGrad" in str(x.message) for x in w1))
print("current stream", torch.cuda.current_stream(), "default", torch.cuda.default_stream())
