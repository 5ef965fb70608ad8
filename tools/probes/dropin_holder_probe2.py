"""Are the parameters' AccumulateGrad nodes of GraphedQuery3D.__init__ (capture stream) still alive afterwards, and which
attribute of the wrapper keeps them?  A tiny backward over all parameters on the default stream takes ~2 ms of GPU-stream
time when they are (one cross-stream event wait per parameter), microseconds when fresh nodes are created.
python tools/probes/dropin_holder_probe2.py <config>"""
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
params = [p for p in model.parameters() if p.requires_grad]
def tiny(tag):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s = torch.stack([p.reshape(-1)[0] for p in params]).sum()
    e0.record(); s.backward(); e1.record()
    torch.cuda.synchronize()
    for p in params: p.grad = None
    print(cfg, tag, "tiny backward over %d parameters: %.3f ms of GPU-stream time" % (len(params), e0.elapsed_time(e1)))
tiny("before wrapper (1st)")
tiny("before wrapper (2nd)")
gm = GraphedQuery3D(model, dd, mode="autograd")
gc.collect()
tiny("after __init__ (1st)")
tiny("after __init__ (2nd)")
for name in ("_bwd", "fwd_graph", "static_out", "static_gout", "zero_gout", "static_gin", "_args", "static_in", "_flat", "reducer"):
    if hasattr(gm, name):
        try:
            setattr(gm, name, None)
        except Exception as e:
            print("cannot clear", name, e); continue
        gc.collect()
        tiny(f"after clearing gm.{name}")
