"""Who keeps the captured forward's autograd graph alive after GraphedQuery3D.__init__?  Lists live tensors that still carry
a grad_fn and what refers to them.    python tools/probes/graph_holder_probe.py <config>"""
import gc, os, sys
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
gm = GraphedQuery3D(model, dd, mode="autograd")
gc.collect()
found = [o for o in gc.get_objects() if isinstance(o, torch.Tensor) and o.grad_fn is not None]
print(cfg, "live tensors with a grad_fn:", len(found))
for t in found[:30]:
    refs = [r for r in gc.get_referrers(t) if r is not found]
    desc = []
    for r in refs[:6]:
        if isinstance(r, dict):
            owners = [type(o).__name__ for o in gc.get_referrers(r) if hasattr(o, "__dict__") and o.__dict__ is r][:2]
            keys = [k for k, v in r.items() if v is t][:3]
            desc.append(f"dict{keys} of {owners}")
        elif isinstance(r, (list, tuple)):
            owners = []
            for o in gc.get_referrers(r)[:4]:
                if isinstance(o, dict):
                    owners += [f"{type(oo).__name__}.{k}" for oo in gc.get_referrers(o) if hasattr(oo, "__dict__") and oo.__dict__ is o for k, v in o.items() if v is r][:2]
                else:
                    owners.append(type(o).__name__)
            desc.append(f"{type(r).__name__}[{len(r)}] <- {owners}")
        else:
            desc.append(type(r).__name__)
    print("  ", tuple(t.shape), t.dtype, type(t.grad_fn).__name__, "<-", desc)

import torch.autograd.function as F_
nodes = [o for o in gc.get_objects() if isinstance(o, F_.BackwardCFunction)]
print("live custom-Function backward nodes (ctx objects):", len(nodes))
import collections
print(collections.Counter(type(n).__name__ for n in nodes).most_common(20))
for n in nodes[:8]:
    refs = gc.get_referrers(n)
    print("  ", type(n).__name__, "<-", [type(r).__name__ + (str([k for k, v in r.items() if v is n][:2]) if isinstance(r, dict) else "") for r in refs[:6]])
