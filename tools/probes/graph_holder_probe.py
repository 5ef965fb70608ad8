import sys, os, gc
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from pq3d_amd.graphed import GraphedQuery3D
c = dict(bench.CONFIGS["c4"]); dev = torch.device("cuda")
model, sd, dd_cpu = bench.build(c, "bf16", dev, seed=1234)
dd = {k: v.to(dev) for k, v in dd_cpu.items()}
model.train()
for m in model.modules():
    if hasattr(m, "dropout_p"): m.dropout_p = 0.0
gm = GraphedQuery3D(model, dd, mode="autograd")
gc.collect()
n = 0
for o in gc.get_objects():
    try:
        if torch.is_tensor(o) and o.grad_fn is not None:
            n += 1
            if n <= 12:
                refs = [type(r).__name__ + (":" + ",".join(k for k, v in r.items() if v is o)[:60] if isinstance(r, dict) else "") for r in gc.get_referrers(o)][:6]
                print(tuple(o.shape), type(o.grad_fn).__name__, refs)
    except Exception:
        pass
print("tensors with grad_fn alive:", n)
