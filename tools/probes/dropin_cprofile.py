"""cProfile of the drop-in step loop (host side).  python tools/probes/dropin_cprofile.py <config> <mode>"""
import os, sys, cProfile, pstats, io, time
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
def step():
    model.zero_grad(set_to_none=True)
    bench.loss_fn(gm(dd), c["heads"]).backward()
for _ in range(5): step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
print(cfg, mode, "ms/step", round((time.perf_counter() - t) / 20 * 1e3, 3))
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:4500])
