import sys, os, time, gc, warnings
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
from pq3d_amd.graphed import GraphedQuery3D
cfgname = sys.argv[1] if len(sys.argv) > 1 else "c4"
c = dict(bench.CONFIGS[cfgname])
dev = torch.device("cuda")
model, sd, dd_cpu = bench.build(c, "bf16", dev, seed=1234)
dd = {k: v.to(dev) for k, v in dd_cpu.items()}
model.train()
for m in model.modules():
    if hasattr(m, "dropout_p"): m.dropout_p = 0.0
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for mode in sys.argv[2:] or ["autograd", "direct"]:
    gm = GraphedQuery3D(model, dd, mode=mode)
    def step():
        model.zero_grad(set_to_none=True)
        bench.loss_fn(gm(dd), c["heads"]).backward()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        ms = timed(step)
    print(cfgname, mode, round(ms, 3), "ms; warnings:", len(w), (str(w[0].message)[:80] if w else ""))
    # host-side time of one step without waiting for the device
    torch.cuda.synchronize(); t = time.perf_counter(); step(); host = (time.perf_counter() - t) * 1e3; torch.cuda.synchronize()
    print("   host-side issue time of one step:", round(host, 3), "ms")
    del gm, step; gc.collect()

if os.environ.get("PROBE_PROFILE"):
    from torch.profiler import profile, ProfilerActivity
    gm = GraphedQuery3D(model, dd, mode="autograd")
    def step():
        model.zero_grad(set_to_none=True)
        bench.loss_fn(gm(dd), c["heads"]).backward()
    for _ in range(3): step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(5): step()
        torch.cuda.synchronize()
    rows = [(e.key, e.count, e.cpu_time_total) for e in prof.key_averages()]
    rows.sort(key=lambda r: -r[2])
    for k, n, t in rows[:25]:
        print(f"{k[:70]:70s} {n:6d} {t / 5e3:9.3f} ms/step")
