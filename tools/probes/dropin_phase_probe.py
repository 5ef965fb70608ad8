"""Where does GraphedQuery3D spend a step?  Phase timing (device-synchronised) of model.zero_grad / forward replay / loss /
backward, for the two modes.    python tools/probes/dropin_phase_probe.py <config>"""
import os, sys, time
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
sync = torch.cuda.synchronize
for mode in ("direct", "autograd"):
    gm = GraphedQuery3D(model, dd, mode=mode)
    acc = {}
    def phase(name, fn):
        sync(); t = time.perf_counter(); r = fn(); sync()
        acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t) * 1e3
        return r
    for it in range(13):
        if it == 3:
            acc.clear()
        phase("zero_grad", lambda: model.zero_grad(set_to_none=True))
        out = phase("forward", lambda: gm(dd))
        loss = phase("loss", lambda: bench.loss_fn(out, c["heads"]))
        phase("backward", lambda: loss.backward())
    print(cfg, mode, {k: round(v / 10, 3) for k, v in acc.items()}, "sum", round(sum(acc.values()) / 10, 3))
    # unsynchronised step time
    def step():
        model.zero_grad(set_to_none=True)
        bench.loss_fn(gm(dd), c["heads"]).backward()
    for _ in range(3): step()
    sync(); t = time.perf_counter()
    for _ in range(20): step()
    sync(); print(cfg, mode, "pipelined ms/step", round((time.perf_counter() - t) / 20 * 1e3, 3))
    # CPU-only cost of backward: no sync, time the python call
    ts = []
    for _ in range(10):
        model.zero_grad(set_to_none=True)
        loss = bench.loss_fn(gm(dd), c["heads"]); sync()
        t = time.perf_counter(); loss.backward(); ts.append((time.perf_counter() - t) * 1e3); sync()
    print(cfg, mode, "host time inside backward() ms", round(sum(ts) / len(ts), 3))
    del gm
    import gc; sync(); gc.collect()
