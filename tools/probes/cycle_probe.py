"""Which objects of an eager forward+backward end up in reference cycles (collected only by the cyclic GC)?
python tools/probes/cycle_probe.py <config>"""
import gc, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
cfg = sys.argv[1]
dev = torch.device("cuda", 0)
c = dict(bench.CONFIGS[cfg])
model, sd, dd_cpu = bench.build(c, "bf16", dev, seed=1234)
dd = {k: v.to(dev) for k, v in dd_cpu.items()}
model.train()
def step(backward=True):
    model.zero_grad(set_to_none=True)
    out = model(dict(dd))
    loss = bench.loss_fn(out, c["heads"])
    if backward:
        loss.backward()
for bw in (True, False):
    step(bw); gc.collect()
    gc.set_debug(gc.DEBUG_SAVEALL)
    step(bw)
    n = gc.collect()
    gc.set_debug(0)
    kinds = collections.Counter(type(o).__name__ for o in gc.garbage)
    print(cfg, "backward" if bw else "forward only (graph dropped)", ": objects only the cycle collector freed:", n, kinds.most_common(12))
    for o in gc.garbage:
        if type(o).__name__ in ("function", "cell") or "Backward" in type(o).__name__:
            if type(o).__name__ == "function":
                print("   function", o.__qualname__, "closure cells:", [type(cc.cell_contents).__name__ for cc in (o.__closure__ or ())][:10])
    del gc.garbage[:]
