"""Which stream is current inside _Replay.forward / backward at run time?  python tools/probes/dropin_stream_probe.py <config>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import pq3d_amd.graphed as G
cfg = sys.argv[1]
dev = torch.device("cuda", 0)
c = dict(bench.CONFIGS[cfg])
model, sd, dd_cpu = bench.build(c, "bf16", dev, seed=1234)
dd = {k: v.to(dev) for k, v in dd_cpu.items()}
model.train()
gm = G.GraphedQuery3D(model, dd, mode="autograd")
of, ob = G._Replay.forward, G._Replay.backward
seen = {}
def fwd(ctx, *a):
    seen["fwd"] = torch.cuda.current_stream().cuda_stream
    return of(ctx, *a)
def bwd(ctx, *g):
    seen["bwd"] = torch.cuda.current_stream().cuda_stream
    seen["gout_devices"] = sorted({str(x.device) for x in g if x is not None})
    return ob(ctx, *g)
G._Replay.forward, G._Replay.backward = staticmethod(fwd), staticmethod(bwd)
for _ in range(2):
    model.zero_grad(set_to_none=True)
    out = gm(dd)
    first = [(k, str(v.device), v.dtype, v.requires_grad) for k, v in out.items() if torch.is_tensor(v)][:4]
    loss = bench.loss_fn(out, c["heads"])
    loss.backward()
print(cfg, "caller stream", torch.cuda.current_stream().cuda_stream, seen, "static_out[0..3]:",
      [(str(o.device), o.dtype, tuple(o.shape)) for o in gm.static_out[:4]], "gout_idx", gm.gout_idx[:6], "n_out", len(gm.static_out))
