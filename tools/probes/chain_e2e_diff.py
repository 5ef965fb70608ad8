"""Gradient differences of one model step between chain launches on / off, next to the run-to-run noise of each setting.
usage (GPU box): python tools/probes/chain_e2e_diff.py [mask_head|plain]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests import util
from pq3d_amd import fused, ops
from pq3d_amd.modules import set_compute

case = sys.argv[1] if len(sys.argv) > 1 else "mask_head"
dev = torch.device("cuda")
if case == "mask_head":
    args = dict(B=4, Ns=512, Nq=200, d=256, H=8, L=2, memories=["voxel", "mv", "pc"], heads=["mask"], spatial=True,
                structure="parallel", use_self_mask=True, C=201, foc=(0, 2), seed=0, data_seed=1234)
else:
    args = dict(B=8, Ns=256, Nq=100, d=256, H=8, L=2, memories=["voxel", "mv", "pc"], heads=[], spatial=True,
                structure="parallel", seed=0, data_seed=1234)
_cfg, model, _sd, dd = util.model_case(args)
set_compute(model, "bf16")
model.unified_encoder.fused = True
model.to(dev)
ddv = {k: v.to(dev) for k, v in dd.items()}


def step(on):
    fused.set_chain(on)
    model.zero_grad()
    out = model(dict(ddv))
    util.synthetic_loss(out, args["heads"], out["query_embeds"]).backward()
    torch.cuda.synchronize()
    return {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}


def diff(a, b, label):
    gmax = max(float(v.norm()) for v in b.values())
    errs = sorted(((float((a[n] - b[n]).norm() / max(float(b[n].norm()), 1e-3 * gmax)), n) for n in b), reverse=True)
    print(label, " ".join(f"{n}: {e:.2e}" for e, n in errs[:6]))
    if os.environ.get("ALL"):
        d_ = dict((n, e) for e, n in errs)
        for n in b:
            print(f"    {d_[n]:.2e}  {n}")


off1, off2, on1, on2 = step(False), step(False), step(True), step(True)
diff(off2, off1, "off vs off:")
diff(on2, on1, "on vs on:  ")
diff(on1, off1, "on vs off: ")
