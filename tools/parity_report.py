#!/usr/bin/env python3
"""Measured parity of the HIP path (fp32 and bf16 compute modes) against the fp32 oracle for every model fixture and
at the full BASELINE sizes (c2, c4 with live and with pinned self-masks).  Run on the GPU box; the output is committed
under profiles/ (parity_rNN.txt, after the per-sublayer table of tools/sublayer_parity.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pq3d_amd.modules import set_compute  # noqa: E402
from tests import util  # noqa: E402


def rel(a, b, floor=0.0):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    fin = torch.isfinite(b) & (b > -1e5)
    return float((a[fin] - b[fin]).abs().max() / max(float(b[fin].abs().max()), floor, 1e-6))


def one(name, args, compute, dev="cuda"):
    _cfg, model, sd, dd = util.model_case(args)
    set_compute(model, compute)
    model.to(dev)
    out = model({k: v.to(dev) for k, v in dd.items()})
    loss = util.synthetic_loss(out, args["heads"], out["query_embeds"])
    grads = args.get("train_grads", True)
    if grads:
        loss.backward()
    oout, collect, oloss, og = util.run_oracle(args, sd, dd, grads=grads)
    q = rel(out["query_embeds"], collect[-1])
    head = rel(out["ground_logits"], oout["ground_logits"]) if "ground" in args["heads"] else float("nan")
    ml = fl = float("nan")
    if "mask" in args["heads"]:
        ml = max(rel(m, r) for m, r in zip(out["predictions_mask"], oout["predictions_mask"]))
        fl = max(float(((m.detach().float().cpu() < 0) != (r < 0)).float().mean())
                 for m, r in zip(out["predictions_mask"], oout["predictions_mask"]))
        head = max(rel(c, r) for c, r in zip(out["predictions_class"], oout["predictions_class"]))
    le = abs(loss.item() - oloss.item()) / max(1.0, abs(oloss.item()))
    l2, l2n, cos = float("nan"), "", float("nan")
    if grads:
        g = dict(model.named_parameters())
        # pairwise_loc_fc (the 5 -> H spatial-bias projection) is reported apart: its gradient is a sum of ~1e5 terms
        # dS_ij / v_ij with v clamped near 1e-6 (transformers.py:226) that cancel to ~1e-3 of their magnitude, so ANY
        # relative noise eps in the upstream gradient shows as ~1e3 eps there (fp32: 2e-3; bf16 gradients: tens of %)
        names = [n for n in sorted(og) if "pairwise_loc_fc" not in n]
        nmax = max(float(og[n].norm()) for n in names)
        l2, l2n = max((float((g[n].grad.detach().float().cpu() - og[n]).norm() / max(float(og[n].norm()), 1e-2 * nmax)), n)
                      for n in names)
        a = torch.cat([g[n].grad.detach().float().cpu().flatten() for n in names]).double()
        b = torch.cat([og[n].flatten() for n in names]).double()
        cos = float((a * b).sum() / (a.norm() * b.norm()))
        pl = [n for n in og if "pairwise_loc_fc" in n]
        plw = max((float((g[n].grad.detach().float().cpu() - og[n]).norm() / float(og[n].norm())) for n in pl), default=float("nan"))
        l2n = f"{l2n}   [pairwise_loc_fc worst relL2 {plw:.2e}]"
    print(f"{name:22s} {compute:7s} {q:9.2e} {head:9.2e} {ml:10.2e} {fl:10.2e} {le:9.2e} {l2:11.2e} {1 - cos:10.2e}  {l2n}")


def main():
    print(f"{'case':22s} {'mode':7s} {'query':>9s} {'head':>9s} {'mask-logit':>10s} {'flip-rate':>10s} {'loss':>9s} "
          f"{'worst-relL2':>11s} {'1-cos(all)':>10s}  worst-gradient parameter     (all vs the fp32 oracle; max|err|/max|ref|)")
    for name in util.model_fixtures():
        _z, args = util.load_fixture(name)
        for compute in ("fp32", "bf16x3", "bf16"):
            one(name, args, compute)
    from tests.test_gpu_fullsize import C2, C4, C4_PINNED
    for name, args in (("FULL c2", C2), ("FULL c4 live masks", C4), ("FULL c4 pinned masks", C4_PINNED)):
        for compute in ("fp32", "bf16x3", "bf16"):
            one(name, args, compute)


if __name__ == "__main__":
    main()
