#!/usr/bin/env python3
"""Print the measured parity errors of the HIP path (fp32 and bf16 compute) against the golden fixtures /
the oracle for every model fixture.  Run on the GPU box; the output is committed under profiles/."""
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


def main():
    dev = "cuda"
    print(f"{'fixture':18s} {'compute':8s} {'vs':16s} {'query':>9s} {'head':>9s} {'mask-logit':>10s} {'flip-rate':>10s} "
          f"{'loss':>9s} {'worst-grad':>10s}  worst-grad-name")
    model_fixtures = [f for f in util.fixtures()
                      if not f.startswith(("F3_", "F6_", "F7_", "F8_", "F9_", "F10_", "F11_", "F12_"))]
    for name in model_fixtures:
        z, args = util.load_fixture(name)
        for compute, emu in (("fp32", None), ("bf16", torch.bfloat16), ("bf16", None)):
            _cfg, model, sd, dd = util.model_case(args)
            set_compute(model, compute)
            model.to(dev)
            out = model({k: v.to(dev) for k, v in dd.items()})
            grads = any(k.startswith("grad/") for k in z.files)
            loss = util.synthetic_loss(out, args["heads"], out["query_embeds"])
            if grads:
                loss.backward()
            oout, collect, oloss, og = util.run_oracle(args, sd, dd, grads=grads, emulate=emu)
            q = rel(out["query_embeds"], collect[-1])
            head = rel(out["ground_logits"], oout["ground_logits"]) if "ground" in args["heads"] else float("nan")
            ml = fl = float("nan")
            if "mask" in args["heads"]:
                ml = max(rel(m, r) for m, r in zip(out["predictions_mask"], oout["predictions_mask"]))
                fl = max(float(((m.detach().float().cpu() < 0) != (r < 0)).float().mean())
                         for m, r in zip(out["predictions_mask"], oout["predictions_mask"]))
                head = max(rel(c, r) for c, r in zip(out["predictions_class"], oout["predictions_class"]))
            le = abs(loss.item() - oloss.item()) / max(1.0, abs(oloss.item()))
            wg, wn = float("nan"), ""
            if grads:
                gmax = max(float(v.abs().max()) for v in og.values())
                g = dict(model.named_parameters())
                wg, wn = max((rel(g[n].grad, og[n], floor=1e-2 * gmax), n) for n in og)
                nmax = max(float(v.norm()) for v in og.values())
                l2, l2n = max((float((g[n].grad.detach().float().cpu() - og[n]).norm()
                                     / max(float(og[n].norm()), 1e-2 * nmax)), n) for n in og)
                wn = f"{wn}  | worst relL2 {l2:.2e} {l2n}"
            vs = "oracle fp32" if emu is None else "oracle bf16-round"
            print(f"{name:18s} {compute:8s} {vs:16s} {q:9.2e} {head:9.2e} {ml:10.2e} {fl:10.2e} {le:9.2e} {wg:10.2e}  {wn}")


if __name__ == "__main__":
    main()
