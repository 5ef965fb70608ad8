"""Why does F15_d768 (d 768, H 12) miss the 3e-4 gradient tolerance on ffn.linear1.bias in fp32?  Count ReLU sign flips."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import util
from oracle import pq3d_oracle as O
z, args = util.load_fixture("F15_d768")
_cfg, model, sd, dd = util.model_case(args)
model.to("cuda")
out = model({k: v.to("cuda") for k, v in dd.items()})
loss = util.synthetic_loss(out, args["heads"], out["query_embeds"])
loss.backward()
oout, collect, oloss, og = util.run_oracle(args, sd, dd)
g = dict(model.named_parameters())
for n in ("unified_encoder.unified_encoder.0.ffn.linear1.bias", "unified_encoder.unified_encoder.0.ffn.linear1.weight",
          "unified_encoder.unified_encoder.0.self_attn.self_attn.w_qs.weight"):
    a, b = g[n].grad.float().cpu(), og[n]
    e = (a - b).abs()
    idx = e.flatten().topk(5).indices
    print(n, "max err", float(e.max()), "ref max", float(b.abs().max()), "relL2", float((a - b).norm() / b.norm()))
    print("   top errs", [(int(i), float(e.flatten()[i]), float(b.flatten()[i])) for i in idx])
# fp64 oracle as the arbiter
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
dd64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in dd.items()}
o64, c64, l64, g64 = util.run_oracle(args, sd64, dd64)
for n in ("unified_encoder.unified_encoder.0.ffn.linear1.bias", "unified_encoder.unified_encoder.0.self_attn.self_attn.w_qs.weight"):
    a, b, c = g[n].grad.double().cpu(), og[n].double(), g64[n]
    print(n, "HIP vs fp64", float((a - c).norm() / c.norm()), " oracle-fp32 vs fp64", float((b - c).norm() / c.norm()))
