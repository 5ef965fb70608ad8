"""Which parameter gradients OUTSIDE the decoder are final (and in their arena slot) when the fused decoder's backward starts
(enc.grads_ready('heads')) -- the precondition for all-reducing the heads' bucket under the decoder backward.
usage: python tools/probes/heads_arena_probe.py [c5|s2|c4]"""
import sys
sys.path.insert(0, '/root/repo')
import torch
import bench
from pq3d_amd import ops
from pq3d_amd.parallel import FlatGradAllReducer

cfg = sys.argv[1] if len(sys.argv) > 1 else "c5"
c = dict(bench.CONFIGS[cfg])
dev = torch.device("cuda", 0)
model, sd, dd = bench.build(c, "bf16", dev, seed=1234)
dd = {k: v.to(dev) for k, v in dd.items()}
model.train()
for m in model.modules():
    if hasattr(m, "dropout_p"):
        m.dropout_p = 0.0
params = [p for p in model.parameters() if p.requires_grad]
names = {id(p): n for n, p in model.named_parameters()}
enc = model.unified_encoder
dec_ids = {id(p) for p in enc.parameters()} | ({id(p) for p in model.mask_head.parameters()} if hasattr(model, "mask_head") else set())
head_ids = set()
for h in ("generation_head", "ground_head"):
    if hasattr(model, h):
        head_ids |= {id(p) for p in getattr(model, h).parameters()}
groups = [[p for p in params if id(p) in dec_ids], [p for p in params if id(p) in head_ids],
          [p for p in params if id(p) not in dec_ids and id(p) not in head_ids]]
reducer = FlatGradAllReducer(params, groups=[g for g in groups if g])
slots = reducer.slots()
enc.grad_arena, enc.grad_arena_buffers = slots, list(reducer.flat)
snap = {}


def on_ready(tag):
    if tag == "heads":
        for p in groups[1]:
            flat, off, n = slots[id(p)]
            snap[id(p)] = flat[off:off + n].clone()


enc.grads_ready, enc.grad_bucket_per_layer = on_ready, False
one = torch.ones((), device=dev)
for it in range(2):
    model.zero_grad(set_to_none=True)
    out = model(dict(dd))
    loss = bench.loss_fn(out, c["heads"])
    with ops.grad_arena(slots, list(reducer.flat), pack_follows=True):
        loss.backward(gradient=one)
    reducer.pack()
torch.cuda.synchronize()
bad = []
for p in groups[1]:
    flat, off, n = slots[id(p)]
    final = flat[off:off + n]
    if p.grad is None:
        continue
    if not torch.equal(final, snap[id(p)]):
        rel = float((final - snap[id(p)]).norm() / (final.norm() + 1e-30))
        bad.append((names[id(p)], tuple(p.shape), rel, bool(p.grad.data_ptr() == final.data_ptr())))
print(f"{cfg}: {len(groups[1])} head parameters, {len(bad)} NOT final at 'heads' time")
for b in bad[:60]:
    print("  ", b)
