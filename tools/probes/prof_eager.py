import sys, cProfile, pstats, io
sys.path.insert(0, '/root/repo')
import torch, bench
c = dict(bench.CONFIGS["c2"])
model, sd, dd = bench.build(c, "bf16", torch.device("cuda"), seed=1234)
dd = {k: v.cuda() for k, v in dd.items()}
model.train()
for m in model.modules():
    if hasattr(m, "dropout_p"): m.dropout_p = 0.0
def step():
    model.zero_grad(set_to_none=True)
    out = model(dict(dd))
    bench.loss_fn(out, c["heads"]).backward()
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:4500])
