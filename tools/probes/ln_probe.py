"""Probe: LayerNorm forward/backward launch times at the encoder (big-row) and query (R = 800) shapes, in a HIP graph."""
import sys, time
sys.path.insert(0, '/root/repo')
import torch
from pq3d_amd import ops, _lib as L
dev = 'cuda'


def graph_time(fn, n=20, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps / n * 1e6


for (R, G) in ((8192, 2), (8192, 1), (800, 1)):
    d = 256
    xs = [torch.randn(R, d, device=dev, requires_grad=True) for _ in range(G)]
    Ws = [torch.randn(d, d, device=dev, requires_grad=True) * 0.05 for _ in range(G)]
    bs = [torch.zeros(d, device=dev, requires_grad=True) for _ in range(G)]
    gs = [torch.ones(d, device=dev, requires_grad=True) for _ in range(G)]
    bt = [torch.zeros(d, device=dev, requires_grad=True) for _ in range(G)]
    lin = torch.randn(G, R, d, device=dev)
    ys = torch.empty(G, R, d, device=dev)
    mean = torch.empty(G, R, device=dev); rstd = torch.empty(G, R, device=dev)
    dsc = ops._ln_desc(None, [lin[g] for g in range(G)], gs, bt, None, 1e-5, R, None, mean, rstd)
    dsc.independent = 1; dsc.dt_y = L.F32
    for g in range(G): dsc.ys[g] = L.ptr(ys[g])
    import ctypes as C
    t_f = graph_time(lambda: L.check(L.lib().pq3d_add_ln_fwd(C.byref(dsc), L.stream()), "f"))
    dys = torch.randn(G, R, d, device=dev); dlin = torch.empty(G, R, d, device=dev)
    zb = torch.zeros(G * 2 * d, device=dev)
    dsc2 = ops._ln_desc(None, [lin[g] for g in range(G)], gs, bt, None, 1e-5, R, None, mean, rstd)
    dsc2.independent = 1; dsc2.accumulate = 1
    for g in range(G):
        dsc2.dys[g], dsc2.d_o[g], dsc2.dgamma[g], dsc2.dbeta[g] = L.ptr(dys[g]), L.ptr(dlin[g]), L.ptr(zb[g * d:(g + 1) * d]), L.ptr(zb[(G + g) * d:(G + g + 1) * d])
    t_b = graph_time(lambda: L.check(L.lib().pq3d_add_ln_bwd(C.byref(dsc2), L.stream()), "b"))
    gb_f, gb_b = 2 * G * R * d * 4 / 1e3, 3 * G * R * d * 4 / 1e3
    print(f"R={R} G={G}: fwd {t_f:6.1f} us ({gb_f / t_f / 1e3:.2f} TB/s)   bwd {t_b:6.1f} us ({gb_b / t_b / 1e3:.2f} TB/s)")
    # correctness of dgamma against torch
    zb.zero_()
    L.check(L.lib().pq3d_add_ln_fwd(C.byref(dsc), L.stream()), "f")
    L.check(L.lib().pq3d_add_ln_bwd(C.byref(dsc2), L.stream()), "b")
    x = lin[0].clone().requires_grad_(True); gam = gs[0].detach().clone().requires_grad_(True); bet = bt[0].detach().clone().requires_grad_(True)
    y = torch.nn.functional.layer_norm(x, (d,), gam, bet, 1e-5)
    y.backward(dys[0])
    print("   y err %.2e  dx err %.2e  dgamma relerr %.2e  dbeta relerr %.2e" % (
        float((ys[0] - y).abs().max()), float((dlin[0] - x.grad).abs().max()),
        float((zb[:d] - gam.grad).abs().max() / gam.grad.abs().max()), float((zb[G * d:(G + 1) * d] - bet.grad).abs().max() / bet.grad.abs().max())))
