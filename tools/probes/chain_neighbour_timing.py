"""Probe (round 6): how long does a model step take beside a CU-holding neighbour (pq3d_test_occupy_cus), per structure / row count /
neighbour size -- does a chain group have to WAIT for the neighbour to end?  usage: python tools/probes/chain_neighbour_timing.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests import util
from pq3d_amd import _lib as L, fused, ops
from pq3d_amd.modules import set_compute

dev = torch.device("cuda")
for case, B in (("plain", 8), ("plain", 16), ("mask_head", 8), ("mask_head", 16), ("plain", 8), ("mask_head", 8)):
    if case == "mask_head":
        args = dict(B=B // 2, Ns=512, Nq=200, d=256, H=8, L=2, memories=["voxel", "mv", "pc"], heads=["mask"], spatial=True,
                    structure="parallel", use_self_mask=True, C=201, foc=(0, 2), seed=0, data_seed=1234)
    else:
        args = dict(B=B, Ns=256, Nq=100, d=256, H=8, L=2, memories=["voxel", "mv", "pc"], heads=[], spatial=True,
                    structure="parallel", seed=0, data_seed=1234)
    _cfg, model, _sd, dd = util.model_case(args)
    set_compute(model, "bf16"); model.unified_encoder.fused = True; model.to(dev)
    ddv = {k: v.to(dev) for k, v in dd.items()}

    def step():
        model.zero_grad()
        out = model(dict(ddv))
        loss = out["query_embeds"].float().square().mean()      # device-only loss (util.synthetic_loss builds weights on the host: 100+ ms)
        for m in out.get("predictions_mask", []):
            loss = loss + m.float().clamp(min=-50.0).mean()
        loss.backward()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step(); torch.cuda.synchronize(); alone = (time.perf_counter() - t0) * 1e3
    for held, lds_kb in ((32, 100), (64, 8), (96, 100)):
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            L.check(L.lib().pq3d_test_occupy_cus(held, lds_kb * 1024, 100000, L.stream()), "occ")
            ev = torch.cuda.Event(); ev.record()
        t0 = time.perf_counter(); step(); torch.cuda.current_stream().synchronize(); t = (time.perf_counter() - t0) * 1e3
        done = ev.query()
        torch.cuda.synchronize()
        print(f"{case:10s} R={B * 100:5d} neighbour {held:3d} x {lds_kb:3d} KB for 100 ms: step {t:7.1f} ms (alone {alone:5.1f}), neighbour finished first: {done}, "
              f"chain_error {ops.chain_error(dev)}", flush=True)
