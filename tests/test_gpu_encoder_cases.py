"""GPU: the shipped stage-1 / stage-2 decoder configurations on BOTH execution paths (fused executor and modular ops)
against reference-generated fixtures: F13 = multi-scale voxel memory (per-layer scale select, num_blocks 3, self-mask,
mask head on the last scale; configs/instseg_sceneverse.yaml:114,141-146), F14 = training-time memory dropout 0.6
(configs/unified_tasks_sceneverse.yaml:162) with the reference's draws fed to both sides."""
import pytest
import torch

import pq3d_amd.fused as F
from tests import encoder_cases as E
from tests import util

pytestmark = pytest.mark.gpu


def _count_fused(fn):
    calls = []
    orig = F._FusedDecoder.apply
    F._FusedDecoder.apply = staticmethod(lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    try:
        return fn(), bool(calls)
    finally:
        F._FusedDecoder.apply = orig


@pytest.mark.parametrize("fused", [True, False])
def test_multiscale_voxel_fp32_matches_reference_fixture(fused):
    z, a = util.load_fixture("F13_multiscale")
    (query, pcls, pmask, loss, g, gin), took = _count_fused(lambda: E.f13_hip(a, "fp32", fused))
    assert took == fused, "the fused executor must cover the multi-scale voxel list"
    util.check_against(z, "query", query, atol=1e-5, rtol=1e-5)
    assert len(pcls) == a["L"] * a["nb"] + 1
    for i, (c, m) in enumerate(zip(pcls, pmask)):
        util.check_against(z, f"pred_class/{i}", c, atol=1e-5, rtol=1e-5)
        util.check_against(z, f"pred_mask/{i}", m, atol=1e-4, rtol=1e-5)
    assert abs(loss.item() - float(z["loss"])) <= 2e-5 * max(1.0, abs(float(z["loss"])))
    names = sorted(k[5:-4] for k in z.files if k.startswith("grad/") and k.endswith("/sum"))
    assert names == sorted(g.keys())
    for n in names:
        util.check_against(z, "grad/" + n, g[n], atol=2e-6, rtol=3e-3 if "pairwise_loc_fc" in n else 3e-4, cap=util.MAX_GRAD)
    for k, v in gin.items():
        util.check_against(z, "grad_in/" + k, v, atol=2e-6, rtol=3e-4, cap=util.MAX_GRAD)


def test_multiscale_voxel_bf16_fused_close_to_fp32_reference():
    z, a = util.load_fixture("F13_multiscale")
    query, pcls, pmask, loss, g, gin = E.f13_hip(a, "bf16", True)
    _enc, _mh, sd = E.f13_state(a)
    rq, rc, rm, rl, rg, rgin = E.f13_oracle(a, sd)
    q, r = query.detach().float().cpu(), rq.detach()
    assert float((q - r).abs().max()) <= 2e-2 * float(r.abs().max())
    for m, r_ in zip(pmask, rm):
        flips = float(((m.detach().float().cpu() < 0) != (r_ < 0)).float().mean())
        assert flips <= 1e-2, flips


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("compute", ["fp32", "bf16"])
def test_memory_dropout_matches_reference_fixture(fused, compute):
    z, a = util.load_fixture("F14_memory_dropout")
    (query, g), took = _count_fused(lambda: E.f14_hip(a, compute, fused))
    assert took == fused
    tol = dict(atol=1e-5, rtol=1e-5) if compute == "fp32" else dict(atol=5e-3, rtol=5e-3)
    util.check_against(z, "query", query, **tol)
    if compute == "fp32":
        names = sorted(k[5:-4] for k in z.files if k.startswith("grad/") and k.endswith("/sum"))
        assert names == sorted(g.keys())
        for n in names:
            util.check_against(z, "grad/" + n, g[n], atol=2e-6, rtol=3e-3 if "pairwise_loc_fc" in n else 3e-4, cap=util.MAX_GRAD)


def test_memory_dropout_draws_a_new_mask_per_layer_application():
    """query_encoder.py:145: torch.rand is drawn inside every layer call -- the fused path must not reuse one mask."""
    _z, a = util.load_fixture("F14_memory_dropout")
    seen = []
    enc, _ = E.f14_module(a)
    enc.to("cuda").train()
    keep = E.f14_keep(a).cuda()
    enc.memory_keep_hook = lambda app, B, Mm, device: (seen.append(app), keep[app])[1]
    feats, pad, qpos, fpos, centers = E.f14_inputs(a)
    from pq3d_amd import modules as M
    B, Nq, d = qpos.shape
    idict = {"query": (torch.zeros(B, Nq, d, device="cuda"), torch.zeros(B, Nq, dtype=torch.bool, device="cuda"), qpos.cuda())}
    for m in a["memories"]:
        idict[m] = [feats[m].cuda(), pad.cuda(), fpos.cuda()]
    enc(idict, M.calc_pairwise_locs(centers.cuda()), None)
    assert seen == list(range(a["L"]))
