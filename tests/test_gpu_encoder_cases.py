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


def _pyramid(B=2, S=40, seed=3):
    """A synthetic 5-level voxel pyramid with the reference's channel widths: coordinates -> stride-2 pooling chains."""
    from oracle import pq3d_oracle as O
    from pq3d_amd import ops
    g = torch.Generator().manual_seed(seed)
    planes = (256, 256, 128, 96, 96)
    feats = [[None] * B for _ in range(5)]
    parents = [[None] * B for _ in range(5)]
    p2s = []
    for b in range(B):
        fine = torch.cat([torch.zeros(900 + 300 * b, 1, dtype=torch.long),
                          torch.randint(-24, 24, (900 + 300 * b, 3), generator=g)], 1).unique(dim=0)
        N = fine.shape[0]
        p2s.append(torch.randint(0, S - 3 * b, (N,), generator=g))
        maps, cur = [], fine
        for lvl in range(1, 5):                       # strides 2, 4, 8, 16
            cc, par = O.pooling_transpose_parents(cur, 2 ** lvl)
            maps.append(par); cur = cc
        for h in range(5):                            # backbone level h sits (4 - h) poolings above full resolution
            npool = 4 - h
            par = ops.compose_parents(maps[:npool]) if npool else torch.arange(N)
            n_rows = int(par.max()) + 1
            feats[h][b] = torch.randn(n_rows, planes[h], generator=g)
            parents[h][b] = par
    return [(feats[h], parents[h]) for h in range(5)], p2s


@pytest.mark.parametrize("compute", ["fp32", "bf16"])
def test_pcd_seg_level_encoder_post_backbone_matches_oracle(compute):
    """PCDMask3DSegLevelEncoder after the backbone (pcd_mask3d_encoder.py:139-154): 5 levels -> 5 projected segment-feature
    tensors, the multi-scale list of the stage-1 decoder.  State-dict keys as the reference's feat_proj_list."""
    from oracle import pq3d_oracle as O
    from pq3d_amd import modules as M
    from pq3d_amd import synth
    enc = M.PCDMask3DSegLevelEncoder(None, None, hidden_size=64, hlevels=[0, 1, 2, 3], dropout=0.1)
    assert sorted(enc.state_dict()) == sorted(f"feat_proj_list.{i}.{j}.{w}" for i in range(5) for j in (0, 1) for w in ("weight", "bias"))
    sd = synth.fill_module(enc, 2)
    M.set_compute(enc, compute)
    enc.to("cuda").eval()
    pyr, p2s = _pyramid()
    S = 40
    ref = O.pcd_seg_level_encoder({k: v.double() for k, v in sd.items()},
                                  "", [([f.double() for f in fs], ps) for fs, ps in pyr], p2s, S)
    pyr_d = [([f.cuda().requires_grad_(True) for f in fs], [p.cuda() for p in ps]) for fs, ps in pyr]
    out = enc(pyr_d, [p.cuda() for p in p2s], S)
    assert len(out) == 5 and all(o.shape == (2, S, 64) for o in out)
    tol = 1e-5 if compute == "fp32" else 1e-3     # 'bf16' mode: split-bf16 projection (fp32-grade)
    for o, r in zip(out, ref):
        assert float((o.detach().double().cpu() - r).abs().max()) <= tol * float(r.abs().max())
    sum(o.square().mean() for o in out).backward()
    assert all(f.grad is not None and torch.isfinite(f.grad).all() for fs, _ in pyr_d for f in fs)


def test_pcd_seg_level_encoder_batched_form_equals_list_form():
    """One plan + one launch per level for the whole batch (concatenated features, ids offset by b * max_seg, parents
    indexing the concatenated coarse level) gives the list form's outputs bit for bit (same summation order per segment)."""
    from pq3d_amd import modules as M
    from pq3d_amd import synth
    enc = M.PCDMask3DSegLevelEncoder(None, None, hidden_size=64, hlevels=[0, 1, 2, 3], dropout=0.0)
    synth.fill_module(enc, 2)
    enc.to("cuda").eval()
    pyr, p2s = _pyramid()
    S = 40
    pyr_d = [([f.cuda() for f in fs], [p.cuda() for p in ps]) for fs, ps in pyr]
    ref = enc(pyr_d, [p.cuda() for p in p2s], S)
    ids = torch.cat([p.cuda() + b * S for b, p in enumerate(p2s)])
    pyr_b = []
    for fs, ps in pyr_d:
        offs = [0]
        for f in fs:
            offs.append(offs[-1] + f.shape[0])
        pyr_b.append((torch.cat(fs), torch.cat([torch.where(p >= 0, p + o, p) for p, o in zip(ps, offs)])))
    out = enc(pyr_b, ids, S, batch_size=len(p2s))
    for a, b in zip(out, ref):
        assert torch.equal(a, b)


def test_model_with_online_voxel_pyramid_takes_the_fused_path():
    """Query3DUnified with use_offline_voxel_fts = False: the voxel memory is the post-backbone encoder's multi-scale list;
    the fused executor runs it and agrees with the modular path."""
    from pq3d_amd import synth
    from pq3d_amd.model import Cfg, Query3DUnified, make_cfg
    cfg = make_cfg(d=64, H=4, L=4, memories=["voxel", "mv"], heads=["mask"], use_self_mask=True, num_blocks=2, C=21, foc=(0, 2))
    cfg.model["use_offline_voxel_fts"] = False
    cfg.model["voxel_encoder"] = Cfg(name="PCDMask3DSegLevelEncoder", args=Cfg(backbone_kwargs=None, hidden_size=64,
                                                                               hlevels=[0, 1, 2, 3], dropout=0.1))
    model = Query3DUnified(cfg, compute="fp32")
    synth.fill_module(model, 0)
    model.to("cuda").eval()
    pyr, p2s = _pyramid()
    dd = synth.synth_data_dict(2, 40, 9, {"mv": 64, "voxel": 64}, seed=5, memories=["mv"])
    dd = {k: v.cuda() for k, v in dd.items()}
    dd["voxel_pyramid"] = [([f.cuda() for f in fs], [p.cuda() for p in ps]) for fs, ps in pyr]
    dd["voxel2segment"] = [p.cuda() for p in p2s]
    res = []
    for fused in (True, False):
        model.unified_encoder.fused = fused
        (out, took) = _count_fused(lambda: model(dict(dd)))
        assert took == fused
        res.append(out)
    assert len(res[0]["predictions_mask"]) == 4 * 2 + 1
    for a, b in zip(res[0]["predictions_mask"], res[1]["predictions_mask"]):
        assert float((a - b).abs().max()) <= 1e-4 * float(b[b > -1e5].abs().max())
    assert float((res[0]["query_embeds"] - res[1]["query_embeds"]).abs().max()) <= 1e-4


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("compute", ["fp32", "bf16"])
def test_stage2_mixed_prompt_matches_reference_fixture(fused, compute):
    """F17: structure 'mixed' with a prompt memory + memory dropout (the stage-2 shipped configuration,
    configs/unified_tasks_sceneverse.yaml:113,159-165) on the fused executor and on the modular path, against the fixture
    made from the reference's QueryMaskEncoder + GroundHead."""
    z, a = util.load_fixture("F17_stage2_mixed_prompt")
    (query, logits, loss, g, gin), took = _count_fused(lambda: E.f17_hip(a, compute, fused))
    assert took == fused
    tol = dict(atol=1e-5, rtol=1e-5) if compute == "fp32" else dict(atol=5e-3, rtol=5e-3)
    util.check_against(z, "query", query, **tol)
    util.check_against(z, "ground_logits", logits, **(tol if compute == "fp32" else dict(atol=1e-2, rtol=1e-2)))
    if compute == "fp32":
        assert abs(float(loss) - float(z["loss"])) <= 2e-5 * max(1.0, abs(float(z["loss"])))
        names = sorted(k[5:-4] for k in z.files if k.startswith("grad/") and k.endswith("/sum"))
        assert names == sorted(g.keys())
        for n in names:
            util.check_against(z, "grad/" + n, g[n], atol=2e-6, rtol=3e-3 if "pairwise_loc_fc" in n else 3e-4, cap=util.MAX_GRAD)
        for k, v in gin.items():
            util.check_against(z, "grad_in/" + k, v, atol=2e-6, rtol=3e-4, cap=util.MAX_GRAD)
    else:
        # bf16 mode: gradients close to the fp32 reference's in relative L2 (the per-parameter bar of the model-level tests)
        import numpy as np
        for k, v in gin.items():
            ref_l2 = float(z[f"grad_in/{k}/l2"])
            assert abs(float(v.float().norm()) - ref_l2) <= 5e-2 * ref_l2, k


def test_stage2_mixed_prompt_fused_equals_modular_in_fp32():
    _z, a = util.load_fixture("F17_stage2_mixed_prompt")
    qf, lf, _, gf, ginf = E.f17_hip(a, "fp32", True)
    qm, lm, _, gm_, ginm = E.f17_hip(a, "fp32", False)
    assert float((qf - qm).abs().max()) <= 1e-6 * max(1.0, float(qm.abs().max()))
    fin = torch.isfinite(lm)
    assert torch.equal(fin, torch.isfinite(lf)) and float((lf[fin] - lm[fin]).abs().max()) <= 1e-5
    gmax = max(float(v.norm()) for v in gm_.values())
    for n in gm_:
        assert float((gf[n] - gm_[n]).norm()) <= 1e-4 * max(float(gm_[n].norm()), 1e-2 * gmax), n
    for k in ginm:
        assert float((ginf[k] - ginm[k]).norm()) <= 1e-4 * float(ginm[k].norm()), k
