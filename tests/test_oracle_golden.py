"""CPU: the oracle (oracle/pq3d_oracle.py) against every golden fixture generated from the reference."""
import numpy as np
import pytest
import torch

from oracle import pq3d_oracle as O
from pq3d_amd import synth
from tests import util

TOL = dict(atol=1e-5, rtol=1e-5)
MODEL_FIXTURES = util.model_fixtures()


@pytest.mark.parametrize("name", MODEL_FIXTURES)
def test_model_fixture(name):
    z, args = util.load_fixture(name)
    _cfg, _model, sd, dd = util.model_case(args)
    assert abs(synth.state_checksum(sd) - float(z["meta/weights_checksum"])) < 1e-6 * float(z["meta/weights_checksum"]), \
        "synthetic weight generator drifted from the one that produced the fixtures"
    grads = any(k.startswith("grad/") for k in z.files)
    out, collect, loss, g = util.run_oracle(args, sd, dd, grads=grads)
    for i, q in enumerate(collect):
        util.check_against(z, f"layer_query/{i}", q, **TOL)
    if "ground" in args["heads"]:
        util.check_against(z, "ground_logits", out["ground_logits"], **TOL)
    if "mask" in args["heads"]:
        assert len(out["predictions_mask"]) == sum(1 for k in z.files if k.startswith("pred_mask/") and k.endswith("/sum"))
        for i, (c, m) in enumerate(zip(out["predictions_class"], out["predictions_mask"])):
            util.check_against(z, f"pred_class/{i}", c, **TOL)
            util.check_against(z, f"pred_mask/{i}", m, atol=1e-4, rtol=1e-5)
    assert abs(loss.item() - float(z["loss"])) <= 1e-5 * max(1.0, abs(float(z["loss"])))
    if grads:
        names = sorted(k[5:-4] for k in z.files if k.startswith("grad/") and k.endswith("/sum"))
        assert names == sorted(g.keys()), "oracle parameter-gradient set differs from the reference's"
        for n in names:
            util.check_against(z, "grad/" + n, g[n], atol=2e-6, rtol=2e-5, cap=util.MAX_GRAD)


@pytest.mark.parametrize("name", util.fixtures("F3_"))
def test_encoder_structures(name):
    z, a = util.load_fixture(name)
    d, B, Ns, Nq, T = a["d"], a["B"], a["Ns"], a["Nq"], a["T"]
    from pq3d_amd.modules import QueryMaskEncoder
    enc = QueryMaskEncoder(None, memories=a["memories"], hidden_size=d, num_attention_heads=a["H"],
                           num_layers=a["L"], spatial_selfattn=a["spatial"], structure=a["structure"])
    sd = synth.fill_module(enc, a["seed"])
    assert abs(synth.state_checksum(sd) - float(z["meta/weights_checksum"])) < 1e-6 * float(z["meta/weights_checksum"])
    r = np.random.default_rng(a["data_seed"])
    t = lambda *s: torch.from_numpy(r.standard_normal(s).astype(np.float32))
    dd = synth.synth_data_dict(B, Ns, Nq, {m: d for m in a["memories"]}, seed=a["data_seed"], memories=a["memories"],
                               prompt_len=T, d_model=d)
    qpos, fpos = t(B, Nq, d), t(B, Ns, d)
    util.check_against(z, "qpos", qpos, atol=0, rtol=0)
    input_dict = {"query": (torch.zeros(B, Nq, d), dd["query_pad_masks"].logical_not(), qpos)}
    for m in a["memories"]:
        if m == "prompt":
            input_dict[m] = [dd["prompt_feat"], dd["prompt_pad_masks"].logical_not(), None]
        else:
            input_dict[m] = [dd[f"{m}_seg_fts"], dd[f"{m}_seg_pad_masks"].logical_not(), fpos]
    pl = O.calc_pairwise_locs(dd["query_locs"]) if a["spatial"] else None
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    query, _, _ = O.query_mask_encoder(sdo, "", input_dict, pl, None, memories=a["memories"], H=a["H"],
                                       num_layers=a["L"], structure=a["structure"], spatial_selfattn=a["spatial"])
    util.check_against(z, "query", query, **TOL)
    (query * util.loss_weight("query", query.shape)).mean().backward()
    for k, v in sdo.items():
        n = k
        if f"grad/{n}/sum" in z.files:
            util.check_against(z, "grad/" + n, v.grad, atol=2e-6, rtol=2e-5, cap=util.MAX_GRAD)
        else:
            assert v.grad is None or float(v.grad.abs().max()) == 0.0, n


def test_misc_fixture():
    z, _ = util.load_fixture("F6_misc")
    locs, cmin, cmax = (torch.from_numpy(z[k]) for k in ("locs", "cmin", "cmax"))
    util.check_against(z, "pairwise_locs", O.calc_pairwise_locs(locs), atol=1e-6, rtol=1e-6)
    from pq3d_amd.modules import CoordinateEncoder
    sd = synth.fill_module(CoordinateEncoder(64), 3)
    util.check_against(z, "coord_enc", O.coordinate_encoder(sd, "", locs, cmin, cmax), **TOL)


def _scatter_mean_literal(src, idx, dim_size):
    """The published torch_scatter definition as a literal per-segment loop (sum, count clamped to >= 1, divide)."""
    out = torch.zeros(dim_size, *src.shape[1:], dtype=torch.float64)
    cnt = torch.zeros(dim_size, dtype=torch.float64)
    for r in range(src.shape[0]):
        out[int(idx[r])] += src[r].double()
        cnt[int(idx[r])] += 1
    cnt[cnt < 1] = 1
    return (out / cnt.view(-1, *([1] * (src.dim() - 1)))).float()


SCATTER_CASES = {
    "random unsorted ids with gaps": lambda r: (r.standard_normal((500, 7)), r.integers(0, 40, 500), 48),
    "every second segment empty": lambda r: (r.standard_normal((300, 5)), 2 * r.integers(0, 20, 300), 41),
    "one segment takes all": lambda r: (r.standard_normal((64, 3)), np.full(64, 9), 12),
    "ids sorted descending, many duplicates": lambda r: (r.standard_normal((200, 4)), np.sort(r.integers(0, 6, 200))[::-1].copy(), 6),
    "dim_size far beyond the largest id": lambda r: (r.standard_normal((10, 2)), r.integers(0, 3, 10), 100),
    "empty source": lambda r: (np.zeros((0, 6)), np.zeros((0,), dtype=np.int64), 5),
    "single row": lambda r: (r.standard_normal((1, 8)), np.array([3]), 4),
}


@pytest.mark.parametrize("case", sorted(SCATTER_CASES))
def test_scatter_mean_matches_definition(case):
    """Row (a)15: torch_scatter is un-vendored (parity unpinned by execution); the oracle restates its published definition
    and is checked here against a literal loop on adversarial inputs (see O.scatter_mean's docstring for the clauses)."""
    src, idx, n = SCATTER_CASES[case](np.random.default_rng(0))
    src, idx = torch.from_numpy(np.asarray(src, dtype=np.float32)), torch.from_numpy(np.asarray(idx, dtype=np.int64))
    out = O.scatter_mean(src, idx, n)
    assert out.shape == (n, src.shape[1])
    assert torch.allclose(out, _scatter_mean_literal(src, idx, n), atol=1e-6)
    empty = torch.ones(n, dtype=torch.bool)
    empty[idx] = False
    assert float(out[empty].abs().sum()) == 0.0      # clause 2: empty segments are exactly zero


def test_pooling_transpose_parents_semantics():
    """Row (f)2: MinkowskiPoolingTranspose(kernel 2, stride 2) coordinate map (MinkowskiEngine not installed: unpinned by
    execution): floor division also for negative coordinates, one ancestor per fine voxel, duplicates share it, batch items
    never mix, chained stride-2 maps compose to the stride-2^k map."""
    fine = torch.tensor([[0, -1, -1, -1], [0, -2, -2, -2], [0, 0, 1, 1], [0, 1, 0, 0], [1, 0, 1, 1], [0, 3, 3, 3], [0, -3, 5, 0]])
    cc, par = O.pooling_transpose_parents(fine, 2)
    assert cc.tolist() == [[0, -2, -2, -2], [0, 0, 0, 0], [1, 0, 0, 0], [0, 2, 2, 2], [0, -4, 4, 0]]
    assert par.tolist() == [0, 0, 1, 1, 2, 3, 4]              # -1 -> -2 (floor), duplicates share, batch 1 is its own cell
    g = torch.Generator().manual_seed(1)
    f2 = torch.cat([torch.randint(0, 2, (400, 1), generator=g), torch.randint(-33, 33, (400, 3), generator=g)], 1).unique(dim=0)
    c1, p1 = O.pooling_transpose_parents(f2, 2)
    c2, p2 = O.pooling_transpose_parents(c1, 4)
    c4, p4 = O.pooling_transpose_parents(f2, 4)
    assert torch.equal(c2[p2[p1]], c4[p4])                    # composition of the chain == the direct stride-4 map
    feat = torch.arange(c2.shape[0], dtype=torch.float32)[:, None]
    seg = torch.randint(0, 5, (f2.shape[0],), generator=g)
    up = feat[p2][p1]                                          # two transposed poolings materialised
    assert torch.equal(O.multiscale_segment_pool(feat, p2[p1], seg, 7), O.scatter_mean(up, seg, 7))


def test_t5_input_proj_against_reference_head():
    """F8: the in-repo part of the generation head (generation_head.py:16,22); the HF body is third-party."""
    z, a = util.load_fixture("F8_t5_head")
    from pq3d_amd import synth
    import ast
    names = {"input_proj.0.weight": (a["hf_config"]["d_model"], a["d"]), "input_proj.0.bias": (a["hf_config"]["d_model"],),
             "input_proj.1.weight": (a["hf_config"]["d_model"],), "input_proj.1.bias": (a["hf_config"]["d_model"],)}
    sd = synth.synth_state_dict(names, a["seed"])
    y = O.t5_input_proj({"generation_head." + k: v for k, v in sd.items()}, "generation_head.", torch.from_numpy(z["q"]))
    util.check_against(z, "input_proj", y, atol=1e-5, rtol=1e-5)


def test_multiscale_voxel_encoder_fixture():
    """F13: per-layer voxel scale select (query_encoder.py:90-91), num_blocks 3, self-mask, mask head on the last scale."""
    from tests import encoder_cases as E
    z, a = util.load_fixture("F13_multiscale")
    _enc, _mh, sd = E.f13_state(a)
    assert abs(synth.state_checksum(sd) - float(z["meta/weights_checksum"])) < 1e-6 * float(z["meta/weights_checksum"])
    query, pcls, pmask, loss, g, gin = E.f13_oracle(a, sd)
    util.check_against(z, "query", query, **TOL)
    assert len(pcls) == a["L"] * a["nb"] + 1
    for i, (c, m) in enumerate(zip(pcls, pmask)):
        util.check_against(z, f"pred_class/{i}", c, **TOL)
        util.check_against(z, f"pred_mask/{i}", m, atol=1e-4, rtol=1e-5)
    assert abs(loss.item() - float(z["loss"])) <= 1e-5 * max(1.0, abs(float(z["loss"])))
    names = sorted(k[5:-4] for k in z.files if k.startswith("grad/") and k.endswith("/sum"))
    assert names == sorted(g.keys())
    for n in names:
        util.check_against(z, "grad/" + n, g[n], atol=2e-6, rtol=2e-5, cap=util.MAX_GRAD)
    for k, v in gin.items():
        util.check_against(z, "grad_in/" + k, v, atol=2e-6, rtol=2e-5, cap=util.MAX_GRAD)


def test_memory_dropout_fixture():
    """F14: training-time memory dropout (query_encoder.py:145-151) with the reference's draws fixed from outside."""
    from tests import encoder_cases as E
    z, a = util.load_fixture("F14_memory_dropout")
    _enc, sd = E.f14_module(a)
    assert abs(synth.state_checksum(sd) - float(z["meta/weights_checksum"])) < 1e-6 * float(z["meta/weights_checksum"])
    keep = E.f14_keep(a)
    assert bool((keep.sum(-1) == 0).any()) and bool((keep.sum(-1) == 1).any() | (keep.sum(-1) == 2).any())
    query, g = E.f14_oracle(a, sd)
    util.check_against(z, "query", query, **TOL)
    names = sorted(k[5:-4] for k in z.files if k.startswith("grad/") and k.endswith("/sum"))
    assert names == sorted(g.keys())
    for n in names:
        util.check_against(z, "grad/" + n, g[n], atol=2e-6, rtol=2e-5, cap=util.MAX_GRAD)


def test_stage2_mixed_prompt_fixture():
    """F17: the stage-2 shipped decoder configuration (memories [mv, pc, voxel, prompt], structure 'mixed', memory_dropout
    0.6 with the reference's draws fixed from outside, T = 32 prompt tokens, GroundHead on the final query)."""
    from tests import encoder_cases as E
    z, a = util.load_fixture("F17_stage2_mixed_prompt")
    _enc, _gh, sd = E.f17_modules(a)
    assert abs(synth.state_checksum(sd) - float(z["meta/weights_checksum"])) < 1e-6 * abs(float(z["meta/weights_checksum"]))
    query, logits, loss, g, gin = E.f17_oracle(a, sd)
    util.check_against(z, "query", query, **TOL)
    util.check_against(z, "ground_logits", logits, **TOL)
    assert abs(float(loss) - float(z["loss"])) <= 1e-6 * max(1.0, abs(float(z["loss"])))
    names = sorted(k[5:-4] for k in z.files if k.startswith("grad/") and k.endswith("/sum"))
    assert names == sorted(g.keys())
    for n in names:
        util.check_against(z, "grad/" + n, g[n], atol=2e-6, rtol=2e-5, cap=util.MAX_GRAD)
    for k, v in gin.items():
        util.check_against(z, "grad_in/" + k, v, atol=2e-6, rtol=2e-5, cap=util.MAX_GRAD)
