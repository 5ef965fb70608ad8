"""CPU: the oracle's restatement of the reference's non-shipped variants (pre-norm sublayers, spatial_attn_fusion='bias',
GroundHeadV1) against fixture F21, generated from the reference by tests/golden/make_golden.py::run_variants_case."""
import ast

import torch

from oracle import pq3d_oracle as O
from pq3d_amd import modules as M
from pq3d_amd import synth
from tests import util, variants_case as VC


def test_prenorm_layers_match_the_reference():
    z, _ = util.load_fixture("F21_variants")
    a = ast.literal_eval(str(z["meta/args"]))
    v = synth.variant_inputs(d=a["d"])
    for tag, spatial, structure in VC.CASES:
        _layer, sd = VC.layer_modules(a["d"], a["H"], spatial, structure, a["seed"])
        assert abs(synth.state_checksum(sd) - float(z[f"meta/{tag}/weights_checksum"])) < 1e-6 * abs(float(z[f"meta/{tag}/weights_checksum"]))
        y, gq, g = VC.oracle_layer(sd, v, a["H"], spatial, structure)
        util.check_against(z, f"{tag}/out", y, atol=1e-5, rtol=1e-5)
        util.check_against(z, f"{tag}/grad_in/query", gq, atol=2e-6, rtol=2e-5)
        names = sorted(k[len(tag) + 6:-4] for k in z.files if k.startswith(f"{tag}/grad/") and k.endswith("/sum"))
        assert names == sorted(g), (tag, set(names) ^ set(g))
        for n in names:
            util.check_against(z, f"{tag}/grad/{n}", g[n], atol=2e-6, rtol=2e-5, cap=util.MAX_GRAD)


def test_bias_fusion_and_ground_head_v1_match_the_reference():
    z, _ = util.load_fixture("F21_variants")
    a = ast.literal_eval(str(z["meta/args"]))
    d, H, seed = a["d"], a["H"], a["seed"]
    v = synth.variant_inputs(d=d)
    msa = M.MultiHeadAttentionSpatial(d, H, dropout=0.0, spatial_attn_fusion="bias")
    sd = synth.fill_module(msa, seed + 1)
    assert abs(synth.state_checksum(sd) - float(z["meta/msa_bias/weights_checksum"])) < 1e-6 * abs(float(z["meta/msa_bias/weights_checksum"]))
    x = v["query"] + v["qpos"]
    y = O.spatial_mha(sd, "", x, x, v["query"], O.calc_pairwise_locs(v["centers"]), H, key_padding_mask=v["qpad"], fusion="bias")
    util.check_against(z, "msa_bias/out", y, atol=1e-5, rtol=1e-5)
    gh = M.GroundHeadV1(None, input_size=d, hidden_size=d, sem_cls_size=37, dropout=0.3)
    sd = synth.fill_module(gh, seed + 2)
    assert abs(synth.state_checksum(sd) - float(z["meta/ghv1/weights_checksum"])) < 1e-6 * abs(float(z["meta/ghv1/weights_checksum"]))
    outs = O.ground_head_v1(sd, "", v["txt"], v["query"], v["pre"], v["qpad"].logical_not())
    for k, t in zip(("txt", "obj", "pre", "og3d"), outs):
        util.check_against(z, "ghv1/" + k, t, atol=1e-5, rtol=1e-5)
