"""GPU: the HIP modules' non-shipped variants (pre-norm sublayers on the modular path -- residual + dropout in the last GEMM's
epilogue --, spatial_attn_fusion='bias', GroundHeadV1) against fixture F21 from the reference (VERDICT r5 'missing' item 5)."""
import ast

import pytest
import torch

from pq3d_amd import modules as M
from pq3d_amd import synth
from tests import util, variants_case as VC

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("tag,spatial,structure", VC.CASES)
def test_prenorm_layer_fp32_matches_the_reference(tag, spatial, structure):
    z, _ = util.load_fixture("F21_variants")
    a = ast.literal_eval(str(z["meta/args"]))
    v = synth.variant_inputs(d=a["d"])
    layer, _sd = VC.layer_modules(a["d"], a["H"], spatial, structure, a["seed"], "fp32")
    y, gq, g = VC.hip_layer(layer, v, spatial)
    util.check_against(z, f"{tag}/out", y, atol=1e-5, rtol=1e-5)
    util.check_against(z, f"{tag}/grad_in/query", gq, atol=5e-6, rtol=3e-4)
    for n, t in g.items():
        util.check_against(z, f"{tag}/grad/{n}", t, atol=1e-5, rtol=3e-3 if "pairwise_loc_fc" in n else 3e-4, cap=util.MAX_GRAD)


@pytest.mark.parametrize("tag,spatial,structure", VC.CASES)
def test_prenorm_layer_bf16_modes_close_to_fp32(tag, spatial, structure):
    z, _ = util.load_fixture("F21_variants")
    a = ast.literal_eval(str(z["meta/args"]))
    v = synth.variant_inputs(d=a["d"])
    ref = None
    for compute, bar in (("fp32", 0.0), ("bf16x3", 1e-3), ("bf16", 2e-2)):
        layer, _sd = VC.layer_modules(a["d"], a["H"], spatial, structure, a["seed"], compute)
        y, _gq, _g = VC.hip_layer(layer, v, spatial)
        if ref is None:
            ref = y.float()
        else:
            assert float((y.float() - ref).abs().max()) <= bar * float(ref.abs().max()), compute


def test_bias_fusion_and_ground_head_v1():
    z, _ = util.load_fixture("F21_variants")
    a = ast.literal_eval(str(z["meta/args"]))
    d, H, seed = a["d"], a["H"], a["seed"]
    v = synth.variant_inputs(d=d)
    msa = M.MultiHeadAttentionSpatial(d, H, dropout=0.0, spatial_attn_fusion="bias")
    synth.fill_module(msa, seed + 1)
    M.set_compute(msa, "fp32")
    msa.to(DEV).eval()
    x = (v["query"] + v["qpos"]).to(DEV)
    y = msa(x, x, v["query"].to(DEV), M.calc_pairwise_locs(v["centers"].to(DEV)), key_padding_mask=v["qpad"].to(DEV))
    util.check_against(z, "msa_bias/out", y, atol=1e-5, rtol=1e-5)
    gh = M.GroundHeadV1(None, input_size=d, hidden_size=d, sem_cls_size=37, dropout=0.3)
    synth.fill_module(gh, seed + 2)
    M.set_compute(gh, "fp32")
    gh.to(DEV).eval()
    outs = gh(v["txt"].to(DEV), v["query"].to(DEV), v["pre"].to(DEV), v["qpad"].logical_not().to(DEV))
    for k, t in zip(("txt", "obj", "pre", "og3d"), outs):
        util.check_against(z, "ghv1/" + k, t, atol=1e-5, rtol=1e-5)
    with pytest.raises(NotImplementedError):
        M.MultiHeadAttentionSpatial(d, H, spatial_attn_fusion="ctx")
