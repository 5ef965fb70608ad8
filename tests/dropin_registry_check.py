#!/usr/bin/env python3
"""Executed by tests/test_dropin_registry.py in a process of its own (build container only: imports /root/reference).

INTEGRATION.md section 1, executed: the HIP modules are registered into the REFERENCE'S OWN registries
(modules/build.py:6-9) under the names a maintainer's YAML would select, and the REFERENCE'S Query3DUnified
(model/query3d_unified.py:31-78) is built around them through its own build_module_by_name (modules/build.py:24-31) and
MODEL_REGISTRY (model/build.py:6,17-19).  Checked: the classes that were built, load_state_dict(strict=True) of the
all-reference model's checkpoint, get_opt_params() (its "Some parameters are not optimized!" assertion, :224-238), the
.spatial_selfattn attribute the model reads (:182), and that a CPU batch fails loudly (no CPU fallback behind the HIP
modules).  Nothing of the reference is copied: it is imported where it lies."""
import copy
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as G  # noqa: E402  (the SURVEY 8c import recipe + config / input generators)

ref = G.import_reference()
from model.build import MODEL_REGISTRY, build_model  # noqa: E402  (the reference's)
from modules.build import GROUNDING_REGISTRY, HEADS_REGISTRY, VISION_REGISTRY  # noqa: E402  (the reference's)

import pq3d_amd.modules as hip  # noqa: E402
from pq3d_amd import _lib, synth  # noqa: E402


# ---- what INTEGRATION.md section 1 tells a maintainer to add (one file per registry) ---------------------------------
@GROUNDING_REGISTRY.register()
class QueryMaskEncoderHIP(hip.QueryMaskEncoder):      # query_encoder.py:53-54
    pass


@HEADS_REGISTRY.register()
class MaskHeadSegLevelHIP(hip.MaskHeadSegLevel):      # mask_head.py:12
    pass


@HEADS_REGISTRY.register()
class GroundHeadHIP(hip.GroundHead):                  # grounding_head.py:44
    pass


@VISION_REGISTRY.register()
class ObjectEncoderHIP(hip.ObjectEncoder):            # object_encoder.py:16
    pass


SWAP = {"QueryMaskEncoder": "QueryMaskEncoderHIP", "MaskHeadSegLevel": "MaskHeadSegLevelHIP", "GroundHead": "GroundHeadHIP",
        "ObjectEncoder": "ObjectEncoderHIP"}


def swapped(cfg):
    cfg = copy.deepcopy(cfg)
    for k, v in cfg["model"].items():
        if isinstance(v, dict) and v.get("name") in SWAP:
            v["name"] = SWAP[v["name"]]        # the one YAML string per module a maintainer changes
    return cfg


def check(tag, **kw):
    mem, heads = kw.pop("memories"), kw.pop("heads")
    d = kw.pop("d")
    cfg_ref = G.model_cfg(d, kw.pop("H"), kw.pop("L"), mem, heads, {m: d for m in mem}, **kw)
    cfg_hip = swapped(cfg_ref)
    assert MODEL_REGISTRY.get(cfg_hip.model.name) is ref.model.Query3DUnified
    torch.manual_seed(0)
    m_ref = build_model(cfg_ref)                     # the reference's model around the reference's modules
    m_hip = build_model(cfg_hip)                     # the reference's model around the HIP modules
    assert type(m_hip) is ref.model.Query3DUnified
    assert isinstance(m_hip.unified_encoder, hip.QueryMaskEncoder) and type(m_hip.unified_encoder).__name__ == "QueryMaskEncoderHIP"
    for m in mem:
        if m != "prompt":
            assert isinstance(getattr(m_hip, m + "_encoder"), hip.ObjectEncoder)
    if "mask" in heads:
        assert isinstance(m_hip.mask_head, hip.MaskHeadSegLevel)
    if "ground" in heads:
        assert isinstance(m_hip.ground_head, hip.GroundHead)
    sd = synth.fill_module(m_ref, 3)
    missing, unexpected = m_hip.load_state_dict(m_ref.state_dict(), strict=True)
    assert not missing and not unexpected
    for k, v in m_hip.state_dict().items():
        assert torch.equal(v, sd[k]), k
    assert sorted(m_hip.state_dict()) == sorted(m_ref.state_dict())
    groups = m_hip.get_opt_params()                  # asserts "Some parameters are not optimized!" inside (:236-237)
    n_opt = sum(len(g["params"]) for g in groups)
    assert n_opt == len(list(m_hip.parameters())) == len(list(m_ref.parameters()))
    ref_groups = m_ref.get_opt_params()
    assert [(len(g["params"]), g["weight_decay"], g["lr"]) for g in groups] == \
        [(len(g["params"]), g["weight_decay"], g["lr"]) for g in ref_groups]
    assert m_hip.unified_encoder.spatial_selfattn == kw["spatial"]          # read by the model at :182
    # the product path has no CPU fallback: a host batch through the reference's forward fails loudly in the HIP modules
    dd = synth.synth_data_dict(2, 24, 6, {m: d for m in mem}, seed=1, memories=mem, loc_dim=kw.get("dim_loc", 3))
    dd["tgt_object_id"] = torch.zeros(2, dtype=torch.long)
    try:
        m_hip.eval()(dict(dd))
    except _lib.Pq3dError:
        pass
    else:
        raise AssertionError("a CPU batch went through the HIP modules without an error")
    print(f"{tag}: ok ({n_opt} parameters in {len(groups)} groups)")


check("stage-1 like (parallel, self-mask, mask head)", d=64, H=4, L=2, memories=["voxel", "mv", "pc"], heads=["mask"],
      spatial=True, structure="parallel", use_self_mask=True, num_blocks=2, C=21, foc=(0, 2))
check("stage-2 like (mixed, ground head, 6-D locations)", d=48, H=4, L=2, memories=["mv", "pc", "voxel"], heads=["ground"],
      spatial=True, structure="sequential", dim_loc=6)
check("plumbing (c1: non-spatial, one stream)", d=64, H=4, L=1, memories=["voxel"], heads=[], spatial=False,
      structure="sequential")
print("DROPIN-OK")
