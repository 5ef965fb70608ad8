"""CPU: initialisation parity.  The reference initialises the decoder with per-sublayer
xavier_uniform (``_reset_parameters``), deep-copies the layer (``layer_repeat``, modules/utils.py:28-32) and then applies
``_init_weights_bert`` (modules/weights.py:3-20; query_encoder.py:61): every nn.Linear ~ N(0, 0.02) with zero bias,
LayerNorm (1, 0), the packed MHA ``in_proj_weight`` (a bare Parameter) keeps xavier-uniform (and is therefore the one
matrix the L layers / M cross-attention copies still share after the re-draw).  Fixture F16 holds the reference's own initial state_dicts under fixed torch seeds;
the HIP modules construct in the same order, so the same seed must reproduce them bit for bit."""
import math

import numpy as np
import pytest
import torch

from pq3d_amd import modules as M
from tests import util
from tests.golden.make_golden import INIT_CASES, MAX_INIT


@pytest.mark.parametrize("case", sorted(INIT_CASES))
def test_same_seed_reproduces_reference_initial_state(case):
    z, _ = util.load_fixture("F16_init")
    _mod, cls, seed, a, kw = INIT_CASES[case]
    torch.manual_seed(seed)
    m = getattr(M, cls)(None, *a, **kw)
    sd = {k: v for k, v in m.state_dict().items()}
    assert sorted(sd) == [str(k) for k in z[f"{case}/keys"]]
    for k, v in sd.items():
        c = util.compress(v, MAX_INIT)
        assert tuple(z[f"{case}/{k}/shape"]) == c["shape"], k
        assert np.array_equal(z[f"{case}/{k}/sample"], c["sample"]), f"{case}/{k}: initial values differ from the reference's"
        assert abs(float(z[f"{case}/{k}/l2"]) - c["l2"]) <= 1e-6 * max(1.0, c["l2"]), k


def test_init_statistics_and_identical_copies():
    """The properties row 13 names, checked directly on a config-2-sized encoder (independent of the fixture)."""
    torch.manual_seed(0)
    d, H, L, mem = 256, 8, 4, ["voxel", "mv", "pc"]
    enc = M.QueryMaskEncoder(None, memories=mem, hidden_size=d, num_attention_heads=H, num_layers=L, spatial_selfattn=True,
                             structure="parallel")
    sd = enc.state_dict()
    for k, v in sd.items():
        if k.endswith("in_proj_weight"):          # bare Parameter: untouched by _init_weights_bert -> xavier_uniform
            bound = math.sqrt(6.0 / (v.shape[0] + v.shape[1]))
            assert float(v.abs().max()) <= bound and abs(float(v.std()) - bound / math.sqrt(3)) < 0.02 * bound, k
        elif k.endswith("in_proj_bias") or (k.endswith(".bias")):
            assert float(v.abs().max()) == 0.0, k
        elif ".norm." in k and k.endswith("weight"):
            assert bool((v == 1).all()), k
        elif k.endswith(".weight") and v.ndim == 2 and v.numel() >= 4096:
            assert abs(float(v.std()) - 0.02) < 0.002 and abs(float(v.mean())) < 0.002, (k, float(v.std()))
    # layer_repeat deep-copies ONE initialised layer, but QueryMaskEncoder then runs apply(_init_weights_bert) over the
    # copies (query_encoder.py:60-61): every nn.Linear is re-drawn independently, so what the L layers / M cross-attention
    # copies still share at start is exactly what _init_weights_bert does not touch -- the packed in_proj_weight (and the
    # constant biases / LayerNorms).  (SURVEY row 13 says "all layers start identical"; the reference's own initial
    # state_dict in F16 shows the Linear weights differ.)
    for k, v in sd.items():
        if not k.startswith("unified_encoder.0."):
            continue
        for i in range(1, L):
            other = sd[k.replace("unified_encoder.0.", f"unified_encoder.{i}.", 1)]
            if k.endswith("in_proj_weight") or k.endswith("bias") or ".norm." in k:
                assert torch.equal(v, other), k
            else:
                assert not torch.equal(v, other), k
        if ".cross_attn_list.0." in k and k.endswith("in_proj_weight"):
            for j in range(1, len(mem)):
                assert torch.equal(v, sd[k.replace(".cross_attn_list.0.", f".cross_attn_list.{j}.")]), k
    mh = M.MaskHeadSegLevel(None, d, 201, memories_for_match=mem, filter_out_classes=[0, 2])
    msd = mh.state_dict()   # MaskHeadSegLevel does not re-initialise: its layer_repeat copies stay identical
    for k, v in msd.items():
        if k.startswith("mask_pred_list.0."):
            for j in range(1, 3):
                assert torch.equal(v, msd[k.replace("mask_pred_list.0.", f"mask_pred_list.{j}.")]), k


def test_zero_rate_model_does_not_advance_the_dropout_rng():
    """Host logic (no kernels): the nn.Dropout inside a head's class MLP (kept for module-tree parity with utils.py:17-26) is a
    placeholder -- the applied rate is the head's dropout_p.  With every applied rate at zero a training forward must not advance
    the device RNG (two launches per step otherwise); with a non-zero rate it must."""
    import torch
    from pq3d_amd import modules as M, ops
    head = M.MaskHeadSegLevel(None, 32, 7, memories_for_match=["voxel"], dropout=0.1)
    head.train()
    dev = torch.device("cpu")
    rng = ops.drop_rng(dev)
    M.begin_dropout_step(head, dev)          # rate 0.1: a fresh epoch for this call
    e0 = rng.epoch
    head.__dict__.pop("_pq3d_drop_cache", None)
    head.dropout_p = 0.0                      # what bench.py / set_dropout do; cls_head[3].p stays 0.1
    M.begin_dropout_step(head, dev)
    M.begin_dropout_step(head, dev)
    assert rng.epoch == e0
    head.dropout_p = 0.1
    head._drop_epoch = rng.epoch              # (a second call within the same epoch)
    M.begin_dropout_step(head, dev)
    assert rng.epoch == e0 + 1
    M.set_dropout(head, 0.0)
    assert all(sub.p == 0.0 for sub in head.cls_head if isinstance(sub, torch.nn.Dropout))
