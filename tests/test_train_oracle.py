"""CPU: the train-step oracle (oracle/train_oracle.py: clip_grad_norm_ + AdamW + LambdaLR restated) against the F7
fixtures produced with the reference's own optimizer / scheduler objects (tests/golden/make_golden.py:run_train_case)."""
import numpy as np
import pytest
import torch

from oracle import pq3d_oracle as O
from oracle import train_oracle as T
from tests import util


def oracle_train(args, steps):
    _cfg, _model, sd, dd = util.model_case(args)
    params = {k: v.clone() for k, v in sd.items() if v.dtype.is_floating_point and not k.endswith("gauss_B")}
    frozen = {k: v for k, v in sd.items() if k not in params}
    lr_of = {n: (args["head_lr"] if (args.get("head_lr") and n.startswith("ground_head.")) else args["lr"])
             for n in params}
    st = T.AdamWState()
    init = {k: v.clone() for k, v in params.items()}
    hist = []
    for _ in range(steps):
        leaf = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        collect = []
        out = O.query3d_unified_forward({**leaf, **frozen}, util.oracle_cfg(args),
                                        {k: (v.clone() if torch.is_tensor(v) else v) for k, v in dd.items()},
                                        collect=collect)
        loss = util.synthetic_loss(out, args["heads"], collect[-1])
        loss.backward()
        grads = {k: v.grad for k, v in leaf.items() if v.grad is not None}
        lr, norm = T.adamw_step(params, grads, st, lr=args["lr"], grad_norm=args["grad_norm"],
                                warmup_steps=args["warmup_steps"], total_steps=args["total_steps"], lr_of=lr_of)
        hist.append((loss.item(), float(norm), lr, {k: params[k] - init[k] for k in params}))
    return hist


@pytest.mark.parametrize("name", util.fixtures("F7_"))
def test_train_oracle_matches_reference_optimizer(name):
    z, args = util.load_fixture(name)
    hist = oracle_train(args, args["steps"])
    for s, (loss, norm, lr, delta) in enumerate(hist):
        assert abs(loss - z["loss"][s]) <= 2e-5 * max(1.0, abs(z["loss"][s])), (s, loss, z["loss"][s])
        assert abs(norm - z["grad_norm"][s]) <= 1e-4 * z["grad_norm"][s]
        assert abs(lr - z["lr"][s]) <= 1e-9
        for n, dlt in delta.items():
            # updates are O(lr); Adam's m/sqrt(v) amplifies gradient rounding where |g| ~ eps -> absolute tolerance
            # 2e-3 of the step size, relative 1e-3
            util.check_against(z, f"delta/{s}/{n}", dlt, atol=2e-3 * max(args["lr"], 1e-12) * (1 if lr > 0 else 0) + 1e-9,
                               rtol=2e-3, cap=util.MAX_TRAIN, what=f"step {s} ")


def test_weight_decay_groups_follow_reference_quirk():
    assert T.weight_decay_of("unified_encoder.unified_encoder.0.ffn.norm.weight") == 0.01   # NOT exempt in the reference
    assert T.weight_decay_of("unified_encoder.unified_encoder.0.ffn.norm.bias") == 0.0
    assert T.weight_decay_of("unified_encoder.unified_encoder.0.ffn.linear1.weight") == 0.01
    assert np.isclose(T.warmup_cosine(0, 2, 10), 0.0) and np.isclose(T.warmup_cosine(2, 2, 10), 1.0)
