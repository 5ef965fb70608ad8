"""CPU: the set-criterion oracle (oracle/loss_oracle.py) against fixture F9 produced with the reference's own
HungarianMatcher / SetCriterion classes (tests/golden/make_golden.py:run_criterion_case)."""
import numpy as np
import torch

from oracle import loss_oracle as LO
from pq3d_amd import synth
from tests import util

W = dict(cost_class=2.0, cost_mask=5.0, cost_dice=2.0)


def run_oracle():
    masks, logits, labels, seg = synth.criterion_inputs()
    masks = [m.requires_grad_(True) for m in masks]
    logits = [l.requires_grad_(True) for l in logits]
    losses, idx = LO.set_criterion(masks, logits, labels, seg, num_classes=20, **W)
    total, _ = LO.instseg_loss(losses, **W)
    total.backward()
    return masks, logits, losses, idx, total


def test_set_criterion_oracle_matches_reference():
    z, _ = util.load_fixture("F9_set_criterion")
    masks, logits, losses, idx, total = run_oracle()
    assert abs(total.item() - float(z["total"])) <= 1e-5 * abs(float(z["total"]))
    for k, v in losses.items():
        assert abs(v.item() - float(z["loss/" + k])) <= 2e-6 * max(1.0, abs(float(z["loss/" + k]))), k
    for b, (i, j) in enumerate(idx):
        assert np.array_equal(i.numpy(), z[f"indices/{b}/q"]) and np.array_equal(j.numpy(), z[f"indices/{b}/t"])
    for l in range(len(masks)):
        util.check_against(z, f"grad/mask/{l}", masks[l].grad, atol=1e-7, rtol=1e-4)
        util.check_against(z, f"grad/logits/{l}", torch.nan_to_num(logits[l].grad), atol=1e-7, rtol=1e-4, cap=util.MAX_GRAD)


def test_direct_losses_oracle_matches_reference():
    z, _ = util.load_fixture("F10_direct_losses")
    masks, logits, tgt, pad, labels, obj_masks, lab2 = synth.direct_loss_inputs()
    masks = [m.requires_grad_(True) for m in masks]
    logits = [l.requires_grad_(True) for l in logits]
    losses = LO.direct_criterion(masks, logits, tgt, pad, labels)
    total, _ = LO.instseg_loss(losses, **W)
    ml = LO.mask_loss({"gt_attn_mask": tgt.logical_not(), "instance_labels": lab2, "obj_masks": obj_masks,
                       "padding_mask": pad, "predictions_mask": masks, "predictions_class": logits})
    (total + ml).backward()
    assert abs(total.item() - float(z["total"])) <= 1e-5 * abs(float(z["total"]))
    assert abs(ml.item() - float(z["mask_loss"])) <= 1e-5 * abs(float(z["mask_loss"]))
    for k, v in losses.items():
        assert abs(v.item() - float(z["loss/" + k])) <= 2e-6 * max(1.0, abs(float(z["loss/" + k]))), k
    for l in range(len(masks)):
        util.check_against(z, f"grad/mask/{l}", masks[l].grad, atol=1e-7, rtol=1e-4)
        util.check_against(z, f"grad/logits/{l}", logits[l].grad, atol=1e-7, rtol=1e-4, cap=util.MAX_GRAD)
