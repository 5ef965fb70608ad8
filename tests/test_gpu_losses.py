"""GPU: the HIP set criterion (pq3d_amd/losses.py: cost-matrix kernels + grouped fp32 MFMA GEMM, host LSA, losses as
gathers of the cost entries, gradient kernels) against fixture F9 (reference classes) and the loss oracle."""
import numpy as np
import pytest
import torch

from oracle import loss_oracle as LO
from pq3d_amd import synth
from pq3d_amd.losses import HungarianMatcher, SetCriterion
from tests import util

pytestmark = pytest.mark.gpu
DEV = "cuda"
W = dict(cost_class=2.0, cost_mask=5.0, cost_dice=2.0)
WD = {"loss_ce": 2.0, "loss_mask": 5.0, "loss_dice": 2.0}


def make_criterion():
    matcher = HungarianMatcher(num_points=-1, ignore_label=-100, **W)
    return SetCriterion(num_classes=20, matcher=matcher, weight_dict=WD, losses=["labels", "masks"], num_points=-1,
                        class_weights=-1, ignore_label=-100)


def test_set_criterion_matches_reference_fixture():
    z, _ = util.load_fixture("F9_set_criterion")
    masks, logits, labels, seg = synth.criterion_inputs()
    masks = [m.to(DEV).requires_grad_(True) for m in masks]
    logits = [l.to(DEV).requires_grad_(True) for l in logits]
    losses, idx = make_criterion()(masks, logits, labels, seg)
    total = sum(v * WD["_".join(k.split("_")[:2])] for k, v in losses.items())
    total.backward()
    assert abs(total.item() - float(z["total"])) <= 2e-5 * abs(float(z["total"]))
    assert sorted(losses) == sorted(k[5:] for k in z.files if k.startswith("loss/"))
    for k, v in losses.items():
        assert abs(v.item() - float(z["loss/" + k])) <= 1e-5 * max(1.0, abs(float(z["loss/" + k]))), k
    for b, (i, j) in enumerate(idx):
        assert np.array_equal(i.numpy(), z[f"indices/{b}/q"]) and np.array_equal(j.numpy(), z[f"indices/{b}/t"])
    for l in range(len(masks)):
        util.check_against(z, f"grad/mask/{l}", masks[l].grad, atol=1e-7, rtol=2e-4)
        util.check_against(z, f"grad/logits/{l}", torch.nan_to_num(logits[l].grad), atol=1e-7, rtol=2e-4, cap=util.MAX_GRAD)


@pytest.mark.parametrize("B,Ns,Nq,C,nl", [(2, 300, 100, 201, 2), (4, 1024, 200, 201, 3)])
def test_set_criterion_matches_oracle_at_larger_sizes(B, Ns, Nq, C, nl):
    """cost matrices, assignments, losses and gradients at decoder-like sizes (ragged scenes, more targets than the
    block sizes of the kernels, Nq not a multiple of 64)."""
    r = np.random.default_rng(B * 1000 + Ns)
    seg_len = [Ns] + [int(x) for x in r.integers(Ns // 2, Ns, B - 1)]
    n_inst = [int(x) for x in r.integers(3, min(Nq, 90), B)]
    masks, logits, labels, seg = synth.criterion_inputs(seed=B + Ns, B=B, Ns=Ns, Nq=Nq, C=C, n_layers=nl, seg_len=seg_len,
                                                        n_inst=n_inst)
    crit = make_criterion()
    crit.num_classes = C - 1
    dm = [m.to(DEV).requires_grad_(True) for m in masks]
    dl = [l.to(DEV).requires_grad_(True) for l in logits]
    losses, idx = crit(dm, dl, labels, seg)
    sum(v * WD["_".join(k.split("_")[:2])] for k, v in losses.items()).backward()
    om = [m.clone().requires_grad_(True) for m in masks]
    ol = [l.clone().requires_grad_(True) for l in logits]
    olosses, oidx = LO.set_criterion(om, ol, labels, seg, num_classes=C - 1, **W)
    LO.instseg_loss(olosses, **W)[0].backward()
    for (i, j), (oi, oj) in zip(idx, oidx):
        assert torch.equal(i, oi) and torch.equal(j, oj)
    for k in olosses:
        assert abs(losses[k].item() - olosses[k].item()) <= 2e-5 * max(1.0, abs(olosses[k].item())), k
    for a, b in zip(dm + dl, om + ol):
        ga, gb = torch.nan_to_num(a.grad.cpu()), torch.nan_to_num(b.grad)
        assert float((ga - gb).abs().max()) <= 1e-6 + 2e-4 * float(gb.abs().max())


def test_cost_matrix_entries_match_oracle():
    from pq3d_amd import losses as HL
    masks, logits, labels, seg = synth.criterion_inputs(seed=5, B=2, Ns=520, Nq=70, C=21, n_layers=1, seg_len=(520, 333),
                                                        n_inst=(11, 40))
    T, lab, seg_len, n_inst_dev, n_inst, t_sum = HL._targets_to_device(labels, seg, 520, DEV)
    cost, _ = HL._costs([masks[0].to(DEV)], [logits[0].to(DEV)], T, lab, seg_len, n_inst_dev, t_sum, (2.0, 5.0, 2.0), -100)
    for b in range(2):
        ref = LO.cost_matrix(logits[0][b], masks[0][b], labels[b], seg[b], **W)
        got = cost[0, 0, b, :, :n_inst[b]].cpu()
        assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


def test_direct_criterion_and_stage2_mask_loss_match_reference_fixture():
    from pq3d_amd.losses import DirectCriterion, mask_loss
    z, _ = util.load_fixture("F10_direct_losses")
    masks, logits, tgt, pad, labels, obj_masks, lab2 = synth.direct_loss_inputs()
    masks = [m.to(DEV).requires_grad_(True) for m in masks]
    logits = [l.to(DEV).requires_grad_(True) for l in logits]
    tgt, pad, labels, obj_masks, lab2 = (t.to(DEV) for t in (tgt, pad, labels, obj_masks, lab2))
    losses = DirectCriterion(losses=["labels", "masks"], ignore_label=-100)(masks, logits, tgt, pad, labels)
    total = sum(v * WD["_".join(k.split("_")[:2])] for k, v in losses.items())
    ml = mask_loss({"gt_attn_mask": tgt.logical_not(), "instance_labels": lab2, "obj_masks": obj_masks, "padding_mask": pad,
                    "predictions_mask": masks, "predictions_class": logits})
    (total + ml).backward()
    assert abs(total.item() - float(z["total"])) <= 2e-5 * abs(float(z["total"]))
    assert abs(ml.item() - float(z["mask_loss"])) <= 2e-5 * abs(float(z["mask_loss"]))
    for k, v in losses.items():
        assert abs(v.item() - float(z["loss/" + k])) <= 1e-5 * max(1.0, abs(float(z["loss/" + k]))), k
    for l in range(len(masks)):
        util.check_against(z, f"grad/mask/{l}", masks[l].grad, atol=1e-7, rtol=2e-4)
        util.check_against(z, f"grad/logits/{l}", logits[l].grad, atol=1e-7, rtol=2e-4, cap=util.MAX_GRAD)


def test_padded_mask_losses_match_oracle_at_decoder_sizes():
    from pq3d_amd.losses import padded_mask_losses
    masks, logits, tgt, pad, labels, obj_masks, lab2 = synth.direct_loss_inputs(seed=3, B=4, S=1500, N=200, C=21, n_layers=1,
                                                                               seg_len=(1500, 900, 1333, 64),
                                                                               n_inst=(200, 37, 5, 150))
    x = masks[0].to(DEV).requires_grad_(True)
    lm, ld = padded_mask_losses(x, tgt.to(DEV), pad.to(DEV))
    (3 * lm + 7 * ld).backward()
    xo = masks[0].clone().requires_grad_(True)
    om, od = LO.batch_mask_loss(xo.permute(0, 2, 1), tgt, pad), LO.batch_dice_loss(xo.permute(0, 2, 1), tgt, pad)
    (3 * om + 7 * od).backward()
    assert abs(lm.item() - om.item()) <= 2e-5 * abs(om.item()) and abs(ld.item() - od.item()) <= 2e-5 * abs(od.item())
    assert float((x.grad.cpu() - xo.grad).abs().max()) <= 1e-7 + 2e-4 * float(xo.grad.abs().max())


def _loss_cfg(criterion_type):
    """cfg.model.InstSegLoss as configs/instseg_sceneverse.yaml:157-168 / instseg_sceneverse_gt.yaml:160 lay it out."""
    from pq3d_amd.model import Cfg
    return Cfg({"model": {"InstSegLoss": {
        "criterion_type": criterion_type,
        "matcher": dict(num_points=-1, ignore_label=-100, **W),
        "criterion": dict(num_classes=20, losses=["labels", "masks"], num_points=-1, class_weights=-1, ignore_label=-100)}}})


def test_instseg_loss_wrapper_direct_and_set_match_reference_fixtures():
    """InstSegLoss.forward (optim/loss/instseg_loss.py:38-52): 'direct' -> DirectCriterion on target_masks /
    target_masks_pad_masks / target_labels (fixture F10's weighted total), 'set' -> SetCriterion + indices (F9)."""
    from pq3d_amd.losses import InstSegLoss
    z, _ = util.load_fixture("F10_direct_losses")
    masks, logits, tgt, pad, labels, _om, _l2 = synth.direct_loss_inputs()
    dd = {"predictions_mask": [m.to(DEV).requires_grad_(True) for m in masks],
          "predictions_class": [l.to(DEV).requires_grad_(True) for l in logits],
          "target_masks": tgt.to(DEV), "target_masks_pad_masks": pad.to(DEV), "target_labels": labels.to(DEV)}
    total, losses = InstSegLoss(_loss_cfg("direct"))(dd)
    assert abs(total.item() - float(z["total"])) <= 2e-5 * abs(float(z["total"]))
    assert "indices" not in dd and sorted(losses) == sorted(k[5:] for k in z.files if k.startswith("loss/"))
    z9, _ = util.load_fixture("F9_set_criterion")
    masks, logits, labels9, seg = synth.criterion_inputs()
    dd = {"predictions_mask": [m.to(DEV) for m in masks], "predictions_class": [l.to(DEV) for l in logits],
          "instance_labels": labels9, "segment_masks": seg}
    total9, _ = InstSegLoss(_loss_cfg("set"))(dd)
    assert abs(total9.item() - float(z9["total"])) <= 2e-5 * abs(float(z9["total"]))
    assert len(dd["indices"]) == len(labels9)


def test_cross_entropy_out_of_range_label_poisons_the_loss():
    """torch device-asserts on a label outside [0, C); the CE kernel must not read out of bounds silently: NaN loss."""
    from pq3d_amd.losses import cross_entropy_rows
    x = torch.randn(16, 7, device="cuda")
    t = torch.randint(0, 7, (16,), device="cuda")
    assert torch.isfinite(cross_entropy_rows(x, t))
    for bad in (7, -3, 1 << 40):
        tb = t.clone(); tb[5] = bad
        assert torch.isnan(cross_entropy_rows(x, tb))
    ti = t.clone(); ti[5] = -100
    assert torch.isfinite(cross_entropy_rows(x, ti))


@pytest.mark.parametrize("R,Cls", [(512, 32128), (37, 1031), (64, 2048), (16, 201), (5, 1024)])
def test_cross_entropy_rows_matches_torch_narrow_and_wide_rows(R, Cls):
    """F.cross_entropy (mean over kept rows, ignore_index) forward and backward: the wave-per-row kernels (class logits) and
    the workgroup-per-row kernels of wide rows (>= 1024 classes: the caption head's 32128-way LM head; 1031 = rows that do
    not start on 16-byte boundaries), the mean + 1 / kept on the device, an addend folded into the mean launch."""
    from pq3d_amd.losses import cross_entropy_rows
    g = torch.Generator().manual_seed(R + Cls)
    x = (torch.randn(R, Cls, generator=g) * 3).to("cuda").requires_grad_(True)
    t = torch.randint(0, Cls, (R,), generator=g)
    t[::5] = -100
    t = t.to("cuda")
    xr = x.detach().double().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(xr, t)
    add = torch.full((), 0.75, device="cuda", requires_grad=True)
    out = cross_entropy_rows(x, t, add=add)
    assert abs(float(out) - 0.75 - float(ref)) <= 2e-6 * max(1.0, abs(float(ref)))
    w = torch.tensor(1.7, device="cuda")
    (out * w).backward()
    (ref * 1.7).backward()
    assert float(add.grad) == pytest.approx(1.7)
    err = float((x.grad.double() - xr.grad).abs().max())
    assert err <= 2e-6 * float(xr.grad.abs().max()) + 1e-9, err
    assert float(x.grad[::5].abs().max()) == 0.0     # ignored rows: exact zeros
    # plain call (no addend) on a 3-D logits tensor, as the caption head hands it over
    out2 = cross_entropy_rows(x.detach().view(1, R, Cls), t.view(1, R))
    assert abs(float(out2) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref)))
