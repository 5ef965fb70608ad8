"""CPU oracle of the instance-segmentation set criterion.  TEST INFRASTRUCTURE ONLY.

Plain-torch restatement of HungarianMatcher.memory_efficient_forward (modules/third_party/mask3d/matcher.py:104-184),
SetCriterion.loss_labels / loss_masks / forward (criterion.py:136-268) with num_points = -1, class_weights = -1, and
InstSegLoss.forward's weighting (optim/loss/instseg_loss.py:38-52).  Written in the reference's own formulation (two
BCE einsums, not the algebraic shortcut the HIP path uses) so that the shortcut is what gets tested.  Pinned by fixture
F9 produced with the reference's classes (tests/golden/make_golden.py:run_criterion_case)."""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment

Tensor = torch.Tensor


def cost_matrix(pred_logits: Tensor, pred_masks: Tensor, labels: Tensor, tgt_mask: Tensor, *, cost_class: float,
                cost_mask: float, cost_dice: float, ignore_label: int = -100) -> Tensor:
    """One scene: pred_logits [Nq, C], pred_masks [Ns, Nq] (segments first), labels [Nt], tgt_mask [Nt, S] -> [Nq, Nt]."""
    out_prob = pred_logits.softmax(-1)
    tgt_ids = labels.clone()
    ign = tgt_ids == ignore_label
    tgt_ids[ign] = 0
    c_class = -out_prob[:, tgt_ids]
    c_class[:, ign] = -1.0
    S = tgt_mask.shape[1]
    x = pred_masks.T[:, :S].float()                  # matcher.py:136-148: the first S columns
    t = tgt_mask.float()
    pos = F.binary_cross_entropy_with_logits(x, torch.ones_like(x), reduction="none")
    neg = F.binary_cross_entropy_with_logits(x, torch.zeros_like(x), reduction="none")
    c_mask = (torch.einsum("nc,mc->nm", pos, t) + torch.einsum("nc,mc->nm", neg, 1 - t)) / S
    sg = x.sigmoid()
    c_dice = 1 - (2 * torch.einsum("nc,mc->nm", sg, t) + 1) / (sg.sum(-1)[:, None] + t.sum(-1)[None, :] + 1)
    return cost_mask * c_mask + cost_class * c_class + cost_dice * c_dice


def match(pred_logits: Tensor, pred_masks: Tensor, instance_labels: Sequence[Tensor], segment_masks: Sequence[Tensor],
          **w) -> List[Tuple[Tensor, Tensor]]:
    out = []
    with torch.no_grad():
        for b in range(pred_logits.shape[0]):
            Cm = cost_matrix(pred_logits[b], pred_masks[b], instance_labels[b], segment_masks[b], **w)
            i, j = linear_sum_assignment(Cm.numpy())
            out.append((torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64)))
    return out


def layer_losses(pred_logits: Tensor, pred_masks: Tensor, instance_labels, segment_masks, indices, *, num_classes: int,
                 ignore_label: int = -100) -> Dict[str, Tensor]:
    B, Nq = pred_logits.shape[:2]
    target_classes = torch.full((B, Nq), num_classes, dtype=torch.int64)
    for b, (i, j) in enumerate(indices):
        target_classes[b, i] = instance_labels[b][j]
    loss_ce = F.cross_entropy(pred_logits.float().transpose(1, 2), target_classes, ignore_index=ignore_label)
    lm, ld = [], []
    for b, (i, j) in enumerate(indices):
        t = segment_masks[b][j].float()
        x = pred_masks[b][:, i].T[:, :t.shape[1]]
        n = t.shape[0]
        lm.append(F.binary_cross_entropy_with_logits(x, t, reduction="none").mean(1).sum() / n)
        sg = x.sigmoid()
        ld.append((1 - (2 * (sg * t).sum(-1) + 1) / (sg.sum(-1) + t.sum(-1) + 1)).sum() / n)
    return {"loss_ce": loss_ce, "loss_mask": torch.stack(lm).mean(), "loss_dice": torch.stack(ld).mean()}


def set_criterion(predictions_mask, predictions_class, instance_labels, segment_masks, *, num_classes: int,
                  cost_class: float, cost_mask: float, cost_dice: float, ignore_label: int = -100):
    w = dict(cost_class=cost_class, cost_mask=cost_mask, cost_dice=cost_dice, ignore_label=ignore_label)
    losses: Dict[str, Tensor] = {}
    idx_last = match(predictions_class[-1], predictions_mask[-1], instance_labels, segment_masks, **w)
    losses.update(layer_losses(predictions_class[-1], predictions_mask[-1], instance_labels, segment_masks, idx_last,
                               num_classes=num_classes, ignore_label=ignore_label))
    for i, (lg, mk) in enumerate(zip(predictions_class[:-1], predictions_mask[:-1])):
        idx = match(lg, mk, instance_labels, segment_masks, **w)
        for k, v in layer_losses(lg, mk, instance_labels, segment_masks, idx, num_classes=num_classes,
                                 ignore_label=ignore_label).items():
            losses[f"{k}_{i}"] = v
    return losses, idx_last


def instseg_loss(losses: Dict[str, Tensor], *, cost_class: float, cost_mask: float, cost_dice: float):
    wd = {"loss_ce": cost_class, "loss_mask": cost_mask, "loss_dice": cost_dice}
    weighted = {k: v * wd["_".join(k.split("_")[:2])] for k, v in losses.items()}
    return sum(weighted.values()), weighted


# ---- padded (direct) losses -------------------------------------------------------------------------------------
def batch_dice_loss(logits: Tensor, targets: Tensor, padding_mask: Tensor) -> Tensor:
    """optim/loss/instseg_loss.py:54-75 (logits [B, N, S])."""
    probs = logits.sigmoid()
    inter = (probs * targets * padding_mask).sum(-1)
    union = ((probs + targets) * padding_mask).sum(-1)
    dice = 1.0 - (2.0 * inter + 1e-6) / (union + 1e-6)
    inst = padding_mask.sum(-1) > 0
    dice = torch.where(inst, dice, torch.zeros_like(dice))
    return dice.sum() / inst.sum()


def batch_mask_loss(logits: Tensor, targets: Tensor, padding_mask: Tensor) -> Tensor:
    """optim/loss/instseg_loss.py:77-85."""
    loss = F.binary_cross_entropy_with_logits(logits, targets, reduction="none")
    loss = (loss * padding_mask).sum(-1) / (padding_mask.sum(-1) + 1e-6)
    inst = padding_mask.sum(-1) > 0
    loss = torch.where(inst, loss, torch.zeros_like(loss))
    return loss.sum() / inst.sum()


def direct_criterion(predictions_mask, predictions_class, target_masks, target_masks_pad_masks, target_labels, *,
                     ignore_label: int = -100) -> Dict[str, Tensor]:
    """DirectCriterion.forward (optim/loss/instseg_loss.py:88-133), losses = ['labels', 'masks']."""
    def one(lg, mk):
        lab = target_labels.clone()
        if ignore_label != -100:
            lab[lab == ignore_label] = -100
        pm = mk.permute(0, 2, 1)
        return {"loss_ce": F.cross_entropy(lg.reshape(-1, lg.shape[-1]), lab.reshape(-1)),
                "loss_mask": batch_mask_loss(pm, target_masks, target_masks_pad_masks),
                "loss_dice": batch_dice_loss(pm, target_masks, target_masks_pad_masks)}
    losses = one(predictions_class[-1], predictions_mask[-1])
    for i in range(len(predictions_mask) - 1):
        losses.update({f"{k}_{i}": v for k, v in one(predictions_class[i], predictions_mask[i]).items()})
    return losses


def mask_loss(data_dict) -> Tensor:
    """optim/loss/query3d_loss.py:28-39."""
    mask_gt = data_dict["gt_attn_mask"].logical_not()
    total = 0
    for mask_pred, mask_cls in zip(data_dict["predictions_mask"], data_dict["predictions_class"]):
        mp = mask_pred.permute(0, 2, 1)
        total = total + batch_mask_loss(mp, mask_gt.float(), data_dict["padding_mask"]) * 5 \
            + batch_dice_loss(mp, mask_gt.float(), data_dict["padding_mask"]) * 2
        ce = F.cross_entropy(mask_cls.reshape(-1, mask_cls.shape[-1]), data_dict["instance_labels"].reshape(-1),
                             reduction="none")
        om = data_dict["obj_masks"].reshape(-1)
        total = total + (ce * om).sum() / (om.sum() + 1e-6) * 2
    return total
