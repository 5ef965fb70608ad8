"""Instance-segmentation set criterion on the decoder's outputs: drop-in counterparts of
``HungarianMatcher`` (modules/third_party/mask3d/matcher.py:67-215), ``SetCriterion`` (criterion.py:95-270) and
``InstSegLoss`` (optim/loss/instseg_loss.py:9-52) for the shipped configuration (criterion_type 'set',
num_points = -1, class_weights = -1).

Where the reference loops over scenes and layers with a dozen torch ops each ((L n_b + 1) x B cost matrices per step,
each followed by a device->host copy), this runs three launches for the cost matrices of ALL layers and scenes
(include/pq3d_hip.h: prep, one grouped + batched fp32 MFMA GEMM, cost), ONE device->host copy, the assignments on
the host (scipy, as the reference), then the losses as gathers of the already computed cost entries plus one
gradient launch and two cross-entropy launches.  No CPU fallback for the device part."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn
from scipy.optimize import linear_sum_assignment
from torch.autograd import Function

from . import _lib as L


class HungarianMatcher(nn.Module):
    """matcher.py:67-103 (parameter holder; the cost matrix is built by SetCriterion's fused path)."""

    def __init__(self, cost_class: float = 1, cost_mask: float = 1, cost_dice: float = 1, num_points: int = 0,
                 ignore_label: int = -100):
        super().__init__()
        assert cost_class != 0 or cost_mask != 0 or cost_dice != 0, "all costs cant be 0"
        if num_points != -1:
            raise NotImplementedError("random point sub-sampling (num_points != -1) is not used by the shipped config")
        self.cost_class, self.cost_mask, self.cost_dice = cost_class, cost_mask, cost_dice
        self.num_points, self.ignore_label = num_points, ignore_label


def _targets_to_device(instance_labels: Sequence[torch.Tensor], segment_masks: Sequence[torch.Tensor], Ns: int, device):
    """list of [n_b] labels / [n_b, S_b] masks -> padded T [B,Nt,Ns] fp32, labels [B,Nt], seg_len, n_inst (int32)."""
    B = len(segment_masks)
    n_inst = [int(m.shape[0]) for m in segment_masks]
    seg_len = [int(m.shape[1]) for m in segment_masks]
    assert max(seg_len) <= Ns, "target masks have more segments than the prediction"
    Nt = max(max(n_inst), 1)
    T = torch.zeros(B, Nt, Ns, dtype=torch.float32)
    lab = torch.zeros(B, Nt, dtype=torch.int64)
    for b in range(B):
        T[b, :n_inst[b], :seg_len[b]] = segment_masks[b].float().cpu()
        lab[b, :n_inst[b]] = instance_labels[b].cpu()
    T = T.to(device)
    return T, lab.to(device), torch.tensor(seg_len, dtype=torch.int32, device=device), \
        torch.tensor(n_inst, dtype=torch.int32, device=device), n_inst, T.sum(-1)


def _costs(masks, logits, T, labels, seg_len, n_inst_dev, t_sum, w, ignore_label):
    """Cost matrices of ALL prediction layers and scenes in three launches: cost[l] = (total, mask term, dice term)
    [layers, 3, B, Nq, Nt].  Also returns what the gradient stage needs: sigma(X), (T X | T sigma(X)), sum_s sigma."""
    n_layers = len(masks)
    B, Ns, Nq = masks[0].shape
    Nt, Ccls = T.shape[1], logits[0].shape[-1]
    dev = T.device
    assert n_layers <= L.MAXG // 2, "too many prediction layers for one grouped launch"
    nsplit = L.lib().pq3d_mask_cost_nsplit(Ns)
    sig = torch.empty(n_layers, B, Ns, Nq, dtype=torch.float32, device=dev)
    part = torch.empty(2, n_layers, B, nsplit, Nq, dtype=torch.float32, device=dev)
    p = L.MaskPrepDesc()
    p.layers, p.B, p.Ns, p.Nq, p.nsplit = n_layers, B, Ns, Nq, nsplit
    for l in range(n_layers):
        p.X[l] = L.ptr(masks[l])
    p.seg_len, p.sig, p.sp_part, p.sg_part = L.ptr(seg_len), L.ptr(sig), L.ptr(part[0]), L.ptr(part[1])
    L.check(L.lib().pq3d_mask_cost_prep(C.byref(p), L.stream()), "pq3d_mask_cost_prep")
    # ONE grouped + batched exact-f32 MFMA GEMM: groups = (layer, X | sigma(X)), batch = scenes
    TXS = torch.empty(n_layers, 2, B, Nt, Nq, dtype=torch.float32, device=dev)
    Bops = [t for l in range(n_layers) for t in (masks[l], sig[l])]
    Cs = [TXS[l, k] for l in range(n_layers) for k in (0, 1)]
    L.gemm(M=Nt, N=Nq, K=Ns, A=[T] * (2 * n_layers), B=Bops, Cs=Cs, ct=L.F32, lda=Ns, ldb=Nq, ldc=Nq, transB=True,
           batch=B, strideA=Nt * Ns, strideB=Ns * Nq, strideC=Nt * Nq)
    cost = torch.empty(n_layers, 3, B, Nq, Nt, dtype=torch.float32, device=dev)
    d = L.MatchCostDesc()
    d.layers, d.B, d.Nq, d.Nt, d.Ns, d.C, d.nsplit = n_layers, B, Nq, Nt, Ns, Ccls, nsplit
    d.w_class, d.w_mask, d.w_dice, d.ignore_label = w[0], w[1], w[2], ignore_label
    d.TXS, d.sp_part, d.sg_part, d.t_sum = map(L.ptr, (TXS, part[0], part[1], t_sum))
    d.seg_len, d.n_inst, d.labels, d.cost = map(L.ptr, (seg_len, n_inst_dev, labels, cost))
    for l in range(n_layers):
        d.cls_logits[l] = L.ptr(logits[l])
    L.check(L.lib().pq3d_match_cost(C.byref(d), L.stream()), "pq3d_match_cost")
    return cost, (sig, TXS, part[1].sum(2))


class _SetCriterionFn(Function):
    """(mask logits of every layer, class logits of every layer) -> losses [n_layers, 3] = (ce, mask, dice).
    Six kernel launches per step (prep, GEMM, cost, CE forward | mask gradient, CE backward) + a handful of small
    index ops vectorised over layers."""

    @staticmethod
    def forward(ctx, crit, T, labels, seg_len, n_inst_dev, n_inst, t_sum, n_layers, *preds):
        masks = [p.contiguous().float() for p in preds[:n_layers]]
        logits = [p.contiguous().float() for p in preds[n_layers:]]
        B, Ns, Nq = masks[0].shape
        dev, Ccls = T.device, logits[0].shape[-1]
        m = crit.matcher
        w = (float(m.cost_class), float(m.cost_mask), float(m.cost_dice))
        cost_all, keep = _costs(masks, logits, T, labels, seg_len, n_inst_dev, t_sum, w, m.ignore_label)
        host = cost_all[:, 0].cpu().numpy()            # the ONE device->host copy of the step: [layers, B, Nq, Nt]
        Nm = max(min(Nq, max(n_inst)), 1)
        q_idx = np.zeros((n_layers, B, Nm), dtype=np.int32)
        t_idx = np.zeros((n_layers, B, Nm), dtype=np.int32)
        n_match = np.zeros((n_layers, B), dtype=np.int32)
        indices: List[List[Tuple[torch.Tensor, torch.Tensor]]] = []
        for l in range(n_layers):
            per = []
            for b in range(B):
                i, j = linear_sum_assignment(host[l, b, :, :n_inst[b]])    # scipy, as matcher.py:184
                q_idx[l, b, :len(i)], t_idx[l, b, :len(j)], n_match[l, b] = i, j, len(i)
                per.append((torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64)))
            indices.append(per)
        packed = torch.from_numpy(np.concatenate([q_idx.reshape(-1), t_idx.reshape(-1), n_match.reshape(-1)])).to(dev)
        nqt = n_layers * B * Nm
        q_idx_d, t_idx_d = packed[:nqt].view(n_layers, B, Nm), packed[nqt:2 * nqt].view(n_layers, B, Nm)
        n_match_d = packed[2 * nqt:].view(n_layers, B)
        valid = torch.arange(Nm, device=dev)[None, None, :] < n_match_d[:, :, None]             # [layers, B, Nm]
        lidx = torch.arange(n_layers, device=dev)[:, None, None].expand(n_layers, B, Nm)
        bidx = torch.arange(B, device=dev)[None, :, None].expand(n_layers, B, Nm)
        qi, ti = q_idx_d.long(), t_idx_d.long()
        nmf = n_match_d.float().clamp(min=1.0)
        # matched-pair losses ARE the matched entries of the cost matrices (criterion.py:27-70 with num_points = -1)
        pair = cost_all[:, 1:3][lidx, :, bidx, qi, ti] * valid[..., None]                        # [layers, B, Nm, 2]
        losses = torch.zeros(n_layers, 3, dtype=torch.float32, device=dev)
        losses[:, 1:3] = (pair.sum(2) / nmf[..., None]).mean(1)
        # classification (criterion.py:136-163): unmatched queries -> the no-object class (= num_classes)
        tgt = torch.full((n_layers, B, Nq), crit.num_classes, dtype=torch.int64, device=dev)
        tgt[lidx[valid], bidx[valid], qi[valid]] = labels[bidx[valid], ti[valid]]
        row_loss = torch.empty(n_layers, B * Nq, dtype=torch.float32, device=dev)
        lse = torch.empty_like(row_loss)
        ce = L.CeDesc()
        ce.layers, ce.C, ce.R, ce.ignore_index = n_layers, Ccls, B * Nq, crit.ignore_label
        for l in range(n_layers):
            ce.logits[l] = L.ptr(logits[l])
        ce.target, ce.row_loss, ce.lse = L.ptr(tgt), L.ptr(row_loss), L.ptr(lse)
        L.check(L.lib().pq3d_cross_entropy_fwd(C.byref(ce), L.stream()), "pq3d_cross_entropy_fwd")
        cnt = (tgt != crit.ignore_label).sum((1, 2)).clamp(min=1).float()
        losses[:, 0] = row_loss.sum(1) / cnt
        ctx.crit, ctx.n_layers, ctx.keep = crit, n_layers, keep
        ctx.masks, ctx.logits, ctx.tgt, ctx.lse, ctx.cnt = masks, logits, tgt, lse, cnt
        ctx.targets = (T, seg_len, t_sum, q_idx_d.contiguous(), t_idx_d.contiguous(), n_match_d.contiguous(), nmf, Nm)
        ctx.in_dtypes = [p.dtype for p in preds]
        crit._last_indices = indices
        return losses

    @staticmethod
    def backward(ctx, g):
        T, seg_len, t_sum, q_idx_d, t_idx_d, n_match_d, nmf, Nm = ctx.targets
        n_layers = ctx.n_layers
        B, Ns, Nq = ctx.masks[0].shape
        Nt = T.shape[1]
        sig, TXS, sig_sum = ctx.keep
        g = g.contiguous().float()
        gmd = (g[:, 1:3, None] / (nmf[:, None, :] * B)).contiguous()    # [layers, 2, B]; the 1/S_b is applied in-kernel
        scale = (g[:, 0] / ctx.cnt).contiguous()                        # [layers]
        dmasks = [torch.empty_like(m) for m in ctx.masks]
        dlogits = [torch.empty_like(t) for t in ctx.logits]
        d = L.MaskGradDesc()
        d.layers, d.B, d.Ns, d.Nq, d.Nt, d.Nm = n_layers, B, Ns, Nq, Nt, Nm
        d.sig, d.T, d.TXS, d.sig_sum, d.t_sum, d.seg_len = map(L.ptr, (sig, T, TXS, sig_sum.contiguous(), t_sum, seg_len))
        d.q_idx, d.t_idx, d.n_match, d.g = map(L.ptr, (q_idx_d, t_idx_d, n_match_d, gmd))
        for l in range(n_layers):
            d.dX[l] = L.ptr(dmasks[l])
        L.check(L.lib().pq3d_matched_mask_grad(C.byref(d), L.stream()), "pq3d_matched_mask_grad")
        ce = L.CeDesc()
        ce.layers, ce.C, ce.R, ce.ignore_index = n_layers, dlogits[0].shape[-1], B * Nq, ctx.crit.ignore_label
        for l in range(n_layers):
            ce.logits[l], ce.dlogits[l] = L.ptr(ctx.logits[l]), L.ptr(dlogits[l])
        ce.target, ce.lse, ce.scale = L.ptr(ctx.tgt), L.ptr(ctx.lse), L.ptr(scale)
        L.check(L.lib().pq3d_cross_entropy_bwd(C.byref(ce), L.stream()), "pq3d_cross_entropy_bwd")
        grads = [t.to(dt) for t, dt in zip(dmasks + dlogits, ctx.in_dtypes)]
        return (None,) * 8 + tuple(grads)


class SetCriterion(nn.Module):
    """criterion.py:95-270: forward(predictions_mask, predictions_class, instance_labels, segment_masks) ->
    (losses dict with 'loss_ce', 'loss_mask', 'loss_dice' and the '_i' auxiliary copies, indices of the last layer)."""

    def __init__(self, num_classes, matcher, weight_dict, losses, num_points, class_weights, ignore_label):
        super().__init__()
        if num_points != -1 or class_weights != -1:
            raise NotImplementedError("only num_points = -1, class_weights = -1 (configs/instseg_sceneverse.yaml:163-168)")
        self.num_classes, self.matcher, self.weight_dict, self.losses = num_classes, matcher, weight_dict, list(losses)
        self.num_points, self.class_weights, self.ignore_label = num_points, class_weights, ignore_label
        self._last_indices = None

    def forward(self, predictions_mask, predictions_class, instance_labels, segment_masks):
        n = len(predictions_mask)
        # reference order: the LAST prediction is the main output, the others are the auxiliary '_i' losses
        order = [n - 1] + list(range(n - 1))
        masks = [predictions_mask[i] for i in order]
        logits = [predictions_class[i] for i in order]
        dev = masks[0].device
        T, labels, seg_len, n_inst_dev, n_inst, t_sum = _targets_to_device(instance_labels, segment_masks,
                                                                         masks[0].shape[1], dev)
        out = _SetCriterionFn.apply(self, T, labels, seg_len, n_inst_dev, n_inst, t_sum, n, *masks, *logits)
        names = {"labels": [(0, "loss_ce")], "masks": [(1, "loss_mask"), (2, "loss_dice")]}
        losses: Dict[str, torch.Tensor] = {}
        for l in range(n):
            suffix = "" if l == 0 else f"_{l - 1}"
            for kind in self.losses:
                for col, name in names[kind]:
                    losses[name + suffix] = out[l, col]
        return losses, self._last_indices[0]


class InstSegLoss(nn.Module):
    """optim/loss/instseg_loss.py:9-52: criterion_type 'set' (Hungarian matching, stage 1) or 'direct' (ground-truth
    masks, query i <-> instance i, configs/instseg_sceneverse_gt.yaml:160)."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        loss_cfg = cfg.model.get(self.__class__.__name__)
        self.criterion_type = loss_cfg.get("criterion_type", "set")
        assert self.criterion_type in ["set", "direct"]
        matcher = HungarianMatcher(**dict(loss_cfg.matcher))
        self.weight_dict = {"loss_ce": matcher.cost_class, "loss_mask": matcher.cost_mask, "loss_dice": matcher.cost_dice}
        if self.criterion_type == "set":
            self.set_criterion = SetCriterion(matcher=matcher, weight_dict=self.weight_dict, **dict(loss_cfg.criterion))
        else:   # instseg_loss.py:34-35: DirectCriterion(**loss_cfg.criterion) swallows the set criterion's other keys
            self.direct_criterion = DirectCriterion(**dict(loss_cfg.criterion))

    def forward(self, data_dict):
        if self.criterion_type == "direct":   # instseg_loss.py:41-43
            losses = self.direct_criterion(data_dict["predictions_mask"], data_dict["predictions_class"],
                                           data_dict["target_masks"], data_dict["target_masks_pad_masks"],
                                           data_dict["target_labels"])
        else:
            losses, indices = self.set_criterion(data_dict["predictions_mask"], data_dict["predictions_class"],
                                                 data_dict["instance_labels"], data_dict["segment_masks"])
            data_dict["indices"] = indices
        for k in list(losses.keys()):
            losses[k] = losses[k] * self.weight_dict[k.split("_")[0] + "_" + k.split("_")[1]]
        return [sum(losses.values()), losses]


# ------------------------------------------------------------------------------------------------ padded (direct) losses
class _PaddedMaskLoss(Function):
    """(X [B,S,N] mask logits, T [B,N,S], P [B,N,S] bool) -> (batch_mask_loss, batch_dice_loss)
    (optim/loss/instseg_loss.py:54-85): two launches forward (tile sums + tiny finalize ops), one backward."""

    @staticmethod
    def forward(ctx, X, T, P):
        X = X.contiguous().float()
        T = T.contiguous().float()
        P = P.contiguous()
        if P.dtype != torch.bool:
            P = P != 0
        B, S, N = X.shape
        nt = (S + 63) // 64
        part = torch.empty(B, nt, N, 4, dtype=torch.float32, device=X.device)
        L.check(L.lib().pq3d_padded_mask_sums(L.ptr(X), L.ptr(T), L.ptr(P), L.ptr(part), B, S, N, L.stream()),
                "pq3d_padded_mask_sums")
        sums = part.sum(1)                                   # [B, N, 4]
        sb, sp, si, su = sums.unbind(-1)
        valid = sp > 0
        cnt = valid.sum().float()
        lm = torch.where(valid, sb / (sp + 1e-6), torch.zeros_like(sb)).sum() / cnt
        dice = 1.0 - (2.0 * si + 1e-6) / (su + 1e-6)
        ld = torch.where(valid, dice, torch.zeros_like(dice)).sum() / cnt
        ctx.save_for_backward(X, T, P, sums.contiguous(), valid, cnt)
        return lm, ld

    @staticmethod
    def backward(ctx, gm_, gd_):
        X, T, P, sums, valid, cnt = ctx.saved_tensors
        B, S, N = X.shape
        sp = sums[..., 1]
        gm = (torch.where(valid, gm_ / (cnt * (sp + 1e-6)), torch.zeros_like(sp))).contiguous()
        gd = (torch.where(valid, gd_ / cnt, torch.zeros_like(sp))).contiguous()
        dX = torch.empty_like(X)
        L.check(L.lib().pq3d_padded_mask_grad(L.ptr(X), L.ptr(T), L.ptr(P), L.ptr(sums), L.ptr(gm), L.ptr(gd), L.ptr(dX),
                                              B, S, N, L.stream()), "pq3d_padded_mask_grad")
        return dX, None, None


def padded_mask_losses(pred_masks: torch.Tensor, targets: torch.Tensor, padding_mask: torch.Tensor):
    """(batch_mask_loss, batch_dice_loss) for pred_masks in the model's own layout [B, S, N] (segments first)."""
    return _PaddedMaskLoss.apply(pred_masks, targets, padding_mask)


def _segments_first(logits: torch.Tensor) -> torch.Tensor:
    """The reference passes pred_masks.permute(0, 2, 1) ([B, N, S]); undo the view (no copy when it is one)."""
    return logits.permute(0, 2, 1)


def batch_mask_loss(logits, targets, padding_mask):
    """optim/loss/instseg_loss.py:77-85; logits [B, N, S] as in the reference."""
    return padded_mask_losses(_segments_first(logits), targets, padding_mask)[0]


def batch_dice_loss(logits, targets, padding_mask):
    """optim/loss/instseg_loss.py:54-75; logits [B, N, S] as in the reference."""
    return padded_mask_losses(_segments_first(logits), targets, padding_mask)[1]


def cross_entropy_rows(logits: torch.Tensor, labels: torch.Tensor, ignore_index: int = -100,
                       add: Optional[torch.Tensor] = None) -> torch.Tensor:
    """F.cross_entropy(logits.view(-1, C), labels.view(-1), ignore_index) through the CE kernels (mean over kept rows).
    ``add``: a 0-dim fp32 device tensor (the other terms of a total loss) added inside the mean launch -- "total = other + ce"
    without an elementwise launch; its gradient is the upstream gradient."""
    return _RowCE.apply(logits, labels, ignore_index, add)


class _RowCE(Function):
    """Forward: per-row loss + log-sum-exp (one launch), then the mean over the kept rows and 1 / kept (one launch).  Backward:
    one launch; the upstream gradient and 1 / kept stay on the device (scale, scale_mul).  No framework kernels."""

    @staticmethod
    def forward(ctx, logits, labels, ignore_index, add=None):
        assert add is None or (add.numel() == 1 and add.dtype == torch.float32 and add.is_cuda)
        x = logits.contiguous().float().view(-1, logits.shape[-1])
        t = labels.contiguous().view(-1).long()
        R, Ccls = x.shape
        buf = torch.empty(2 * R + 1, dtype=torch.float32, device=x.device)      # row_loss | lse | 1 / kept
        row_loss, lse, inv = buf[:R], buf[R:2 * R], buf[2 * R:]
        # the RETURNED scalar lives in a tensor of its own: sharing the saved tensors' buffer (and version counter) made a caller's
        # in-place `loss /= accum_steps` / `total += aux` fail the backward with "modified by an inplace operation" 
        loss = torch.empty((), dtype=torch.float32, device=x.device)   # 0-dim, not a view: in-place ops on it are legal
        ce = L.CeDesc()
        ce.layers, ce.C, ce.R, ce.ignore_index = 1, Ccls, R, ignore_index
        ce.logits[0], ce.target, ce.row_loss, ce.lse = L.ptr(x), L.ptr(t), L.ptr(row_loss), L.ptr(lse)
        L.check(L.lib().pq3d_cross_entropy_fwd(C.byref(ce), L.stream()), "pq3d_cross_entropy_fwd")
        L.check(L.lib().pq3d_cross_entropy_mean(C.byref(ce), L.ptr(loss), L.ptr(inv), L.ptr(add), L.stream()),
                "pq3d_cross_entropy_mean")
        ctx.has_add = add is not None
        ctx.save_for_backward(x, t, lse, inv)
        ctx.ignore_index, ctx.shape, ctx.dtype = ignore_index, logits.shape, logits.dtype
        return loss

    @staticmethod
    def backward(ctx, g):
        x, t, lse, inv = ctx.saved_tensors
        dl = torch.empty_like(x)
        scale = g.reshape(1)
        if scale.dtype != torch.float32 or not scale.is_contiguous():
            scale = scale.float().contiguous()
        ce = L.CeDesc()
        ce.layers, ce.C, ce.R, ce.ignore_index = 1, x.shape[1], x.shape[0], ctx.ignore_index
        ce.logits[0], ce.dlogits[0], ce.target, ce.lse, ce.scale = L.ptr(x), L.ptr(dl), L.ptr(t), L.ptr(lse), L.ptr(scale)
        ce.scale_mul = L.ptr(inv)
        L.check(L.lib().pq3d_cross_entropy_bwd(C.byref(ce), L.stream()), "pq3d_cross_entropy_bwd")
        dl = dl.view(ctx.shape)
        return (dl if ctx.dtype == torch.float32 else dl.to(ctx.dtype)), None, None, (g if ctx.has_add else None)


class DirectCriterion(nn.Module):
    """optim/loss/instseg_loss.py:88-133: ground-truth masks, no Hungarian matching (query i <-> instance i)."""

    def __init__(self, losses, ignore_label, **kwargs):
        super().__init__()
        self.losses, self.ignore_label = list(losses), ignore_label

    def loss_labels(self, logits, labels):
        if self.ignore_label != -100:
            labels = torch.where(labels == self.ignore_label, torch.full_like(labels, -100), labels)
        return {"loss_ce": cross_entropy_rows(logits, labels, -100)}

    def loss_masks(self, pred_masks, gt_masks, padding_mask):
        lm, ld = padded_mask_losses(pred_masks, gt_masks, padding_mask)     # pred_masks [B, S, N] as the model emits them
        return {"loss_mask": lm, "loss_dice": ld}

    def get_loss(self, loss, outputs, targets):
        if loss == "labels":
            return self.loss_labels(outputs["pred_logits"], targets["labels"])
        if loss == "masks":
            return self.loss_masks(outputs["pred_masks"], targets["masks"], targets["padding_mask"])
        raise AssertionError(loss)

    def forward(self, predictions_mask, predictions_class, target_masks, target_masks_pad_masks, target_labels):
        losses = {}
        targets = {"labels": target_labels, "masks": target_masks, "padding_mask": target_masks_pad_masks}
        for loss in self.losses:
            losses.update(self.get_loss(loss, {"pred_logits": predictions_class[-1], "pred_masks": predictions_mask[-1]},
                                        targets))
        for i in range(len(predictions_mask) - 1):
            for loss in self.losses:
                l_dict = self.get_loss(loss, {"pred_logits": predictions_class[i], "pred_masks": predictions_mask[i]},
                                       targets)
                losses.update({k + f"_{i}": v for k, v in l_dict.items()})
        return losses


def mask_loss(data_dict):
    """optim/loss/query3d_loss.py:28-39 (stage-2 'mask_loss': 5 x BCE + 2 x dice + 2 x object-masked class CE per layer)."""
    mask_gt = data_dict["gt_attn_mask"].logical_not().float()
    labels = torch.where(data_dict["obj_masks"].bool(), data_dict["instance_labels"],
                         torch.full_like(data_dict["instance_labels"], -100))
    total = 0
    for mask_pred, mask_cls in zip(data_dict["predictions_mask"], data_dict["predictions_class"]):
        lm, ld = padded_mask_losses(mask_pred, mask_gt, data_dict["padding_mask"])
        total = total + lm * 5 + ld * 2
        # (CE(none) * obj_masks).sum() / (obj_masks.sum() + 1e-6): rows outside obj_masks are ignored rows of the CE kernel
        n = data_dict["obj_masks"].sum().float()
        total = total + cross_entropy_rows(mask_cls, labels, -100) * (n / (n + 1e-6)) * 2
    return total
