"""The frozen PointNet++ point-cloud tokenizer: ``PointNetPP`` of modules/layers/pointnet.py:22-63 as
``ObjectEncoder(backbone='pointnet++', freeze_backbone=True)`` runs it (modules/vision/object_encoder.py:22-28,63-69;
configs/unified_tasks_sceneverse.yaml:144-152) -- three set-abstraction stages (pointnet2_modules.py:23-70) and a Linear.

MI355X design (not the reference's channel-first conv stack): every (cloud, centre, sample) is one channels-last ROW.
Per stage: furthest-point sampling and ball query (pq3d_amd/pointnet2.py kernels), ``pq3d_group_rows`` writes the
centred-xyz + feature rows straight from the ball-query indices, the SharedMLP's 1x1 Conv2d + BatchNorm2d(eval) + ReLU
layers run as row GEMMs with the frozen BatchNorm folded into weight and bias (bias + ReLU in the GEMM epilogue, bf16 or
exact-fp32 MFMA), and ``pq3d_group_maxpool`` takes the max over the samples, which IS the next stage's feature-row
layout -- no transposes anywhere.  Inference only: batch-statistics BatchNorm / training the backbone raises.

Parameter names and shapes are the reference's (``encoder.{i}.mlps.0.layer{j}.conv.weight`` [Cout,Cin,1,1],
``...layer{j}.bn.bn.{weight,bias,running_mean,running_var}``, ``fc.*``), so ``pointnet_tokenizer.pth`` loads unchanged.
No CPU fallback."""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from . import pointnet2 as P2


def group_rows(xyz, new_xyz, feats, feat_channels: int, idx, np_: int, ns: int, out_dtype) -> torch.Tensor:
    """rows [B*np*ns, Kp] (Kp = 3 + C rounded up to 8): see pq3d_group_rows.  ``feats`` may be a strided row view
    [B, N, >=C] (e.g. the colour columns of an xyz+rgb cloud)."""
    B, N, _ = xyz.shape
    Kp = (3 + feat_channels + 7) // 8 * 8
    out = torch.empty(B * np_ * ns, Kp, dtype=out_dtype, device=xyz.device)
    fs = 0
    if feat_channels:
        assert feats.stride(-1) == 1 and feats.stride(0) == N * feats.stride(1)
        fs = feats.stride(1)
    L.check(L.lib().pq3d_group_rows(L.ptr(xyz), L.ptr(new_xyz), L.ptr(feats) if feat_channels else None,
                                    L.dt_of(feats) if feat_channels else 0, fs, L.ptr(idx), L.ptr(out), L.dt_of(out), B, N,
                                    feat_channels, np_, ns, Kp, L.stream()), "pq3d_group_rows")
    return out


def group_maxpool(rows, G: int, ns: int) -> torch.Tensor:
    C = rows.shape[-1]
    out = torch.empty(G, C, dtype=rows.dtype, device=rows.device)
    L.check(L.lib().pq3d_group_maxpool(L.ptr(rows), L.ptr(out), L.dt_of(rows), G, ns, C, L.stream()), "pq3d_group_maxpool")
    return out


# ------------------------------------------------------------------------------------------------ parameter containers
class _BN(nn.Sequential):
    """pytorch_utils.py:39-63 (_BNBase / BatchNorm2d): child ``bn``."""

    def __init__(self, c):
        super().__init__()
        self.add_module("bn", nn.BatchNorm2d(c))


class _ConvBN(nn.Sequential):
    """pytorch_utils.py:67-190 (Conv2d = 1x1 conv [+ BatchNorm2d] + ReLU, post-activation order)."""

    def __init__(self, cin, cout, bn):
        super().__init__()
        conv = nn.Conv2d(cin, cout, kernel_size=1, bias=not bn)
        nn.init.kaiming_normal_(conv.weight)
        if conv.bias is not None:
            nn.init.constant_(conv.bias, 0)
        self.add_module("conv", conv)
        if bn:
            self.add_module("bn", _BN(cout))


class SharedMLP(nn.Sequential):
    """pytorch_utils.py:11-36."""

    def __init__(self, spec: List[int], bn: bool = True):
        super().__init__()
        for i in range(len(spec) - 1):
            self.add_module(f"layer{i}", _ConvBN(spec[i], spec[i + 1], bn))


class PointnetSAModule(nn.Module):
    """pointnet2_modules.py:125-158 (single-scale set abstraction; npoint=None groups all points)."""

    def __init__(self, *, mlp: List[int], npoint: Optional[int] = None, radius: Optional[float] = None,
                 nsample: Optional[int] = None, bn: bool = True, use_xyz: bool = True):
        super().__init__()
        if not use_xyz:
            raise NotImplementedError("use_xyz=False is not used by the reference's point encoder")
        self.npoint, self.radius, self.nsample = npoint, radius, nsample
        spec = list(mlp)
        spec[0] += 3                         # pointnet2_modules.py:119-120
        self.groupers = nn.ModuleList()      # parameter-free in the reference; kept for module-tree parity
        self.mlps = nn.ModuleList([SharedMLP(spec, bn=bn)])


class PointNetPP(nn.Module):
    """modules/layers/pointnet.py:22-63.  forward(features [M, P, 3 + C]) -> [M, sa_mlps[-1][-1]]."""

    CHUNK = 2048   # clouds per pass (bounds the row buffers: ~0.4 MB of bf16 rows per cloud at the shipped sizes)

    def __init__(self, sa_n_points: list, sa_n_samples: list, sa_radii: list, sa_mlps: list, bn=True, use_xyz=True):
        super().__init__()
        n_sa = len(sa_n_points)
        if not (n_sa == len(sa_n_samples) == len(sa_radii) == len(sa_mlps)):
            raise ValueError("Lens of given hyper-params are not compatible")
        self.encoder = nn.ModuleList(PointnetSAModule(npoint=sa_n_points[i], nsample=sa_n_samples[i], radius=sa_radii[i],
                                                      mlp=sa_mlps[i], bn=bn, use_xyz=use_xyz) for i in range(n_sa))
        out_n_points = sa_n_points[-1] if sa_n_points[-1] is not None else 1
        self.fc = nn.Linear(out_n_points * sa_mlps[-1][-1], sa_mlps[-1][-1])
        self.compute = "bf16"
        self._folded = {}

    @property
    def ct(self) -> int:
        return ops.BF16 if self.compute == "bf16" else ops.F32

    def _fold(self, layer: _ConvBN):
        """Frozen BatchNorm folded into the 1x1 conv: W' = W * g/sqrt(var+eps) (K zero-padded to 8), b' = beta - mean*s
        (+ s*conv.bias).  Cached against the tensors' version counters, so load_state_dict / in-place edits refold."""
        conv = layer.conv
        bn = layer.bn.bn if hasattr(layer, "bn") else None
        src = [conv.weight] + ([conv.bias] if conv.bias is not None else []) + \
              ([bn.weight, bn.bias, bn.running_mean, bn.running_var] if bn is not None else [])
        key = tuple((t.data_ptr(), t._version) for t in src) + (self.ct,)
        hit = self._folded.get(id(layer))
        if hit is not None and hit[0] == key:
            return hit[1], hit[2]
        W = conv.weight.detach()[:, :, 0, 0].float()
        b = conv.bias.detach().float() if conv.bias is not None else torch.zeros(W.shape[0], device=W.device)
        if bn is not None:
            s = bn.weight.detach().float() * torch.rsqrt(bn.running_var.float() + bn.eps)
            W, b = W * s[:, None], bn.bias.detach().float() + (b - bn.running_mean.float()) * s
        Kp = (W.shape[1] + 7) // 8 * 8
        Wp = torch.zeros(W.shape[0], Kp, device=W.device)
        Wp[:, :W.shape[1]] = W
        Wp = Wp.contiguous()
        if self.ct == ops.BF16:   # rounded once here instead of by every row tile (and eligible for the 128x128-tile GEMM)
            Wp = Wp.to(torch.bfloat16)
        self._folded[id(layer)] = (key, Wp, b.contiguous())
        return Wp, b

    def _mlp(self, rows, mlp: SharedMLP):
        ct, ad = self.ct, ops.act_dtype(self.ct)
        for layer in mlp:
            W, b = self._fold(layer)
            rows = ops.linear(rows, W, b, ct=ct, act="relu", out_dtype=ad)
        return rows

    def _forward_chunk(self, pc):
        M, N, D = pc.shape
        ad = ops.act_dtype(self.ct)
        xyz = pc[..., 0:3].contiguous()
        feats, C = (pc[..., 3:], D - 3) if D > 3 else (None, 0)          # a strided view: rows are read in place
        for sa in self.encoder:
            if sa.npoint is not None:
                fps = P2.furthest_point_sample(xyz, sa.npoint)                                   # [M, np] int32
                new_xyz = torch.gather(xyz, 1, fps.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
                idx = P2.ball_query(sa.radius, sa.nsample, xyz, new_xyz)                        # [M, np, ns]
                np_, ns = sa.npoint, sa.nsample
                rows = group_rows(xyz, new_xyz, feats, C, idx, np_, ns, ad)
            else:                                                                              # GroupAll: no centring
                np_, ns, new_xyz = 1, N, None
                rows = group_rows(xyz, None, feats, C, None, 1, N, ad)
            rows = self._mlp(rows, sa.mlps[0])
            C = rows.shape[-1]
            feats = group_maxpool(rows, M * np_, ns).view(M, np_, C)
            xyz, N = new_xyz, np_
        flat = feats.transpose(1, 2).reshape(M, -1) if feats.shape[1] > 1 else feats.reshape(M, -1)   # (C, npoint) order
        return ops.linear(flat, self.fc.weight, self.fc.bias, ct=self.ct)

    def forward(self, features):
        if any(m.training for m in self.modules() if isinstance(m, nn.BatchNorm2d)):
            raise NotImplementedError("PointNetPP on the HIP kernels is the FROZEN tokenizer (BatchNorm in eval mode, "
                                      "ObjectEncoder(freeze_backbone=True)); batch-statistics BatchNorm is not provided")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("training the PointNet++ backbone is not provided: call under torch.no_grad() "
                                      "(freeze_backbone=True) or set requires_grad_(False)")
        if not features.is_cuda:
            raise RuntimeError("pq3d_amd PointNetPP needs a HIP device tensor (no CPU fallback)")
        pc = features.detach().contiguous().float()
        with torch.no_grad():
            outs = [self._forward_chunk(pc[i:i + self.CHUNK]) for i in range(0, pc.shape[0], self.CHUNK)]
        return outs[0] if len(outs) == 1 else torch.cat(outs, 0)


POINTNETPP_TOKENIZER = dict(sa_n_points=[32, 16, None], sa_n_samples=[32, 32, None], sa_radii=[0.2, 0.4, None],
                            sa_mlps=[[3, 64, 64, 128], [128, 128, 128, 256], [256, 256, 512, 768]])   # object_encoder.py:23-28
