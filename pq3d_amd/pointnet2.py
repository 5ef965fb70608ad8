"""PointNet++ point-set operators on the HIP kernels with the names, argument order and tensor
layouts of the reference's ``pointnet2_utils`` (modules/third_party/pointnet2/pointnet2_utils.py:48-419), so that
``pointnet2_modules`` can import this module in place of ``pointnet2._ext``-backed utils.  No CPU fallback."""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib as L


def _f(t):
    return t.contiguous().float()


class FurthestPointSampling(Function):
    """pointnet2_utils.py:48-77: xyz [B,N,3] -> int32 idx [B,npoint]."""

    @staticmethod
    def forward(ctx, xyz, npoint):
        xyz = _f(xyz)
        B, N, _ = xyz.shape
        idx = torch.empty(B, npoint, dtype=torch.int32, device=xyz.device)
        L.check(L.lib().pq3d_furthest_point_sampling(L.ptr(xyz), L.ptr(idx), B, N, npoint, L.stream()),
                "pq3d_furthest_point_sampling")
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    """pointnet2_utils.py:80-114: features [B,C,N], idx [B,npoint] -> [B,C,npoint]."""

    @staticmethod
    def forward(ctx, features, idx):
        features, idx = _f(features), idx.contiguous().int()
        B, C, N = features.shape
        out = torch.empty(B, C, idx.shape[1], dtype=torch.float32, device=features.device)
        L.check(L.lib().pq3d_gather_points(L.ptr(features), L.ptr(idx), L.ptr(out), B, C, N, idx.shape[1], L.stream()),
                "pq3d_gather_points")
        ctx.for_backwards = (idx, C, N)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, C, N = ctx.for_backwards
        g = _f(grad_out)
        B = g.shape[0]
        gf = torch.empty(B, C, N, dtype=torch.float32, device=g.device)
        L.check(L.lib().pq3d_gather_points_grad(L.ptr(g), L.ptr(idx), L.ptr(gf), B, C, N, idx.shape[1], L.stream()),
                "pq3d_gather_points_grad")
        return gf, None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    """pointnet2_utils.py:117-146: (unknown [B,n,3], known [B,m,3]) -> (dist [B,n,3] (sqrt of squared), idx [B,n,3])."""

    @staticmethod
    def forward(ctx, unknown, known):
        unknown, known = _f(unknown), _f(known)
        B, n, _ = unknown.shape
        d2 = torch.empty(B, n, 3, dtype=torch.float32, device=unknown.device)
        idx = torch.empty(B, n, 3, dtype=torch.int32, device=unknown.device)
        L.check(L.lib().pq3d_three_nn(L.ptr(unknown), L.ptr(known), L.ptr(d2), L.ptr(idx), B, n, known.shape[1], L.stream()),
                "pq3d_three_nn")
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(d2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    """pointnet2_utils.py:149-205: features [B,c,m], idx [B,n,3], weight [B,n,3] -> [B,c,n]."""

    @staticmethod
    def forward(ctx, features, idx, weight):
        features, idx, weight = _f(features), idx.contiguous().int(), _f(weight)
        B, c, m = features.shape
        n = idx.shape[1]
        out = torch.empty(B, c, n, dtype=torch.float32, device=features.device)
        L.check(L.lib().pq3d_three_interpolate(L.ptr(features), L.ptr(idx), L.ptr(weight), L.ptr(out), B, c, m, n, L.stream()),
                "pq3d_three_interpolate")
        ctx.three_interpolate_for_backward = (idx, weight, m)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight, m = ctx.three_interpolate_for_backward
        g = _f(grad_out)
        B, c, n = g.shape
        gf = torch.empty(B, c, m, dtype=torch.float32, device=g.device)
        L.check(L.lib().pq3d_three_interpolate_grad(L.ptr(g), L.ptr(idx), L.ptr(weight), L.ptr(gf), B, c, m, n, L.stream()),
                "pq3d_three_interpolate_grad")
        return gf, None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    """pointnet2_utils.py:208-256: features [B,C,N], idx [B,npoint,nsample] -> [B,C,npoint,nsample]."""

    @staticmethod
    def forward(ctx, features, idx):
        features, idx = _f(features), idx.contiguous().int()
        B, C, N = features.shape
        _, npoint, nsample = idx.shape
        out = torch.empty(B, C, npoint, nsample, dtype=torch.float32, device=features.device)
        L.check(L.lib().pq3d_gather_points(L.ptr(features), L.ptr(idx), L.ptr(out), B, C, N, npoint * nsample, L.stream()),
                "pq3d_gather_points")
        ctx.for_backwards = (idx, N)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, N = ctx.for_backwards
        g = _f(grad_out)
        B, C = g.shape[:2]
        gf = torch.empty(B, C, N, dtype=torch.float32, device=g.device)
        L.check(L.lib().pq3d_gather_points_grad(L.ptr(g), L.ptr(idx), L.ptr(gf), B, C, N, idx.shape[1] * idx.shape[2],
                                                L.stream()), "pq3d_gather_points_grad")
        return gf, None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    """pointnet2_utils.py:259-290: (radius, nsample, xyz [B,N,3], new_xyz [B,npoint,3]) -> int32 idx [B,npoint,nsample]."""

    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        xyz, new_xyz = _f(xyz), _f(new_xyz)
        B, N, _ = xyz.shape
        M = new_xyz.shape[1]
        idx = torch.empty(B, M, nsample, dtype=torch.int32, device=xyz.device)
        L.check(L.lib().pq3d_ball_query(L.ptr(new_xyz), L.ptr(xyz), L.ptr(idx), B, N, M, float(radius), int(nsample),
                                        L.stream()), "pq3d_ball_query")
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


class QueryAndGroup(nn.Module):
    """pointnet2_utils.py:293-372 (ball query + grouping, optional xyz concatenation / normalisation)."""

    def __init__(self, radius, nsample, use_xyz=True, ret_grouped_xyz=False, normalize_xyz=False, sample_uniformly=False,
                 ret_unique_cnt=False):
        super().__init__()
        if sample_uniformly or ret_unique_cnt:
            raise NotImplementedError("sample_uniformly / ret_unique_cnt are host-side loops in the reference and unused "
                                      "by its shipped point encoder")
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz
        self.ret_grouped_xyz, self.normalize_xyz = ret_grouped_xyz, normalize_xyz

    def forward(self, xyz, new_xyz, features=None):
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        xyz_trans = xyz.transpose(1, 2).contiguous()
        grouped_xyz = grouping_operation(xyz_trans, idx)
        grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if self.normalize_xyz:
            grouped_xyz = grouped_xyz / self.radius
        if features is not None:
            grouped = grouping_operation(features, idx)
            new_features = torch.cat([grouped_xyz, grouped], dim=1) if self.use_xyz else grouped
        else:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            new_features = grouped_xyz
        return (new_features, grouped_xyz) if self.ret_grouped_xyz else new_features


class GroupAll(nn.Module):
    """pointnet2_utils.py:375-419."""

    def __init__(self, use_xyz=True, ret_grouped_xyz=False):
        super().__init__()
        self.use_xyz, self.ret_grouped_xyz = use_xyz, ret_grouped_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is not None:
            grouped = features.unsqueeze(2)
            new_features = torch.cat([grouped_xyz, grouped], dim=1) if self.use_xyz else grouped
        else:
            new_features = grouped_xyz
        return (new_features, grouped_xyz) if self.ret_grouped_xyz else new_features
