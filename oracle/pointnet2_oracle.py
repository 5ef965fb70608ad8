"""CPU oracle of the PointNet++ point-set operators.  TEST INFRASTRUCTURE ONLY.

numpy restatement of the reference's CUDA kernels (modules/third_party/pointnet2/_ext_src/src/*.cu, file:line on each
function), written from their behaviour: fp32 arithmetic in the kernels' operation order.  PARITY UNPINNED by execution:
the reference's extension is CUDA-only (its CPU path is AT_ASSERT(false), ball_query.cpp:28 etc.) and cannot run in the
build container, and its only test (pointnet2_test.py:18-30) is a gradcheck of three_interpolate at atol 1e-1, which
tests/test_pointnet2.py repeats.  Tie-breaking of furthest point sampling (duplicated points) follows this file
(smallest index), not the reference's block-size-dependent reduction tree."""
from __future__ import annotations

import numpy as np


def furthest_point_sampling(xyz: np.ndarray, m: int) -> np.ndarray:
    """sampling_gpu.cu:70-172 (+ sampling.cpp: temp = 1e10): xyz [B,N,3] fp32 -> idx [B,m] int32."""
    xyz = xyz.astype(np.float32)
    B, N, _ = xyz.shape
    out = np.zeros((B, m), dtype=np.int32)
    for b in range(B):
        p = xyz[b]
        live = (p[:, 0] * p[:, 0] + p[:, 1] * p[:, 1] + p[:, 2] * p[:, 2]) > np.float32(1e-3)
        temp = np.full(N, 1e10, dtype=np.float32)
        old = 0
        for j in range(1, m):
            d = (p[:, 0] - p[old, 0]) ** 2 + (p[:, 1] - p[old, 1]) ** 2 + (p[:, 2] - p[old, 2]) ** 2
            temp = np.where(live, np.minimum(d.astype(np.float32), temp), temp)
            cand = np.where(live, temp, np.float32(-1.0))
            old = int(np.argmax(cand)) if cand.max() > -1.0 else 0     # first maximum = smallest index
            out[b, j] = old
    return out


def ball_query(new_xyz: np.ndarray, xyz: np.ndarray, radius: float, nsample: int) -> np.ndarray:
    """ball_query_gpu.cu:9-44: idx [B,M,nsample] int32 (zeros where no point is inside the ball)."""
    new_xyz, xyz = new_xyz.astype(np.float32), xyz.astype(np.float32)
    B, M, _ = new_xyz.shape
    idx = np.zeros((B, M, nsample), dtype=np.int32)
    r2 = np.float32(radius) * np.float32(radius)
    for b in range(B):
        for j in range(M):
            c = new_xyz[b, j]
            d2 = (c[0] - xyz[b, :, 0]) ** 2 + (c[1] - xyz[b, :, 1]) ** 2 + (c[2] - xyz[b, :, 2]) ** 2
            hits = np.nonzero(d2 < r2)[0][:nsample]
            if len(hits):
                idx[b, j, :] = hits[0]
                idx[b, j, :len(hits)] = hits
    return idx


def gather_points(points: np.ndarray, idx: np.ndarray) -> np.ndarray:
    """sampling_gpu.cu:8-20: points [B,C,N], idx [B,M] -> [B,C,M]."""
    return np.take_along_axis(points, np.broadcast_to(idx[:, None, :], (points.shape[0], points.shape[1], idx.shape[1])), 2)


def gather_points_grad(grad_out: np.ndarray, idx: np.ndarray, N: int) -> np.ndarray:
    """sampling_gpu.cu:34-47."""
    B, C, M = grad_out.shape
    g = np.zeros((B, C, N), dtype=np.float64)
    for b in range(B):
        np.add.at(g[b].T, idx[b], grad_out[b].T)
    return g.astype(np.float32)


def group_points(points: np.ndarray, idx: np.ndarray) -> np.ndarray:
    """group_points_gpu.cu:8-30: points [B,C,N], idx [B,np,ns] -> [B,C,np,ns]."""
    B, npnt, ns = idx.shape
    return gather_points(points, idx.reshape(B, -1)).reshape(B, points.shape[1], npnt, ns)


def three_nn(unknown: np.ndarray, known: np.ndarray):
    """interpolate_gpu.cu:9-58: SQUARED distances [B,n,3] and indices [B,n,3] of the three nearest known points
    (strict '<' updates, i.e. the earliest index wins ties)."""
    unknown, known = unknown.astype(np.float32), known.astype(np.float32)
    d = ((unknown[:, :, None, 0] - known[:, None, :, 0]) ** 2 + (unknown[:, :, None, 1] - known[:, None, :, 1]) ** 2
         + (unknown[:, :, None, 2] - known[:, None, :, 2]) ** 2).astype(np.float32)
    idx = np.argsort(d, axis=2, kind="stable")[:, :, :3].astype(np.int32)
    return np.take_along_axis(d, idx, 2), idx


def three_interpolate(points: np.ndarray, idx: np.ndarray, weight: np.ndarray) -> np.ndarray:
    """interpolate_gpu.cu:72-101: points [B,c,m], idx / weight [B,n,3] -> [B,c,n]."""
    B, c, m = points.shape
    out = np.zeros((B, c, idx.shape[1]), dtype=np.float32)
    for k in range(3):
        out += gather_points(points, idx[:, :, k]) * weight[:, None, :, k]
    return out


# ------------------------------------------------------------------------------------------------ the tokenizer network
def pointnetpp_forward(sd, pc: np.ndarray, sa_n_points, sa_n_samples, sa_radii, n_layers=3, eps: float = 1e-5,
                       prefix: str = "", trace: dict | None = None) -> np.ndarray:
    """PointNetPP.forward (modules/layers/pointnet.py:55-63) with frozen BatchNorm: per stage furthest-point sampling +
    gather of the centres (pointnet2_modules.py:48-53), QueryAndGroup / GroupAll with use_xyz
    (pointnet2_utils.py:331-358, 396-419), SharedMLP = [1x1 conv (no bias) -> BatchNorm2d(eval) -> ReLU] x 3
    (pytorch_utils.py:11-36), max over the samples (pointnet2_modules.py:62-65); then fc on the flattened
    [C, npoint] features.  fp64 accumulation in the matmuls is avoided on purpose: fp32 like the reference.
    ``sd``: the module's state_dict as numpy arrays.  The learnable layers of this function ARE pinned (fixture F12 runs
    the reference's own SharedMLP / fc modules); the point-set operators feeding them are the unpinned ones above."""
    pc = pc.astype(np.float32)
    xyz, feats = pc[..., :3], np.transpose(pc[..., 3:], (0, 2, 1)) if pc.shape[-1] > 3 else None     # [B,N,3], [B,C,N]
    for i, (npoint, nsample, radius) in enumerate(zip(sa_n_points, sa_n_samples, sa_radii)):
        if npoint is not None:
            fps = furthest_point_sampling(xyz, npoint)
            new_xyz = np.take_along_axis(xyz, fps[..., None].astype(np.int64), 1)
            idx = ball_query(new_xyz, xyz, radius, nsample)
            gx = group_points(np.ascontiguousarray(np.transpose(xyz, (0, 2, 1))), idx)               # [B,3,np,ns]
            gx = gx - np.transpose(new_xyz, (0, 2, 1))[..., None]
            x = np.concatenate([gx, group_points(np.ascontiguousarray(feats), idx)], 1) if feats is not None else gx
            if trace is not None:
                trace[f"fps/{i}"], trace[f"ball/{i}"] = fps, idx
        else:
            new_xyz = None
            gx = np.transpose(xyz, (0, 2, 1))[:, :, None, :]                                          # [B,3,1,N]
            x = np.concatenate([gx, feats[:, :, None, :]], 1) if feats is not None else gx
        for j in range(n_layers):
            k = f"{prefix}encoder.{i}.mlps.0.layer{j}."
            W = sd[k + "conv.weight"][:, :, 0, 0].astype(np.float32)
            x = np.einsum("oc,bcps->bops", W, x).astype(np.float32)
            if (k + "conv.bias") in sd:
                x = x + sd[k + "conv.bias"][None, :, None, None]
            if (k + "bn.bn.weight") in sd:
                inv = (1.0 / np.sqrt(sd[k + "bn.bn.running_var"].astype(np.float32) + np.float32(eps))).astype(np.float32)
                x = (x - sd[k + "bn.bn.running_mean"][None, :, None, None]) * inv[None, :, None, None] \
                    * sd[k + "bn.bn.weight"][None, :, None, None] + sd[k + "bn.bn.bias"][None, :, None, None]
            x = np.maximum(x, 0).astype(np.float32)
        feats = x.max(-1)                                                                             # [B,C,np]
        if trace is not None:
            trace[f"pooled/{i}"] = feats
        xyz = new_xyz
    flat = feats.reshape(feats.shape[0], -1)
    return (flat @ sd[prefix + "fc.weight"].T.astype(np.float32) + sd[prefix + "fc.bias"]).astype(np.float32)
