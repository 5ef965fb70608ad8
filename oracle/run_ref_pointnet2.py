"""TEST INFRASTRUCTURE (oracle side): executes the REFERENCE's own PointNet++ kernels (oracle/_ref/pq3d_ref_pointnet2.so,
built from /root/reference by oracle/build_ref_pointnet2.py) on a GPU and writes the golden vectors
tests/golden/F18_pointnet2_ref.npz -- inputs are regenerated from the seeds stored next to the outputs.

    python oracle/run_ref_pointnet2.py [out.npz]        # on a box with a GPU (gpurun); default out: gpurun_out/F18_pointnet2_ref.npz

Cases (shared with tests/test_pointnet2.py through `cases()` / `cloud()`): furthest point sampling incl. zero-padded
points, ball query incl. an empty ball and a ball larger than nsample, grouping + its gradient, gather + its gradient,
three_nn, three_interpolate + its gradient."""
from __future__ import annotations

import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def cloud(B, N, seed, pad_zero=0):
    r = np.random.default_rng(seed)
    p = r.uniform(-1, 1, (B, N, 3)).astype(np.float32)
    if pad_zero:
        p[:, -pad_zero:] = 0.0          # zero-padded points: the sampler skips them (|p|^2 <= 1e-3, sampling_gpu.cu)
    return p


def cases():
    """(name, kind, parameters) -- sizes the reference finishes in milliseconds; the fixture stays small."""
    return [
        ("fps/a", "fps", dict(B=3, N=1024, M=256, seed=11, pad_zero=100)),
        ("fps/b", "fps", dict(B=2, N=700, M=128, seed=12, pad_zero=0)),
        ("fps/c", "fps", dict(B=2, N=3000, M=512, seed=13, pad_zero=300)),
        ("ball/a", "ball", dict(B=2, N=1500, seed=7, radius=0.2, nsample=16, C=5)),
        ("ball/b", "ball", dict(B=2, N=1500, seed=8, radius=0.4, nsample=64, C=3)),
        ("ball/c", "ball", dict(B=2, N=1500, seed=9, radius=3.0, nsample=100, C=2)),
        ("interp/a", "interp", dict(B=2, Nk=333, Nu=2000, seed=1, C=7, M=50)),
    ]


def ball_inputs(p):
    pts = cloud(p["B"], p["N"], p["seed"])
    centers = pts[:, ::11][:, :97].copy()
    centers[0, 3] = 50.0                      # a centre with an empty ball
    r = np.random.default_rng(p["seed"] + 100)
    feats = r.standard_normal((p["B"], p["C"], p["N"])).astype(np.float32)
    gout = r.standard_normal((p["B"], p["C"], centers.shape[1], p["nsample"])).astype(np.float32)
    return pts, centers, feats, gout


def interp_inputs(p):
    known, unknown = cloud(p["B"], p["Nk"], p["seed"]), cloud(p["B"], p["Nu"], p["seed"] + 1)
    r = np.random.default_rng(p["seed"] + 200)
    feats = r.standard_normal((p["B"], p["C"], p["Nk"])).astype(np.float32)
    gout = r.standard_normal((p["B"], p["C"], p["Nu"])).astype(np.float32)
    ggat = r.standard_normal((p["B"], p["C"], p["M"])).astype(np.float32)
    return known, unknown, feats, gout, ggat


def main(out_path):
    import torch
    so = os.path.join(HERE, "_ref", "pq3d_ref_pointnet2.so")
    spec = importlib.util.spec_from_file_location("pq3d_ref_pointnet2", so)
    ext = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ext)
    dev = "cuda"
    t = lambda a: torch.from_numpy(a).to(dev)
    out = {}
    for name, kind, p in cases():
        if kind == "fps":
            pts = cloud(p["B"], p["N"], p["seed"], p["pad_zero"])
            out[name + "/idx"] = ext.furthest_point_sampling(t(pts), p["M"]).cpu().numpy()
        elif kind == "ball":
            pts, centers, feats, gout = ball_inputs(p)
            idx = ext.ball_query(t(centers), t(pts), p["radius"], p["nsample"])
            out[name + "/idx"] = idx.cpu().numpy()
            out[name + "/grouped"] = ext.group_points(t(feats), idx).cpu().numpy()
            out[name + "/grouped_grad"] = ext.group_points_grad(t(gout), idx, p["N"]).cpu().numpy()
        else:
            known, unknown, feats, gout, ggat = interp_inputs(p)
            d2, idx = ext.three_nn(t(unknown), t(known))
            out[name + "/dist2"], out[name + "/idx"] = d2.cpu().numpy(), idx.cpu().numpy()
            dist = torch.sqrt(d2)
            w = 1.0 / (dist + 1e-8)
            w = (w / w.sum(2, keepdim=True)).contiguous()
            out[name + "/weight"] = w.cpu().numpy()
            out[name + "/interp"] = ext.three_interpolate(t(feats), idx, w).cpu().numpy()
            out[name + "/interp_grad"] = ext.three_interpolate_grad(t(gout), idx, w, p["Nk"]).cpu().numpy()
            fidx = ext.furthest_point_sampling(t(known), p["M"])
            out[name + "/fps_idx"] = fidx.cpu().numpy()
            out[name + "/gathered"] = ext.gather_points(t(feats), fidx).cpu().numpy()
            out[name + "/gathered_grad"] = ext.gather_points_grad(t(ggat), fidx, p["Nk"]).cpu().numpy()
    torch.cuda.synchronize()
    out["meta/device"] = np.array(torch.cuda.get_device_name(0))
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    np.savez_compressed(out_path, **out)
    print(f"wrote {out_path}: {len(out)} arrays, {os.path.getsize(out_path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "..", "gpurun_out", "F18_pointnet2_ref.npz"))
