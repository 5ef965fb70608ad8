"""PointNet++ operators: HIP kernels (pq3d_amd/pointnet2.py) against the numpy oracle (oracle/pointnet2_oracle.py) --
indices bit-exact, interpolation / gradients to fp32 rounding -- plus the reference's own test
(modules/third_party/pointnet2/pointnet2_test.py:18-30: gradcheck of three_interpolate)."""
import numpy as np
import pytest
import torch

from oracle import pointnet2_oracle as PO

DEV = "cuda"


def cloud(B, N, seed, pad_zero=0):
    r = np.random.default_rng(seed)
    p = r.uniform(-1, 1, (B, N, 3)).astype(np.float32)
    if pad_zero:
        p[:, -pad_zero:] = 0.0          # zero-padded points: the sampler must skip them (|p|^2 <= 1e-3)
    return p


def test_oracle_fps_and_ball_query_basic_properties():
    p = cloud(2, 300, 0, pad_zero=20)
    idx = PO.furthest_point_sampling(p, 64)
    assert idx.shape == (2, 64) and (idx[:, 0] == 0).all()
    for b in range(2):
        assert len(set(idx[b].tolist())) == 64 and idx[b].max() < 280       # distinct, never a padded point
    bq = PO.ball_query(p[:, :10], p, 0.4, 16)
    for b in range(2):
        for j in range(10):
            d = np.linalg.norm(p[b] - p[b, j], axis=1)
            assert (d[bq[b, j]] < 0.4).all() and bq[b, j, 0] == np.nonzero(d < 0.4)[0][0]


@pytest.mark.gpu
@pytest.mark.parametrize("N,M", [(1024, 256), (700, 128), (3000, 512), (8192, 64)])
def test_furthest_point_sampling_matches_oracle(N, M):
    from pq3d_amd import pointnet2 as P
    p = cloud(3, N, N + M, pad_zero=N // 10)
    got = P.furthest_point_sample(torch.from_numpy(p).to(DEV), M).cpu().numpy()
    assert np.array_equal(got, PO.furthest_point_sampling(p, M))


@pytest.mark.gpu
@pytest.mark.parametrize("radius,nsample", [(0.2, 16), (0.4, 64), (0.05, 32), (3.0, 100)])
def test_ball_query_and_grouping_match_oracle(radius, nsample):
    from pq3d_amd import pointnet2 as P
    p = cloud(2, 1500, 7)
    centers = p[:, ::11][:, :97].copy()
    centers[0, 3] = 50.0                      # a centre with an empty ball
    t, c = torch.from_numpy(p).to(DEV), torch.from_numpy(centers).to(DEV)
    idx = P.ball_query(radius, nsample, t, c)
    ref = PO.ball_query(centers, p, radius, nsample)
    assert np.array_equal(idx.cpu().numpy(), ref)
    feats = torch.randn(2, 5, 1500, device=DEV, requires_grad=True)
    g = P.grouping_operation(feats, idx)
    assert np.array_equal(g.detach().cpu().numpy(), PO.group_points(feats.detach().cpu().numpy(), ref))
    w = torch.randn_like(g)
    g.backward(w)
    rg = PO.gather_points_grad(w.cpu().numpy().reshape(2, 5, -1), ref.reshape(2, -1), 1500)
    assert np.abs(feats.grad.cpu().numpy() - rg).max() <= 1e-4 * max(1.0, np.abs(rg).max())


@pytest.mark.gpu
def test_gather_three_nn_and_interpolate_match_oracle():
    from pq3d_amd import pointnet2 as P
    known, unknown = cloud(2, 333, 1), cloud(2, 2000, 2)
    tk, tu = torch.from_numpy(known).to(DEV), torch.from_numpy(unknown).to(DEV)
    dist, idx = P.three_nn(tu, tk)
    d2, ridx = PO.three_nn(unknown, known)
    assert np.array_equal(idx.cpu().numpy(), ridx)
    assert np.abs(dist.cpu().numpy() - np.sqrt(d2)).max() <= 1e-6
    w = 1.0 / (dist + 1e-8)
    w = (w / w.sum(2, keepdim=True)).contiguous()
    feats = torch.randn(2, 7, 333, device=DEV, requires_grad=True)
    out = P.three_interpolate(feats, idx, w)
    ref = PO.three_interpolate(feats.detach().cpu().numpy(), ridx, w.cpu().numpy())
    assert np.abs(out.detach().cpu().numpy() - ref).max() <= 1e-5
    fidx = P.furthest_point_sample(tk, 50)
    gat = P.gather_operation(feats, fidx)
    assert np.array_equal(gat.detach().cpu().numpy(), PO.gather_points(feats.detach().cpu().numpy(), fidx.cpu().numpy()))
    (out.sum() * 2 + gat.sum() * 3).backward()
    ref_g = np.zeros((2, 7, 333), dtype=np.float64)
    for b in range(2):
        for k in range(3):
            np.add.at(ref_g[b].T, ridx[b, :, k], np.repeat((2 * w[b, :, k].cpu().numpy())[:, None], 7, 1))
        np.add.at(ref_g[b].T, fidx[b].cpu().numpy(), 3.0)
    assert np.abs(feats.grad.cpu().numpy() - ref_g).max() <= 1e-3 * np.abs(ref_g).max()


@pytest.mark.gpu
def test_interpolation_grad_like_the_reference_test():
    """pointnet2_test.py:18-30 (atol / rtol 1e-1 there; fp32 kernels, so gradcheck runs with a large eps)."""
    from torch.autograd import gradcheck
    from pq3d_amd import pointnet2 as P
    feats = torch.randn(1, 2, 4, device=DEV).float().requires_grad_(True)
    idx = torch.tensor([[[0, 1, 2], [1, 2, 3]]], dtype=torch.int32, device=DEV)
    weight = torch.tensor([[[1., 1., 1.], [2., 2., 2.]]], device=DEV)
    assert gradcheck(lambda x: P.three_interpolate(x, idx, weight), feats, eps=1e-2, atol=1e-1, rtol=1e-1)


@pytest.mark.gpu
def test_query_and_group_module():
    from pq3d_amd import pointnet2 as P
    p = torch.from_numpy(cloud(2, 600, 4)).to(DEV)
    new = P.gather_operation(p.transpose(1, 2).contiguous(), P.furthest_point_sample(p, 40)).transpose(1, 2).contiguous()
    feats = torch.randn(2, 6, 600, device=DEV)
    out = P.QueryAndGroup(0.3, 24, use_xyz=True)(p, new, feats)
    assert out.shape == (2, 9, 40, 24)
    assert float(out[:, :3].norm(dim=1).max()) < 0.3 + 1e-5     # grouped xyz are offsets inside the ball


# ---------------------------------------------------------------------------------------------- pinned by the reference's kernels
def _f18():
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "F18_pointnet2_ref.npz")
    if not os.path.exists(path):
        pytest.skip("F18_pointnet2_ref.npz not generated yet (oracle/run_ref_pointnet2.py on a GPU box)")
    return np.load(path)


def test_oracle_matches_the_reference_kernels_fixture():
    """F18 = outputs of the REFERENCE's own _ext_src kernels (built for gfx950 by oracle/build_ref_pointnet2.py, executed by
    oracle/run_ref_pointnet2.py on an MI355X): the numpy oracle reproduces every index bit for bit and the float outputs
    to fp32 rounding -- SURVEY 8f-4 pinned by execution."""
    from oracle import run_ref_pointnet2 as R
    z = _f18()
    for name, kind, p in R.cases():
        if kind == "fps":
            pts = R.cloud(p["B"], p["N"], p["seed"], p["pad_zero"])
            assert np.array_equal(PO.furthest_point_sampling(pts, p["M"]), z[name + "/idx"]), name
        elif kind == "ball":
            pts, centers, feats, gout = R.ball_inputs(p)
            idx = PO.ball_query(centers, pts, p["radius"], p["nsample"])
            assert np.array_equal(idx, z[name + "/idx"]), name
            assert np.array_equal(PO.group_points(feats, idx), z[name + "/grouped"]), name
            rg = PO.gather_points_grad(gout.reshape(p["B"], p["C"], -1), idx.reshape(p["B"], -1), p["N"])
            assert np.abs(rg - z[name + "/grouped_grad"]).max() <= 1e-4 * max(1.0, np.abs(rg).max()), name
        else:
            known, unknown, feats, gout, ggat = R.interp_inputs(p)
            d2, idx = PO.three_nn(unknown, known)
            assert np.array_equal(idx, z[name + "/idx"]), name
            assert np.abs(d2 - z[name + "/dist2"]).max() <= 1e-6
            out = PO.three_interpolate(feats, idx, z[name + "/weight"])
            assert np.abs(out - z[name + "/interp"]).max() <= 1e-5
            fidx = PO.furthest_point_sampling(known, p["M"])
            assert np.array_equal(fidx, z[name + "/fps_idx"]), name
            assert np.array_equal(PO.gather_points(feats, fidx), z[name + "/gathered"]), name


@pytest.mark.gpu
def test_hip_kernels_match_the_reference_kernels_fixture():
    """pq3d_amd/csrc/pointnet2.hip against F18 (the reference's kernels' outputs): indices bit-exact, gathers bit-exact,
    interpolation / scatter-add gradients to fp32 summation order."""
    from oracle import run_ref_pointnet2 as R
    from pq3d_amd import pointnet2 as P
    z = _f18()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    for name, kind, p in R.cases():
        if kind == "fps":
            pts = R.cloud(p["B"], p["N"], p["seed"], p["pad_zero"])
            assert np.array_equal(P.furthest_point_sample(t(pts), p["M"]).cpu().numpy(), z[name + "/idx"]), name
        elif kind == "ball":
            pts, centers, feats, gout = R.ball_inputs(p)
            idx = P.ball_query(p["radius"], p["nsample"], t(pts), t(centers))
            assert np.array_equal(idx.cpu().numpy(), z[name + "/idx"]), name
            f = t(feats).requires_grad_(True)
            g = P.grouping_operation(f, idx)
            assert np.array_equal(g.detach().cpu().numpy(), z[name + "/grouped"]), name
            g.backward(t(gout))
            ref = z[name + "/grouped_grad"]
            assert np.abs(f.grad.cpu().numpy() - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), name
        else:
            known, unknown, feats, gout, ggat = R.interp_inputs(p)
            dist, idx = P.three_nn(t(unknown), t(known))
            assert np.array_equal(idx.cpu().numpy(), z[name + "/idx"]), name
            assert np.abs(dist.cpu().numpy() ** 2 - z[name + "/dist2"]).max() <= 1e-5
            f = t(feats).requires_grad_(True)
            out = P.three_interpolate(f, idx, t(z[name + "/weight"]))
            assert np.abs(out.detach().cpu().numpy() - z[name + "/interp"]).max() <= 1e-5
            out.backward(t(gout))
            ref = z[name + "/interp_grad"]
            assert np.abs(f.grad.cpu().numpy() - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())
            fidx = P.furthest_point_sample(t(known), p["M"])
            assert np.array_equal(fidx.cpu().numpy(), z[name + "/fps_idx"]), name
            f2 = t(feats).requires_grad_(True)
            gat = P.gather_operation(f2, fidx)
            assert np.array_equal(gat.detach().cpu().numpy(), z[name + "/gathered"]), name
            gat.backward(t(ggat))
            ref = z[name + "/gathered_grad"]
            assert np.abs(f2.grad.cpu().numpy() - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())
