"""GPU: segment pooling as bandwidth kernels (pq3d_amd/csrc/segment.hip) through the C ABI.

torch_scatter is un-vendored (parity unpinned by execution): the checker is the oracle's restatement of its published
definition (oracle/pq3d_oracle.py: scatter_mean, multiscale_segment_pool; pinned to a literal loop in
tests/test_oracle_golden.py).  Properties checked here at sizes up to the bench's: exact equality on integer-valued inputs
(any summation order gives the same fp32 result, so every grouping / piece / partial-slot mistake shows as a wrong integer),
fp64 agreement on random inputs, bit-identical results run to run, the giant-segment and many-tiny-segment regimes, row widths
on both the 16-byte-vector and the scalar path, ids out of range, empty inputs."""
import numpy as np
import pytest
import torch

from oracle import pq3d_oracle as O
from pq3d_amd import _lib as L
from pq3d_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ids(kind, N, S, g):
    if kind == "uniform":
        return torch.randint(0, S, (N,), generator=g)
    if kind == "giant":            # one segment owns 60 % of the voxels (a floor), the rest is uniform
        idx = torch.randint(0, S, (N,), generator=g)
        idx[torch.rand(N, generator=g) < 0.6] = S // 3
        return idx
    if kind == "sorted":
        return torch.sort(torch.randint(0, S, (N,), generator=g)).values
    if kind == "skewed":           # geometric segment sizes, many empty segments
        return (torch.rand(N, generator=g).pow(4) * S).long().clamp_(max=S - 1)
    if kind == "invalid":          # 10 % of the ids out of range on both sides
        idx = torch.randint(0, S, (N,), generator=g)
        bad = torch.rand(N, generator=g)
        idx[bad < 0.05] = -1 - torch.randint(0, 5, (int((bad < 0.05).sum()),), generator=g)
        idx[bad > 0.95] = S + torch.randint(0, 5, (int((bad > 0.95).sum()),), generator=g)
        return idx
    raise ValueError(kind)


def _ref_mean(src, idx, S):
    ok = (idx >= 0) & (idx < S)
    out = torch.zeros(S, src.shape[1], dtype=torch.float64)
    out.index_add_(0, idx[ok], src[ok].double())
    cnt = torch.bincount(idx[ok], minlength=S).double()
    return out / cnt.clamp(min=1)[:, None], cnt


@pytest.mark.parametrize("kind", ["uniform", "giant", "sorted", "skewed", "invalid"])
@pytest.mark.parametrize("N,S,C", [(1, 1, 4), (63, 5, 96), (130, 3, 256), (5000, 300, 96), (20011, 1000, 128), (70000, 2048, 256),
                                   (9000, 70000, 8), (3000, 17, 3), (4097, 33, 257), (2500, 40, 768), (2500, 40, 1028),
                                   (1500, 9, 1300)])
def test_segment_mean_exact_on_integer_inputs(kind, N, S, C):
    """Integer-valued fp32 rows (|x| <= 8, sums < 2^24): sums are exact in any order, so the kernel's sums and counts must
    EQUAL the reference's; the mean then differs by at most the one rounding of the division."""
    g = torch.Generator().manual_seed(N * 7 + S + C)
    idx = _ids(kind, N, S, g)
    src = torch.randint(-8, 9, (N, C), generator=g).float()
    plan = ops.SegmentPlan(idx.to(DEV), S)
    tot, cnt = plan.reduce(src.to(DEV), None, None, C, False)
    ok = (idx >= 0) & (idx < S)
    want = torch.zeros(S, C).index_add_(0, idx[ok], src[ok])
    assert torch.equal(tot.cpu(), want), "segment sums differ"
    assert torch.equal(cnt.cpu(), torch.bincount(idx[ok], minlength=S).float()), "segment counts differ"
    mean = ops.scatter_mean(src.to(DEV), idx.to(DEV), S, plan=plan)
    assert torch.equal(mean.cpu(), want / cnt.cpu().clamp(min=1)[:, None])


@pytest.mark.parametrize("kind", ["uniform", "giant"])
@pytest.mark.parametrize("C", [96, 128, 256])
def test_segment_mean_matches_oracle_and_is_bit_identical_run_to_run(kind, C):
    """Random fp32 features at a bench-like size: within fp32 summation error of the fp64 definition and of the oracle,
    forward and backward, and the very same bits on every run (no atomics anywhere)."""
    N, S = 200_000, 3000
    g = torch.Generator().manual_seed(C)
    idx = _ids(kind, N, S, g)
    src = torch.randn(N, C, generator=g)
    sd = src.to(DEV).requires_grad_(True)
    out = ops.scatter_mean(sd, idx.to(DEV), S)
    ref, cnt = _ref_mean(src, idx, S)
    err = float((out.detach().cpu().double() - ref).abs().max())
    assert err <= 2e-6 * max(1.0, float(ref.abs().max())), err
    outo = O.scatter_mean(src, idx, S)
    assert float((out.detach().cpu() - outo).abs().max()) <= 1e-5
    gy = torch.randn(S, C, generator=g)
    out.backward(gy.to(DEV))
    gref = (gy.double() / cnt.clamp(min=1)[:, None])[idx]
    assert float((sd.grad.cpu().double() - gref).abs().max()) <= 1e-6 * float(gref.abs().max())
    for _ in range(3):
        again = ops.scatter_mean(src.to(DEV), idx.to(DEV), S)
        assert torch.equal(again, out.detach()), "segment mean is not bit-identical run to run"


def test_segment_gather_rows_and_invalid_ids():
    g = torch.Generator().manual_seed(3)
    for N, S, C in ((1, 1, 1), (777, 31, 100), (5000, 200, 96), (3000, 64, 256), (1000, 50, 7), (300, 20, 1028)):
        table = torch.randn(S, C, generator=g)
        idx = _ids("invalid", N, S, g)
        cnt = torch.randint(0, 5, (S,), generator=g).float()
        ok = (idx >= 0) & (idx < S)
        want = torch.zeros(N, C)
        want[ok] = table[idx[ok]]
        got = ops.segment_gather(table.to(DEV), idx.to(DEV))
        assert torch.equal(got.cpu(), want)
        want2 = torch.zeros(N, C)
        want2[ok] = table[idx[ok]] * (1.0 / cnt.clamp(min=1))[idx[ok]][:, None]
        got2 = ops.segment_gather(table.to(DEV), idx.to(DEV), cnt.to(DEV))
        assert torch.equal(got2.cpu(), want2)


@pytest.mark.parametrize("C", [96, 256])
def test_multiscale_pool_batched_levels_share_one_plan(C):
    """One plan of the batch's ids serves every level forward AND (regrouped by the coarse parent) backward; the gradient
    w.r.t. a coarse level is deterministic and equals the oracle's (materialised up-sampling + scatter_mean)."""
    g = torch.Generator().manual_seed(C)
    N, S = 60_000, 900
    idx = _ids("skewed", N, S, g)
    plan = ops.SegmentPlan(idx.to(DEV), S)
    for Nc in (N // 8, N // 64, 50):
        parent = torch.randint(0, Nc, (N,), generator=g)
        parent[torch.rand(N, generator=g) < 0.01] = -1            # the coarse level lacks that voxel: skipped, not counted
        feat = torch.randn(Nc, C, generator=g)
        fd = feat.to(DEV).requires_grad_(True)
        out = ops.upsample_scatter_mean(fd, parent.to(DEV), idx.to(DEV), S, plan=plan)
        ok = parent >= 0
        fr = feat.clone().requires_grad_(True)
        outr = O.multiscale_segment_pool(fr, parent[ok], idx[ok], S)
        assert float((out.detach().cpu() - outr.detach()).abs().max()) <= 1e-5
        gy = torch.randn(S, C, generator=g)
        out.backward(gy.to(DEV)); outr.backward(gy)
        assert float((fd.grad.cpu() - fr.grad).abs().max()) <= 2e-5 * max(1.0, float(fr.grad.abs().max()))
        g1 = fd.grad.clone()
        fd.grad = None
        ops.upsample_scatter_mean(fd, parent.to(DEV), idx.to(DEV), S, plan=plan).backward(gy.to(DEV))
        assert torch.equal(fd.grad, g1), "coarse-level gradient is not bit-identical run to run"


def test_segment_plan_argument_errors_and_empty_inputs():
    lib = L.lib()
    assert lib.pq3d_segment_plan_bytes(-1, 4) == -1 and lib.pq3d_segment_ws_bytes(4, 4, 0) == -1
    buf = torch.empty(64, dtype=torch.uint8, device=DEV)
    idx = torch.zeros(100, dtype=torch.int64, device=DEV)
    assert lib.pq3d_segment_plan(idx.data_ptr(), 100, 10, buf.data_ptr(), 64, None) == -1
    assert b"plan buffer" in lib.pq3d_last_error()
    # no voxels: every segment row is zero, the gradient is empty
    out = ops.scatter_mean(torch.empty(0, 8, device=DEV), torch.empty(0, dtype=torch.int64, device=DEV), 5)
    assert out.shape == (5, 8) and float(out.abs().sum()) == 0.0
    # no segments
    assert ops.scatter_mean(torch.ones(3, 4, device=DEV), torch.zeros(3, dtype=torch.int64, device=DEV), 0).shape == (0, 4)
