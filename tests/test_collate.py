"""pad_sequence / pad_sequence_2d: CPU oracle vs fixture F11 (reference functions), HIP kernels vs the same fixture."""
import numpy as np
import pytest
import torch

from oracle import collate_oracle as CO
from pq3d_amd import synth
from tests import util

CASES = [("feats", lambda f, c, l, v, s, P, P2: P(f)), ("labels", lambda f, c, l, v, s, P, P2: P(l, pad=-100)),
         ("valid", lambda f, c, l, v, s, P, P2: P(v)), ("feats_len80", lambda f, c, l, v, s, P, P2: P(f, max_len=80, pad=1.5)),
         ("seg_masks_f", lambda f, c, l, v, s, P, P2: P2([x.float() for x in s], max_height=12, max_width=70, pad=-1))]


def check(z, P, P2, to=lambda t: t):
    f, c, l, v, s = ([to(t) for t in ts] for ts in synth.collate_inputs())
    for key, fn in CASES:
        got = fn(f, c, l, v, s, P, P2).cpu().numpy()
        assert got.dtype == z[key].dtype and np.array_equal(got, z[key]), key
    cc, m = P(c, return_mask=True)
    assert np.array_equal(cc.cpu().numpy(), z["centers"]) and np.array_equal(m.cpu().numpy(), z["centers_mask"])
    sm, pm = P2(s, return_mask=True)
    assert np.array_equal(sm.cpu().numpy(), z["seg_masks"]) and np.array_equal(pm.cpu().numpy(), z["seg_masks_mask"])


def test_collate_oracle_matches_reference():
    z, _ = util.load_fixture("F11_collate")
    check(z, CO.pad_sequence, CO.pad_sequence_2d)


@pytest.mark.gpu
def test_collate_kernels_match_reference():
    from pq3d_amd import collate as HC
    z, _ = util.load_fixture("F11_collate")
    check(z, HC.pad_sequence, HC.pad_sequence_2d, to=lambda t: t.cuda())


@pytest.mark.gpu
def test_ragged_batch_pads_segment_features_at_decoder_sizes():
    from pq3d_amd.collate import RaggedBatch
    r = np.random.default_rng(0)
    lens = [4096, 2500, 3333, 1, 4000, 2048, 777, 4095]
    vals = torch.from_numpy(r.standard_normal((sum(lens), 256)).astype(np.float32)).cuda()
    out, mask = RaggedBatch(vals, lens).pad(return_mask=True)
    off = 0
    for b, n in enumerate(lens):
        assert torch.equal(out[b, :n], vals[off:off + n]) and not out[b, n:].any() and not mask[b, :n].any() and mask[b, n:].all()
        off += n
