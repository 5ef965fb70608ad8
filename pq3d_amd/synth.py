"""Deterministic synthetic weights and inputs (SURVEY §8c fixture plan, §8d synthetic inputs).

Used by the golden-fixture generator, the tests and ``bench.py``.  Everything is drawn from
``numpy.random.default_rng`` keyed by (seed, crc32(name)) so a tensor's values depend only on
its *name and shape*, never on module construction order: the same call fills the reference
modules (golden generation), the oracle's flat dict and the HIP modules.
"""
from __future__ import annotations

import zlib
from typing import Dict, Mapping, Sequence

import numpy as np
import torch


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(name.encode())])


def synth_tensor(name: str, shape: Sequence[int], seed: int = 0) -> torch.Tensor:
    """matrices ~ N(0,0.05) (so attention is not degenerate), biases ~ N(0,0.02),
    LayerNorm / BatchNorm gamma ~ U(0.5,1.5), Fourier ``gauss_B`` ~ N(0,1), BatchNorm running_var ~ U(0.5,1.5),
    running_mean ~ N(0,0.1)."""
    r = _rng(seed, name)
    shape = tuple(int(s) for s in shape)
    if name.endswith("num_batches_tracked"):
        return torch.zeros(shape, dtype=torch.int64)
    if name.endswith("gauss_B"):
        a = r.standard_normal(shape)
    elif name.endswith("running_var"):
        a = r.uniform(0.5, 1.5, shape)
    elif name.endswith("running_mean"):
        a = 0.1 * r.standard_normal(shape)
    elif len(shape) >= 2:
        a = 0.05 * r.standard_normal(shape)
    elif name.endswith("weight"):
        a = r.uniform(0.5, 1.5, shape)
    else:
        a = 0.02 * r.standard_normal(shape)
    return torch.from_numpy(a.astype(np.float32))


def synth_state_dict(shapes: Mapping[str, Sequence[int]], seed: int = 0) -> Dict[str, torch.Tensor]:
    return {k: synth_tensor(k, tuple(v), seed) for k, v in shapes.items()}


def fill_module(module: torch.nn.Module, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Overwrite every parameter/buffer of ``module`` with its synthetic value; returns the dict."""
    sd = module.state_dict()
    new = synth_state_dict({k: v.shape for k, v in sd.items()}, seed)
    module.load_state_dict(new, strict=True)
    return new


def state_checksum(sd: Mapping[str, torch.Tensor]) -> float:
    """Order-independent fingerprint stored in fixtures to detect RNG drift."""
    return float(sum(float(v.double().abs().sum()) for v in sd.values()))


def synth_data_dict(B: int, n_seg: int, n_q: int, d_in: Mapping[str, int], seed: int = 1234,
                    memories: Sequence[str] = ("voxel", "mv", "pc"), prompt_len: int = 0, d_model: int = 0,
                    full_valid: bool = False, query_valid_min: int | None = None,
                    loc_dim: int = 3) -> Dict[str, torch.Tensor]:
    """SURVEY §8d synthetic inputs.  Pad masks here are the data_dict convention: True = valid."""
    r = np.random.default_rng(seed)
    valid_len = r.integers(n_seg // 2, n_seg + 1, size=B)
    valid_len[0] = n_seg
    if full_valid:
        valid_len[:] = n_seg
    seg_valid = np.arange(n_seg)[None, :] < valid_len[:, None]
    dd: Dict[str, torch.Tensor] = {}
    for m in memories:
        if m == "prompt":
            continue
        f = r.standard_normal((B, n_seg, d_in[m])).astype(np.float32)
        f[~seg_valid] = 0.0
        dd[f"{m}_seg_fts"] = torch.from_numpy(f)
        dd[f"{m}_seg_pad_masks"] = torch.from_numpy(seg_valid.copy())
    dd["seg_pad_masks"] = torch.from_numpy(seg_valid.copy())
    dd["seg_center"] = torch.from_numpy(r.uniform(0, 4, (B, n_seg, loc_dim)).astype(np.float32))
    dd["query_locs"] = torch.from_numpy(r.uniform(0, 4, (B, n_q, loc_dim)).astype(np.float32))
    dd["coord_min"] = torch.zeros(B, 3)
    dd["coord_max"] = torch.full((B, 3), 4.0)
    if query_valid_min is None:
        qv = np.ones((B, n_q), dtype=bool)
    else:
        ql = r.integers(query_valid_min, n_q + 1, size=B)
        qv = np.arange(n_q)[None, :] < ql[:, None]
    dd["query_pad_masks"] = torch.from_numpy(qv)
    if prompt_len:
        pl = r.integers(max(1, prompt_len // 4), prompt_len + 1, size=B)
        pv = np.arange(prompt_len)[None, :] < pl[:, None]
        pf = r.standard_normal((B, prompt_len, d_model)).astype(np.float32)
        pf[~pv] = 0.0
        dd["prompt_feat"] = torch.from_numpy(pf)
        dd["prompt_pad_masks"] = torch.from_numpy(pv)
    return dd


def prompt_loc_inputs(B: int, T: int, seed: int = 77) -> Dict[str, torch.Tensor]:
    """Location prompts of query3d_unified.py:80-108 (PromptType.LOC = 3): prompt [B, T] whose first entries are a point in the
    scene's coordinate range, ragged pad masks (True = valid), one prompt type per scene.  Shared by tests/golden/make_golden.py
    (fixture F20) and the tests."""
    r = np.random.default_rng(seed)
    pl = r.integers(1, T + 1, size=B)
    return {"prompt": torch.from_numpy(r.uniform(0, 4, (B, T)).astype(np.float32)),
            "prompt_pad_masks": torch.from_numpy(np.arange(T)[None, :] < pl[:, None]),
            "prompt_type": torch.full((B,), 3, dtype=torch.long)}


def variant_inputs(B=2, Ns=50, Nq=12, d=64, seed=31):
    """Inputs of fixture F21 (pre-norm layers, 'bias' spatial fusion, GroundHeadV1), shared by make_golden.py and the tests:
    a query state, two scene memories with ragged padding, position embeddings, query centres."""
    r = np.random.default_rng(seed)
    t = lambda *sh: torch.from_numpy(r.standard_normal(sh).astype(np.float32))
    vl = r.integers(Ns // 2, Ns + 1, size=B); vl[0] = Ns
    pad = torch.from_numpy(np.arange(Ns)[None, :] >= vl[:, None])          # True = padded
    ql = r.integers(Nq // 2, Nq + 1, size=B); ql[0] = Nq
    qpad = torch.from_numpy(np.arange(Nq)[None, :] >= ql[:, None])
    feats = {}
    for m in ("voxel", "mv"):
        f = t(B, Ns, d); f[pad] = 0.0
        feats[m] = f
    return dict(query=t(B, Nq, d), qpos=t(B, Nq, d), fpos=t(B, Ns, d), feats=feats, pad=pad, qpad=qpad,
                centers=torch.from_numpy(r.uniform(0, 4, (B, Nq, 3)).astype(np.float32)),
                txt=t(B, 5, d), pre=t(B, Nq, d))


def criterion_inputs(seed=21, B=3, Ns=70, Nq=12, C=21, n_layers=3, seg_len=(70, 55, 61), n_inst=(5, 9, 3)):
    """Synthetic predictions / targets of the F9 criterion fixture (also rebuilt by the tests)."""
    r = np.random.default_rng(seed)
    masks, logits = [], []
    for _ in range(n_layers):
        m = (r.standard_normal((B, Ns, Nq)) * 2.0).astype(np.float32)
        for b in range(B):
            m[b, seg_len[b]:] = -1e6          # what the mask head writes for padded segments (mask_head.py:38)
        lg = r.standard_normal((B, Nq, C)).astype(np.float32)
        lg[..., [0, 2]] = -np.inf              # filter_out_classes (mask_head.py:28)
        masks.append(torch.from_numpy(m)); logits.append(torch.from_numpy(lg))
    labels = [torch.from_numpy(r.integers(3, C - 1, n_inst[b])) for b in range(B)]
    labels[1][2] = -100                        # one ignored target
    seg = [torch.from_numpy((r.random((n_inst[b], seg_len[b])) < 0.2).astype(np.int64)) for b in range(B)]
    return masks, logits, labels, seg


def direct_loss_inputs(seed=31, B=3, S=70, N=12, C=21, n_layers=2, seg_len=(70, 55, 61), n_inst=(5, 12, 3)):
    """Synthetic predictions / padded targets of the F10 fixture (DirectCriterion and the stage-2 mask_loss)."""
    r = np.random.default_rng(seed)
    masks, logits = [], []
    for _ in range(n_layers):
        masks.append(torch.from_numpy((r.standard_normal((B, S, N)) * 2.0).astype(np.float32)))
        logits.append(torch.from_numpy(r.standard_normal((B, N, C)).astype(np.float32)))
    tgt = torch.zeros(B, N, S)
    pad = torch.zeros(B, N, S, dtype=torch.bool)          # False for padding pixels and padding instances
    labels = torch.full((B, N), -100, dtype=torch.int64)
    for b in range(B):
        tgt[b, :n_inst[b], :seg_len[b]] = torch.from_numpy((r.random((n_inst[b], seg_len[b])) < 0.25).astype(np.float32))
        pad[b, :n_inst[b], :seg_len[b]] = True
        labels[b, :n_inst[b]] = torch.from_numpy(r.integers(0, C, n_inst[b]))
    obj_masks = labels >= 0
    lab2 = labels.clamp(min=0)
    return masks, logits, tgt, pad, labels, obj_masks, lab2


def collate_inputs(seed=41):
    """Ragged per-sample tensors of the F11 collate fixture: segment features, centres, labels (int64), pad masks (bool),
    instance x segment target masks (int64, 2-D ragged)."""
    r = np.random.default_rng(seed)
    lens = [37, 5, 64, 1]
    feats = [torch.from_numpy(r.standard_normal((n, 24)).astype(np.float32)) for n in lens]
    centers = [torch.from_numpy(r.uniform(0, 4, (n, 3)).astype(np.float32)) for n in lens]
    labels = [torch.from_numpy(r.integers(0, 200, n)) for n in lens]
    valid = [torch.ones(n, dtype=torch.bool) for n in lens]
    ninst = [4, 9, 2, 1]
    seg_masks = [torch.from_numpy((r.random((k, n)) < 0.3).astype(np.int64)) for k, n in zip(ninst, lens)]
    return feats, centers, labels, valid, seg_masks


def pointcloud_inputs(M: int = 4, P: int = 300, C: int = 3, seed: int = 21) -> torch.Tensor:
    """Object point clouds [M, P, 3 + C] for the PointNet++ tokenizer: xyz uniform in a unit cube (so radius-0.2 balls
    hold fewer than nsample points and the fill rule is exercised), one cloud squeezed to a thin slab, colours in [0,1]."""
    r = np.random.default_rng(seed)
    xyz = r.uniform(-0.5, 0.5, (M, P, 3))
    xyz[M - 1, :, 2] *= 0.05
    rgb = r.uniform(0.0, 1.0, (M, P, C))
    return torch.from_numpy(np.concatenate([xyz, rgb], -1).astype(np.float32))
