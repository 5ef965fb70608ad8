"""Ragged -> padded collate on the device: ``pad_sequence`` / ``pad_sequence_2d`` with the reference's
signatures (data/data_utils.py:337-382) plus a packed ``RaggedBatch`` so that a whole key of the batch is padded by ONE
launch instead of one slice-assignment per sample (InstSegDatasetWrapper.collate_fn, instseg_wrapper.py:27-81, runs
these per key on CPU tensors; on the device the per-sample loop is B tiny copies each)."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib as L


def _pad_bytes(dtype: torch.dtype, pad) -> C.Array:
    t = torch.tensor([pad], dtype=dtype) if dtype != torch.bool else torch.tensor([bool(pad)])
    raw = t.view(torch.uint8).numpy().tobytes() if dtype != torch.bool else bytes([int(bool(pad))])
    return (C.c_uint8 * len(raw))(*raw)


class RaggedBatch:
    """values [total, *tail] with sample b = rows offsets[b]:offsets[b+1] (offsets int64 on the device and as a list)."""

    def __init__(self, values: torch.Tensor, lengths: Sequence[int]):
        self.values = values.contiguous()
        self.lengths = [int(x) for x in lengths]
        off = np.zeros(len(self.lengths) + 1, dtype=np.int64)
        off[1:] = np.cumsum(self.lengths)
        assert int(off[-1]) == self.values.shape[0]
        self.offsets = torch.from_numpy(off).to(values.device)

    @classmethod
    def from_list(cls, seqs: Sequence[torch.Tensor]) -> "RaggedBatch":
        return cls(torch.cat(list(seqs), 0), [s.shape[0] for s in seqs])

    def pad(self, max_len: Optional[int] = None, pad=0, return_mask: bool = False):
        v = self.values
        B, Lmax = len(self.lengths), max_len if max_len is not None else max(self.lengths)
        tail = tuple(v.shape[1:])
        D = int(np.prod(tail)) if tail else 1
        out = torch.empty(B, Lmax, *tail, dtype=v.dtype, device=v.device)
        mask = torch.empty(B, Lmax, dtype=torch.bool, device=v.device) if return_mask else None
        pb = _pad_bytes(v.dtype, pad)
        L.check(L.lib().pq3d_pad_sequence(L.ptr(v) if v.numel() else None, L.ptr(self.offsets), L.ptr(out), L.ptr(mask), B,
                                          Lmax, D, v.element_size(), C.cast(pb, C.c_void_p), L.stream()), "pq3d_pad_sequence")
        return (out, mask) if return_mask else out


def pad_sequence(sequence_list: List[torch.Tensor], max_len=None, pad=0, return_mask=False):
    """data/data_utils.py:337-356."""
    return RaggedBatch.from_list(sequence_list).pad(max_len, pad, return_mask)


def pad_sequence_2d(sequence_list: List[torch.Tensor], max_height=None, max_width=None, pad=0, return_mask=False):
    """data/data_utils.py:358-382 (per-sample [h_b, w_b, *tail] -> [B, H, W, *tail], mask True where padded)."""
    dev, dt = sequence_list[0].device, sequence_list[0].dtype
    hs = [int(x.shape[0]) for x in sequence_list]
    ws = [int(x.shape[1]) for x in sequence_list]
    H = max_height if max_height is not None else max(hs)
    W = max_width if max_width is not None else max(ws)
    tail = tuple(sequence_list[0].shape[2:])
    D = int(np.prod(tail)) if tail else 1
    flat = torch.cat([x.contiguous().reshape(-1) for x in sequence_list], 0)
    sizes = np.array([h * w * D for h, w in zip(hs, ws)], dtype=np.int64)
    off = np.zeros(len(hs), dtype=np.int64)
    off[1:] = np.cumsum(sizes)[:-1]
    meta = torch.from_numpy(np.concatenate([off, np.array(hs, dtype=np.int64), np.array(ws, dtype=np.int64)])).to(dev)
    B = len(hs)
    offsets, hd, wd = meta[:B].contiguous(), meta[B:2 * B].to(torch.int32), meta[2 * B:].to(torch.int32)
    out = torch.empty(B, H, W, *tail, dtype=dt, device=dev)
    mask = torch.empty(B, H, W, dtype=torch.bool, device=dev) if return_mask else None
    pb = _pad_bytes(dt, pad)
    L.check(L.lib().pq3d_pad_sequence_2d(L.ptr(flat) if flat.numel() else None, L.ptr(offsets), L.ptr(hd), L.ptr(wd),
                                         L.ptr(out), L.ptr(mask), B, H, W, D, flat.element_size(),
                                         C.cast(pb, C.c_void_p), L.stream()), "pq3d_pad_sequence_2d")
    return (out, mask) if return_mask else out
