"""CPU oracle of the ragged -> padded collate helpers.  TEST INFRASTRUCTURE ONLY.
Restates pad_sequence / pad_sequence_2d (data/data_utils.py:337-382); pinned by fixture F11 produced with the
reference's own functions (tests/golden/make_golden.py:run_collate_case)."""
from __future__ import annotations

from typing import List

import torch


def pad_sequence(sequence_list: List[torch.Tensor], max_len=None, pad=0, return_mask=False):
    lens = [x.shape[0] for x in sequence_list]
    max_len = max(lens) if max_len is None else max_len
    out = torch.full((len(sequence_list), max_len) + tuple(sequence_list[0].shape[1:]), pad, dtype=sequence_list[0].dtype)
    for i, t in enumerate(sequence_list):
        out[i, :t.shape[0]] = t
    if not return_mask:
        return out
    mask = torch.arange(max_len)[None, :] >= torch.tensor(lens)[:, None]      # True as masked
    return out, mask


def pad_sequence_2d(sequence_list: List[torch.Tensor], max_height=None, max_width=None, pad=0, return_mask=False):
    H = max(x.shape[0] for x in sequence_list) if max_height is None else max_height
    W = max(x.shape[1] for x in sequence_list) if max_width is None else max_width
    out = torch.full((len(sequence_list), H, W) + tuple(sequence_list[0].shape[2:]), pad, dtype=sequence_list[0].dtype)
    mask = torch.ones(len(sequence_list), H, W, dtype=torch.bool)
    for i, t in enumerate(sequence_list):
        out[i, :t.shape[0], :t.shape[1]] = t
        mask[i, :t.shape[0], :t.shape[1]] = False
    return (out, mask) if return_mask else out
