"""The generation head's T5 decoder body on the HIP kernels.

Third-party arithmetic: the reference drives HF ``T5ForConditionalGeneration`` (transformers, requirements.txt:63) with
the encoder bypassed (modules/heads/generation_head.py:20-30).  This restates the *decoder* forward of that model from
its published architecture (T5 v1.0: pre-RMSNorm residual blocks, bias-free projections, un-scaled dot-product attention
with a learned relative-position bias shared by all layers, ReLU feed-forward, tied embedding / LM head scaled by
d_model^-0.5) on this package's ops, reading the parameters of the HF module in place -- state_dict keys and
checkpoints stay HF's.  Teacher-forced training path only; greedy generation keeps using HF ``generate``.
Pinned by fixture F8 (reference head class + the installed transformers) in tests/test_gpu_t5_head.py."""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import ops

_BUCKET_CACHE: Dict[Tuple, torch.Tensor] = {}
DROP_BASE_T5 = 6 << 20


def relative_buckets(T: int, num_buckets: int, max_distance: int, device) -> torch.Tensor:
    """Causal (uni-directional) relative-position bucket of every (query, key) pair, [T, T] int64: exact buckets for
    distances < num_buckets/2, logarithmic up to max_distance (T5 paper, section 2.1; Mesh-TensorFlow definition)."""
    key = (T, num_buckets, max_distance, str(device))
    if key not in _BUCKET_CACHE:
        q = np.arange(T)[:, None]
        k = np.arange(T)[None, :]
        n = np.maximum(q - k, 0)                                   # distance back in time; future keys -> 0 (masked anyway)
        max_exact = num_buckets // 2
        nn_ = np.maximum(n, 1).astype(np.float32)                   # n = 0 always takes the exact branch below
        large = max_exact + (np.log(nn_ / max_exact) / math.log(max_distance / max_exact)
                             * (num_buckets - max_exact)).astype(np.int64)
        large = np.minimum(large, num_buckets - 1)
        _BUCKET_CACHE[key] = torch.from_numpy(np.where(n < max_exact, n, large).astype(np.int64)).to(device)
    return _BUCKET_CACHE[key]


def shift_right(labels: torch.Tensor, start_id: int, pad_id: int) -> torch.Tensor:
    """decoder_input_ids = [start, labels[:-1]] with ignored (-100) labels replaced by the pad token."""
    ids = torch.cat([torch.full_like(labels[:, :1], start_id), labels[:, :-1]], 1)
    return torch.where(ids == -100, torch.full_like(ids, pad_id), ids)


def decoder_logits(model, enc: torch.Tensor, enc_valid: Optional[torch.Tensor], labels: torch.Tensor, ct: int,
                   training: bool = False, drop_epoch_owner=None) -> torch.Tensor:
    """Teacher-forced logits [B, T, vocab] of the HF T5 model's decoder attending to ``enc`` [B, N, d_model]
    (``enc_valid`` [B, N] bool, True = attend)."""
    cfg = model.config
    assert cfg.feed_forward_proj == "relu" and not getattr(cfg, "is_gated_act", False), "T5 v1.0 (ReLU) feed-forward only"
    dec = model.decoder
    H, dkv, dm = cfg.num_heads, cfg.d_kv, cfg.d_model
    dev = enc.device
    B, T = labels.shape
    p_drop = float(cfg.dropout_rate) if training else 0.0
    site = [0]

    def next_drop():                   # nn.Dropout sites of the body, numbered in execution order
        if p_drop <= 0.0:
            return None
        site[0] += 1
        return ops.make_drop(p_drop, DROP_BASE_T5 + site[0], dev)

    def drop(x):
        d = next_drop()
        return x if d is None else ops.dropout(x, d)

    def proj_residual(o, w):
        """x + dropout(o @ w^T): the residual add rides the GEMM epilogue when dropout is off."""
        if p_drop <= 0.0:
            return ops.linear(o, w, None, ct=ct, residual=x)
        return x + drop(ops.linear(o, w, None, ct=ct))

    ids = shift_right(labels, cfg.decoder_start_token_id, cfg.pad_token_id)
    x = drop(ops.embedding(model.shared.weight, ids))
    # relative-position + causal bias, shared by all layers (the table lives in block 0)
    rel = dec.block[0].layer[0].SelfAttention.relative_attention_bias.weight           # [num_buckets, H]
    buckets = relative_buckets(T, cfg.relative_attention_num_buckets, getattr(cfg, "relative_attention_max_distance", 128), dev)
    pos_bias = ops.embedding(rel, buckets).permute(2, 0, 1)                              # [H, T, T]
    causal = torch.ones(T, T, dtype=torch.bool, device=dev).triu(1)
    self_bias = pos_bias.masked_fill(causal, float("-inf")).unsqueeze(0).expand(B, H, T, T).contiguous()
    enc_kpm = enc_valid.logical_not().contiguous() if enc_valid is not None else None
    ad = ops.act_dtype(ct)
    for blk in dec.block:
        sa, ca, ff = blk.layer[0], blk.layer[1], blk.layer[2]
        # -- self attention
        h = ops.rmsnorm(x, sa.layer_norm.weight, cfg.layer_norm_epsilon)
        A = sa.SelfAttention
        q, k, v = (ops.linear(h, w_.weight, None, ct=ct, out_dtype=ad) for w_ in (A.q, A.k, A.v))
        o = ops.attention(q, k, v, H=H, ct=ct, scale=1.0, bias=self_bias, drop=next_drop())
        x = proj_residual(o, A.o.weight)
        # -- cross attention to the projected query tokens (no position bias)
        h = ops.rmsnorm(x, ca.layer_norm.weight, cfg.layer_norm_epsilon)
        A = ca.EncDecAttention
        q = ops.linear(h, A.q.weight, None, ct=ct, out_dtype=ad)
        k, v = (ops.linear(enc, w_.weight, None, ct=ct, out_dtype=ad) for w_ in (A.k, A.v))
        o = ops.attention(q, k, v, H=H, ct=ct, scale=1.0, kpm=enc_kpm, drop=next_drop())
        x = proj_residual(o, A.o.weight)
        # -- feed forward
        h = ops.rmsnorm(x, ff.layer_norm.weight, cfg.layer_norm_epsilon)
        hid = drop(ops.linear(h, ff.DenseReluDense.wi.weight, None, ct=ct, act="relu", out_dtype=ad))
        x = proj_residual(hid, ff.DenseReluDense.wo.weight)
    x = drop(ops.rmsnorm(x, dec.final_layer_norm.weight, cfg.layer_norm_epsilon))
    if cfg.tie_word_embeddings:
        x = x * (dm ** -0.5)
    return ops.linear(x, model.lm_head.weight, None, ct=ct)
