"""The generation head's T5 decoder body on the HIP kernels.

Third-party arithmetic: the reference drives HF ``T5ForConditionalGeneration`` (transformers, requirements.txt:63) with
the encoder bypassed (modules/heads/generation_head.py:20-30).  This restates the *decoder* forward of that model from
its published architecture (T5 v1.0: pre-RMSNorm residual blocks, bias-free projections, un-scaled dot-product attention
with a learned relative-position bias shared by all layers, ReLU feed-forward, tied embedding / LM head scaled by
d_model^-0.5) on this package's ops, reading the parameters of the HF module in place -- state_dict keys and
checkpoints stay HF's.  ``decoder_logits`` is the teacher-forced training path; ``GreedyDecoder`` is the eval path
(greedy ``generate`` with a KV cache: one fixed-shape decode step captured in a HIP graph and replayed per token).
Pinned by fixture F8 (reference head class + the installed transformers) in tests/test_gpu_t5_head.py."""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import os

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

    handover = [None]   # (drop site, slot) of the last proj_residual: the norm that consumes its output writes the masked gradient

    def proj_residual(o, w):
        """x + dropout(o @ w^T): dropout and the residual add ride the GEMM epilogue (one launch); in the backward the
        dropout's mask is applied by the NEXT norm's backward kernel (ops.rmsnorm grad_drop) instead of a launch of its own."""
        d = next_drop()
        slot = {} if d is not None else None
        handover[0] = (d, slot) if d is not None else None
        return ops.linear(o, w, None, ct=ct, residual=x, drop=d, masked_grad=slot)

    def norm_res(x_, w_):
        gd, handover[0] = handover[0], None
        return ops.rmsnorm_res(x_, w_, cfg.layer_norm_epsilon, grad_drop=gd)

    T5_TT_MAX = 114 * 114   # pq3d_t5_bias_bwd keeps the [T, T] sums in LDS
    rel = dec.block[0].layer[0].SelfAttention.relative_attention_bias.weight           # [num_buckets, H]
    buckets = relative_buckets(T, cfg.relative_attention_num_buckets, getattr(cfg, "relative_attention_max_distance", 128), dev)
    if T * T <= T5_TT_MAX and rel.shape[0] <= 256 and (enc_valid is None or enc_valid.dtype == torch.bool):
        # shift_right, relative-position + causal bias (shared by all layers; the table lives in block 0) and the encoder's
        # key-padding bytes: one launch (was ~12 framework launches forward, ~8 backward)
        ids, self_bias, enc_kpm = ops.t5_prep(rel, labels, buckets, enc_valid, cfg.decoder_start_token_id, cfg.pad_token_id, H)
    else:
        ids = shift_right(labels, cfg.decoder_start_token_id, cfg.pad_token_id)
        pos_bias = ops.embedding(rel, buckets).permute(2, 0, 1)                              # [H, T, T]
        causal = torch.ones(T, T, dtype=torch.bool, device=dev).triu(1)
        self_bias = pos_bias.masked_fill(causal, float("-inf")).unsqueeze(0).expand(B, H, T, T).contiguous()
        enc_kpm = enc_valid.logical_not().contiguous() if enc_valid is not None else None
    x = ops.embedding(model.shared.weight, ids, drop=next_drop())
    ad = ops.act_dtype(ct)
    # cross-attention K/V projections of ALL layers read the same encoder tokens: one grouped launch (and one
    # K-concatenated input-gradient product, one grouped weight-gradient product in the backward)
    xkv = ops.linear_group([enc] * (2 * len(dec.block)),
                           [w_.weight for blk in dec.block for w_ in (blk.layer[1].EncDecAttention.k,
                                                                       blk.layer[1].EncDecAttention.v)], ct=ct, out_dtype=ad)
    self_biases = ops.fanout(self_bias, len(dec.block))   # every block reads the shared position bias: one gradient sum
    for li, blk in enumerate(dec.block):
        sa, ca, ff = blk.layer[0], blk.layer[1], blk.layer[2]
        # -- self attention
        h, x = norm_res(x, sa.layer_norm.weight)
        A = sa.SelfAttention
        q, k, v = ops.linear_group([h, h, h], [A.q.weight, A.k.weight, A.v.weight], ct=ct, out_dtype=ad)
        o = ops.attention(q, k, v, H=H, ct=ct, scale=1.0, bias=self_biases[li], drop=next_drop())
        x = proj_residual(o, A.o.weight)
        # -- cross attention to the projected query tokens (no position bias)
        h, x = norm_res(x, ca.layer_norm.weight)
        A = ca.EncDecAttention
        q = ops.linear(h, A.q.weight, None, ct=ct, out_dtype=ad)
        k, v = xkv[2 * li], xkv[2 * li + 1]
        o = ops.attention(q, k, v, H=H, ct=ct, scale=1.0, kpm=enc_kpm, drop=next_drop())
        x = proj_residual(o, A.o.weight)
        # -- feed forward
        h, x = norm_res(x, ff.layer_norm.weight)
        hid = ops.linear(h, ff.DenseReluDense.wi.weight, None, ct=ct, act="relu", out_dtype=ad, drop=next_drop())
        x = proj_residual(hid, ff.DenseReluDense.wo.weight)
    gd, handover[0] = handover[0], None
    x = ops.rmsnorm(x, dec.final_layer_norm.weight, cfg.layer_norm_epsilon, grad_drop=gd)
    # final dropout and the tied-embedding scale d_model^-0.5 in one launch (either direction)
    x = ops.dropout(x, next_drop(), alpha=(dm ** -0.5) if cfg.tie_word_embeddings else 1.0)
    return ops.linear(x, model.lm_head.weight, None, ct=ct)


# ------------------------------------------------------------------------------------------------ greedy generation
def new_token_budget(model, max_new_tokens: Optional[int] = None) -> int:
    """How many tokens greedy generation may append after the start token.  Explicit argument first; then the HF
    generation config; then ``config.max_new_tokens`` / ``config.max_length`` -- the reference passes
    ``max_new_tokens: 50`` through ``model.config.update(kwargs)`` (generation_head.py:12,
    configs/unified_tasks_sceneverse.yaml:180), which the transformers version it pins folds into generation; else
    HF's default of 20."""
    if max_new_tokens is not None:
        return int(max_new_tokens)
    gen, cfg = model.generation_config, model.config
    for src in (gen, cfg):
        v = getattr(src, "max_new_tokens", None)
        if v is not None:
            return int(v)
    for src in (gen, cfg):
        v = getattr(src, "max_length", None)
        if v is not None and int(v) != 20:      # 20 is HF's legacy default, not a user choice
            return int(v) - 1
    return 20


class GreedyDecoder:
    """Greedy decoding (HF ``generate(do_sample=False)`` as generation_head.py:28 calls it) of the T5 decoder on the
    HIP kernels, for a fixed (batch, encoder length, token budget).

    Device-resident state, fixed shapes: per-layer self-attention K/V caches [B, T, inner] (T = token budget), the
    cross-attention K/V of the encoder tokens (computed once per call), the current token, the step counter and the
    per-sequence 'unfinished' flag.  One decode step -- embed the current token, 3 sublayers x L layers against the
    caches (the step's row of the causal relative-position bias is gathered by the device-side step counter, so unwritten
    cache slots carry -inf), LM head, argmax, EOS/pad bookkeeping, counter += 1 -- has identical launches for every
    step, so it is captured ONCE in a HIP graph and replayed T times with no host round trip; the only sync is the
    final read-back, where the output is trimmed to the step at which every sequence had emitted EOS (what HF's
    stopping criterion returns)."""

    def __init__(self, model, B: int, N: int, ct: int, max_new_tokens: int, device, use_graph: bool = True):
        cfg = model.config
        assert cfg.feed_forward_proj == "relu" and not getattr(cfg, "is_gated_act", False), "T5 v1.0 (ReLU) feed-forward only"
        self.model, self.ct, self.B, self.N, self.T = model, ct, B, N, int(max_new_tokens)
        self.dev = torch.device(device)
        gen = model.generation_config
        self.eos = gen.eos_token_id if gen.eos_token_id is not None else cfg.eos_token_id
        self.pad = gen.pad_token_id if gen.pad_token_id is not None else cfg.pad_token_id
        self.start = cfg.decoder_start_token_id
        if isinstance(self.eos, (list, tuple)):
            self.eos = self.eos[0]
        assert self.start is not None and (self.eos is None or self.pad is not None)
        H, inner, dm, T = cfg.num_heads, cfg.num_heads * cfg.d_kv, cfg.d_model, self.T
        ad, dev, nl = ops.act_dtype(ct), self.dev, len(model.decoder.block)
        self.enc = torch.zeros(B, N, dm, dtype=torch.float32, device=dev)
        self.enc_kpm = torch.zeros(B, N, dtype=torch.bool, device=dev)
        self.kc = [torch.zeros(B, T, inner, dtype=ad, device=dev) for _ in range(nl)]
        self.vc = [torch.zeros(B, T, inner, dtype=ad, device=dev) for _ in range(nl)]
        self.xk = [torch.zeros(B, N, inner, dtype=ad, device=dev) for _ in range(nl)]
        self.xv = [torch.zeros(B, N, inner, dtype=ad, device=dev) for _ in range(nl)]
        self.tok = torch.zeros(B, dtype=torch.long, device=dev)
        self.t = torch.zeros(1, dtype=torch.long, device=dev)
        self.unfinished = torch.ones(B, dtype=torch.bool, device=dev)
        self.out = torch.zeros(B, T, dtype=torch.long, device=dev)
        self.buckets = relative_buckets(T, cfg.relative_attention_num_buckets,
                                        getattr(cfg, "relative_attention_max_distance", 128), dev)
        self.causal = torch.ones(T, T, dtype=torch.bool, device=dev).triu(1)
        self.bias_all = torch.zeros(T, H, T, dtype=torch.float32, device=dev)
        self.use_graph = bool(use_graph) and self.dev.type == "cuda"
        self._g_prefill = self._g_step = None

    # -- the three captured pieces ------------------------------------------------------------------------------
    def _prefill(self):
        """Per call: cross-attention K/V of the encoder tokens, the [T, H, T] causal relative-position bias (read from
        the live table, so a fine-tuned table is picked up), and the state reset."""
        m, ct, ad = self.model, self.ct, ops.act_dtype(self.ct)
        for l, blk in enumerate(m.decoder.block):
            A = blk.layer[1].EncDecAttention
            self.xk[l].copy_(ops.linear(self.enc, A.k.weight, None, ct=ct, out_dtype=ad))
            self.xv[l].copy_(ops.linear(self.enc, A.v.weight, None, ct=ct, out_dtype=ad))
        rel = m.decoder.block[0].layer[0].SelfAttention.relative_attention_bias.weight
        pos = ops.embedding(rel, self.buckets)                                   # [T, T, H]
        self.bias_all.copy_(pos.masked_fill(self.causal.unsqueeze(-1), float("-inf")).permute(0, 2, 1))
        for l in range(len(self.kc)):
            self.kc[l].zero_(); self.vc[l].zero_()
        self.tok.fill_(self.start); self.t.zero_(); self.unfinished.fill_(True); self.out.fill_(self.pad if self.pad is not None else 0)

    def _step(self):
        m, ct, ad = self.model, self.ct, ops.act_dtype(self.ct)
        cfg = m.config
        B, H, T, eps = self.B, cfg.num_heads, self.T, cfg.layer_norm_epsilon
        x = ops.embedding(m.shared.weight, self.tok.view(B, 1))                  # [B, 1, d_model]
        bias = self.bias_all.index_select(0, self.t).view(1, H, 1, T).expand(B, H, 1, T).contiguous()
        for l, blk in enumerate(m.decoder.block):
            sa, ca, ff = blk.layer[0], blk.layer[1], blk.layer[2]
            A = sa.SelfAttention
            h = ops.rmsnorm(x, sa.layer_norm.weight, eps)
            q, k, v = (ops.linear(h, w_.weight, None, ct=ct, out_dtype=ad) for w_ in (A.q, A.k, A.v))
            self.kc[l].index_copy_(1, self.t, k)
            self.vc[l].index_copy_(1, self.t, v)
            o = ops.attention(q, self.kc[l], self.vc[l], H=H, ct=ct, scale=1.0, bias=bias)
            x = ops.linear(o, A.o.weight, None, ct=ct, residual=x)
            A = ca.EncDecAttention
            h = ops.rmsnorm(x, ca.layer_norm.weight, eps)
            q = ops.linear(h, A.q.weight, None, ct=ct, out_dtype=ad)
            o = ops.attention(q, self.xk[l], self.xv[l], H=H, ct=ct, scale=1.0, kpm=self.enc_kpm)
            x = ops.linear(o, A.o.weight, None, ct=ct, residual=x)
            h = ops.rmsnorm(x, ff.layer_norm.weight, eps)
            hid = ops.linear(h, ff.DenseReluDense.wi.weight, None, ct=ct, act="relu", out_dtype=ad)
            x = ops.linear(hid, ff.DenseReluDense.wo.weight, None, ct=ct, residual=x)
        x = ops.rmsnorm(x, m.decoder.final_layer_norm.weight, eps)
        if cfg.tie_word_embeddings:
            x = x * (cfg.d_model ** -0.5)
        logits = ops.linear(x, m.lm_head.weight, None, ct=ct)                    # [B, 1, vocab]
        nxt = logits[:, 0].argmax(-1)
        if self.eos is not None:
            nxt = torch.where(self.unfinished, nxt, torch.full_like(nxt, self.pad))
        self.out.index_copy_(1, self.t, nxt.view(B, 1))
        if self.eos is not None:
            self.unfinished.logical_and_(nxt != self.eos)
        self.tok.copy_(nxt)
        self.t.add_(1)

    def _capture(self):
        s = torch.cuda.Stream(device=self.dev)
        s.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(s):      # warm up on the side stream (allocator pools, lazy module loads), then capture
            self._prefill(); self._step()
            s.synchronize()
            self._g_prefill, self._g_step = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._g_prefill, stream=s):
                self._prefill()
            with torch.cuda.graph(self._g_step, stream=s):
                self._step()
        torch.cuda.current_stream(self.dev).wait_stream(s)

    @torch.no_grad()
    def __call__(self, enc: torch.Tensor, enc_valid: Optional[torch.Tensor]) -> torch.Tensor:
        """enc [B, N, d_model], enc_valid [B, N] bool (True = attend) -> generated ids [B, <= T] WITHOUT the start
        token (generation_head.py:29), padded with pad_token_id after each sequence's EOS."""
        assert tuple(enc.shape) == (self.B, self.N, self.model.config.d_model)
        self.enc.copy_(enc)
        if enc_valid is None:
            self.enc_kpm.zero_()
        else:
            self.enc_kpm.copy_(enc_valid.to(torch.bool).logical_not())
        if self.use_graph:
            if self._g_step is None:
                self._capture()
            self._g_prefill.replay()
            for _ in range(self.T):
                self._g_step.replay()
        else:
            self._prefill()
            for _ in range(self.T):
                self._step()
        out = self.out.clone()
        if self.eos is None:
            return out
        done = ((out == self.eos).cumsum(1) > 0).all(0)                           # [T]: every sequence has emitted EOS
        idx = torch.nonzero(done)
        n = int(idx[0]) + 1 if idx.numel() else self.T
        return out[:, :n]
