"""nn.Modules of the promptable query decoder with the reference's class names, constructor signatures,
forward contracts and state_dict keys, computing through the HIP kernels in ``ops``.

nn.Linear / nn.LayerNorm / nn.Sequential instances below are *parameter containers* only (they give the
reference's state_dict key names and let DDP / optimisers see ordinary leaf Parameters); their forward() is
never called.  ``compute`` selects the MFMA path: 'bf16' (bf16 operands, fp32 accumulate / softmax / LayerNorm /
residual stream), 'bf16x3' (the same with a split-bf16, fp32-grade key/value side: see ``CT``) or 'fp32' (exact-f32 MFMA
everywhere).

Dropout: in train mode every nn.Dropout / MultiheadAttention(dropout=p) site of the reference is applied INSIDE the
kernels (attention probabilities, residual branches before add+LayerNorm, FFN hidden) from a counter-based generator
(include/pq3d_hip.h "Dropout", ops.DropRNG): same distribution as torch's, different random stream.  Site ids are
structural -- (module base, layer application, kind, memory) -> ops.drop_site -- so the fused executor, the modular
path and the test-side mask generator agree.  eval() or p == 0 turns it off.
"""
from __future__ import annotations

import copy
from functools import partial
from typing import Optional, Sequence

import torch
import torch.nn as nn

from . import ops
from ._lib import BF16, F32

# 'bf16x3' = 'bf16' with the KEY/VALUE side of the decoder's cross-attention (hoisted K/V projections, scores, value contraction,
# out-projection: the only single-bf16 forward products of 'bf16') carried as hi + lo bf16 pairs, 3 MFMAs per product -- fp32-grade
# forward (north_star's 1e-3 end to end), single-bf16 backward.  Everything else is identical to 'bf16', hence the same table entry;
# the fused executor (fused.fused_decoder) and CrossAttentionLayer.branch look at the mode string itself.
CT = {"bf16": BF16, "fp32": F32, "bf16x3": BF16}
# dropout-site bases of the module roles (a second instance of a role in one model may be given its own
# ``_drop_base``; instances sharing a base draw identical masks for identical shapes)
DROP_BASE_ENCODER, DROP_BASE_MASK_HEAD, DROP_BASE_GROUND_HEAD, DROP_BASE_OBJ_ENC, DROP_BASE_LAYER = \
    (1 << 20, 2 << 20, 3 << 20, 4 << 20, 5 << 20)


def set_compute(module: nn.Module, compute: str) -> nn.Module:
    """Select the 'bf16', 'bf16x3' or 'fp32' MFMA path for every pq3d module below ``module``."""
    assert compute in CT
    for m in module.modules():
        if hasattr(m, "compute"):
            m.compute = compute
    return module


def set_dropout(module: nn.Module, p: float) -> nn.Module:
    """Override every dropout probability below ``module`` (layers, heads, encoders); memory_dropout is untouched."""
    for m in module.modules():
        if hasattr(m, "dropout_p"):
            m.dropout_p = float(p)
        if hasattr(m, "cls_dropout_p"):
            m.cls_dropout_p = float(p)
        if hasattr(m, "cls_head") and isinstance(m.cls_head, nn.Sequential):
            # the nn.Dropout inside get_mlp_head's Sequential mirrors the rate the kernels apply (dropout_p / cls_dropout_p): kept in
            # step, so that a model with every rate at zero does not advance the dropout RNG (two launches per step) for nothing
            for sub in m.cls_head:
                if isinstance(sub, nn.Dropout):
                    sub.p = float(p)
    return module


def begin_dropout_step(owner: nn.Module, device, extra: Sequence[nn.Module] = ()) -> None:
    """Start of a training forward of a top-level module: make the dropout RNG epoch new for this call (unless a
    parent module already did for this step) and tell the dropout-bearing children that their site ids are managed
    structurally in this epoch (heads then use (base, call index) instead of advancing the RNG themselves)."""
    rng = ops.drop_rng(device)
    # the module tree is walked once (0.6 ms of host time per eager step otherwise); rates are re-read every call
    cache = owner.__dict__.get("_pq3d_drop_cache")
    if cache is None or cache[0] != len(extra):
        tree = list(owner.modules())
        # the nn.Dropout inside a head's get_mlp_head Sequential is a placeholder (state_dict / module-tree parity with the
        # reference): the rate that is APPLIED there is the head's dropout_p / cls_dropout_p, already looked at through `mods`
        mirrored = {id(sub) for m in tree + list(extra) if isinstance(m, _PostNormBase) and isinstance(getattr(m, "cls_head", None), nn.Sequential)
                    for sub in m.cls_head if isinstance(sub, nn.Dropout)}
        cache = (len(extra), [m for m in tree + list(extra) if isinstance(m, _PostNormBase)],
                 [m for m in tree if isinstance(m, nn.Dropout) and id(m) not in mirrored])
        owner.__dict__["_pq3d_drop_cache"] = cache
    mods, nn_drops = cache[1], cache[2]
    # nothing draws from the generator when every rate is zero: skip the (device-side) epoch advance then
    active = any(getattr(m, "dropout_p", 0.0) > 0 or (hasattr(m, "cls_head") and getattr(m, "cls_dropout_p", 0.0) > 0)
                 for m in mods) or any(m.p > 0 for m in nn_drops)
    if active and rng.epoch == getattr(owner, "_drop_epoch", -1):
        rng.advance()
    owner._drop_epoch = rng.epoch if active else -1
    for m in mods:
        m._drop_managed = rng.epoch


def _init_weights_bert(module: nn.Module, std: float = 0.02) -> None:
    """modules/weights.py:3-20."""
    if isinstance(module, nn.Linear):
        module.weight.data.normal_(mean=0.0, std=std)
        if module.bias is not None:
            module.bias.data.zero_()
    elif isinstance(module, nn.LayerNorm):
        module.bias.data.zero_()
        module.weight.data.fill_(1.0)


def layer_repeat(module: nn.Module, N: int, share_layer: bool = False) -> nn.ModuleList:
    """modules/utils.py:28-32 (deep copies: all repeats start identical)."""
    if share_layer:
        return nn.ModuleList([module] * N)
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N - 1)] + [module])


def get_mlp_head(input_size: int, hidden_size: int, output_size: int, dropout: float = 0) -> nn.Sequential:
    """Parameter container with the key layout of modules/utils.py:18-25 ('0', '2', '4')."""
    return nn.Sequential(nn.Linear(input_size, hidden_size), nn.ReLU(), nn.LayerNorm(hidden_size, eps=1e-12),
                         nn.Dropout(dropout), nn.Linear(hidden_size, output_size))


def mlp_head_forward(seq: nn.Sequential, x: torch.Tensor, ct: int, fill_flag=None, fill_value=0.0,
                     drop=None) -> torch.Tensor:
    ct = ops.small_ct(ct)   # heads run on the B*N_q query rows: split-bf16 in 'bf16' mode
    h = ops.linear(x, seq[0].weight, seq[0].bias, ct=ct, act="relu", out_dtype=torch.float32)
    h = ops.add_layernorm(None, [h], [seq[2].weight], [seq[2].bias], eps=seq[2].eps)
    h = ops.dropout(h, drop)   # nn.Dropout between LayerNorm and the last Linear (utils.py:23)
    return ops.linear(h, seq[4].weight, seq[4].bias, ct=ct, fill_flag=fill_flag, fill_value=fill_value)


def linear_ln_forward(seq: nn.Sequential, x: torch.Tensor, ct: int) -> torch.Tensor:
    """nn.Sequential(Linear, LayerNorm) as used by the encoders (object_encoder.py:34, query3d_unified.py:20,63-70)."""
    # the one-encoder case of the grouped Linear+LN op: 2 launches forward; its backward zero-fills every atomics target
    # (LayerNorm parameter gradients, split-K weight gradient, bias column sums) with ONE fill.
    # 'bf16' mode: split-bf16 forward product.  The encoder outputs (segment features, positions, caption-head input) feed
    # every layer's keys and values, so their operand rounding (1.1e-3 per Linear) would sit under everything downstream:
    # 3.6e-3 of the final query at config 2 by itself (measured with the oracle's rounding emulation); 3 MFMAs on these
    # B*N_seg x d_in x d products cost ~25 us per step.
    return ops.linear_ln_group([x], [seq[0].weight], [seq[0].bias], [seq[1].weight], [seq[1].bias], ct=ops.small_ct(ct),
                               eps=seq[1].eps)[0]


def _xavier(module: nn.Module) -> None:
    for p in module.parameters():
        if p.dim() > 1:
            nn.init.xavier_uniform_(p)


class _MHAParams(nn.Module):
    """Parameter layout of nn.MultiheadAttention (packed in_proj + out_proj)."""

    def __init__(self, d_model: int, nhead: int):
        super().__init__()
        assert d_model % nhead == 0
        self.embed_dim, self.num_heads = d_model, nhead
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d_model, d_model))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d_model))
        self.out_proj = nn.Linear(d_model, d_model)
        nn.init.xavier_uniform_(self.in_proj_weight)


def _split_mask(mask: Optional[torch.Tensor], B: int, H: int):
    """2-D mask -> key padding; 3-D [B,Lq,Lk] or the reference's [B*H,Lq,Lk] (head-broadcast) -> attn mask."""
    if mask is None:
        return None, None
    if mask.ndim == 2:
        return mask, None
    if mask.shape[0] == B * H and H > 1:
        mask = mask.view(B, H, *mask.shape[1:])[:, 0]
    return None, mask


class _PostNormBase(nn.Module):
    def __init__(self):
        super().__init__()
        self.compute = "bf16"
        self.dropout_p = 0.0
        self._drop_base = DROP_BASE_LAYER
        self._drop_epoch = -1
        self._drop_managed = -1    # RNG epoch in which a parent manages this module's dropout sites
        self._drop_call = 0        # call index within a managed epoch (mask head: layer application)

    @property
    def ct(self) -> int:
        return CT[self.compute]

    @property
    def ct_q(self) -> int:
        """Compute type of the query-side GEMMs (M = B*N_q rows): split-bf16 in 'bf16' mode (ops.small_ct)."""
        return ops.small_ct(CT[self.compute])

    def _drop_ctx(self, device, ctx=None, p=None):
        """(site base, layer application) for this call, or None when dropout is inactive.  A caller higher up (the
        encoder) hands its ctx down; a module used on its own makes sure the RNG epoch is fresh for every call."""
        if not self.training or not ((self.dropout_p if p is None else p) > 0.0):
            return None
        if ctx is not None:
            return ctx
        rng = ops.drop_rng(device)
        if rng.epoch == self._drop_epoch:
            rng.advance()
        self._drop_epoch = rng.epoch
        return (self._drop_base, 0)

    def _head_ctx(self, device, p=None):
        """ctx for a head / encoder: structural (base, call) when a parent manages this epoch, else standalone."""
        if not self.training or not ((self.dropout_p if p is None else p) > 0.0):
            return None
        if self._drop_managed == ops.drop_rng(device).epoch:
            return (self._drop_base, self._drop_call)
        return self._drop_ctx(device, None, p)

    def _drop(self, ctx, kind, device, m=0, p=None):
        if ctx is None:
            return None
        return ops.make_drop(self.dropout_p if p is None else p, ops.drop_site(ctx[0], ctx[1], kind, m), device)


class CrossAttentionLayer(_PostNormBase):
    """query_encoder.py:257-351 (add_zero_attn=True; post-norm :288-307 as every shipped config uses it, pre-norm :309-335 on the
    modular path: residual + dropout ride the out-projection's GEMM epilogue)."""

    def __init__(self, d_model, nhead, dropout=0.0, activation="relu", normalize_before=False, batch_first=False):
        super().__init__()
        self.normalize_before = bool(normalize_before)
        self.multihead_attn = _MHAParams(d_model, nhead)
        self.norm = nn.LayerNorm(d_model)
        self.nhead = nhead
        self.dropout_p = float(dropout)
        _xavier(self)

    def branch(self, tgt, memory, attn_mask=None, memory_key_padding_mask=None, pos=None, query_pos=None,
               row_open=None, _drop=None, _m=0, residual=None, res_drop=None) -> torch.Tensor:
        """out_proj(MHA(tgt+query_pos, memory+pos, memory)) -- the pre-residual branch output, fp32 (pre-norm: `residual +
        dropout(.)` formed by the out-projection's epilogue)."""
        ct, d = self.ct, tgt.shape[-1]
        if self.compute == "bf16x3":
            # modular path (structures the fused executor does not cover): the key/value side on the exact-f32 kernels, forward and
            # backward -- the mode's accuracy contract without its split-bf16 kernels (those live in the fused executor)
            ct = F32
        w, b = self.multihead_attn.in_proj_weight, self.multihead_attn.in_proj_bias
        ad = ops.act_dtype(ct)
        # the query projection (B*N_q rows) is formed at fp32 grade (split-bf16) and rounded once to the attention
        # operand type; K / V (B*N_seg rows: the bulk of the FLOPs) are single bf16 products
        q = ops.linear(tgt, w[:d], b[:d], x2=query_pos, ct=self.ct_q, out_dtype=ad)
        k = ops.linear(memory, w[d:2 * d], b[d:2 * d], x2=pos, ct=ct, out_dtype=ad)
        v = ops.linear(memory, w[2 * d:], b[2 * d:], ct=ct, out_dtype=ad)
        o = ops.attention(q, k, v, H=self.nhead, ct=ct, zero_attn=True, kpm=memory_key_padding_mask, mask=attn_mask,
                          row_open=row_open, drop=self._drop(_drop, ops.DROP_CA_ATTN, tgt.device, _m))
        return ops.linear(o, self.multihead_attn.out_proj.weight, self.multihead_attn.out_proj.bias, ct=ct, residual=residual,
                          drop=res_drop)

    def forward(self, tgt, memory, attn_mask=None, memory_key_padding_mask=None, pos=None, query_pos=None,
                row_open=None, _drop=None, _m=0):
        ctx = self._drop_ctx(tgt.device, _drop)
        if self.normalize_before:   # query_encoder.py:309-335: tgt + dropout(MHA(norm(tgt) + query_pos, memory + pos, memory))
            t2 = ops.add_layernorm(None, [tgt], [self.norm.weight], [self.norm.bias], eps=self.norm.eps)
            return self.branch(t2, memory, attn_mask, memory_key_padding_mask, pos, query_pos, row_open, _drop=ctx, _m=_m,
                               residual=tgt, res_drop=self._drop(ctx, ops.DROP_CA_RES, tgt.device, _m))
        o = self.branch(tgt, memory, attn_mask, memory_key_padding_mask, pos, query_pos, row_open, _drop=ctx, _m=_m)
        return ops.add_layernorm(tgt, [o], [self.norm.weight], [self.norm.bias], eps=self.norm.eps, coef=None,
                                 drop=self._drop(ctx, ops.DROP_CA_RES, tgt.device, _m))


class SelfAttentionLayer(_PostNormBase):
    """query_encoder.py:184-254 (stock MHA, no zero-attn; pre-norm :229-243 on the modular path)."""

    def __init__(self, d_model, nhead, dropout=0.0, activation="relu", normalize_before=False, batch_first=False):
        super().__init__()
        self.normalize_before = bool(normalize_before)
        self.self_attn = _MHAParams(d_model, nhead)
        self.norm = nn.LayerNorm(d_model)
        self.nhead = nhead
        self.dropout_p = float(dropout)
        _xavier(self)

    def forward(self, tgt, attn_mask=None, tgt_key_padding_mask=None, query_pos=None, _drop=None):
        # self-attention works on N_q x N_q scores per scene (a negligible share of the FLOPs): projections at fp32 grade
        # (split-bf16 in 'bf16' mode), the attention core on the exact-f32 MFMA path in both modes
        ct, d = self.ct_q, tgt.shape[-1]
        ctx = self._drop_ctx(tgt.device, _drop)
        w, b = self.self_attn.in_proj_weight, self.self_attn.in_proj_bias
        src = tgt
        if self.normalize_before:   # :229-243: q = k = norm(tgt) + query_pos, value = norm(tgt)
            src = ops.add_layernorm(None, [tgt], [self.norm.weight], [self.norm.bias], eps=self.norm.eps)
        q = ops.linear(src, w[:d], b[:d], x2=query_pos, ct=ct)
        k = ops.linear(src, w[d:2 * d], b[d:2 * d], x2=query_pos, ct=ct)
        v = ops.linear(src, w[2 * d:], b[2 * d:], ct=ct)
        o = ops.attention(q, k, v, H=self.nhead, ct=ops.sa_ct(self.ct), kpm=tgt_key_padding_mask, mask=attn_mask,
                          drop=self._drop(ctx, ops.DROP_SA_ATTN, tgt.device))
        if self.normalize_before:
            return ops.linear(o, self.self_attn.out_proj.weight, self.self_attn.out_proj.bias, ct=ct, residual=tgt,
                              drop=self._drop(ctx, ops.DROP_SA_RES, tgt.device))
        o = ops.linear(o, self.self_attn.out_proj.weight, self.self_attn.out_proj.bias, ct=ct)
        return ops.add_layernorm(tgt, [o], [self.norm.weight], [self.norm.bias], eps=self.norm.eps,
                                 drop=self._drop(ctx, ops.DROP_SA_RES, tgt.device))


class MultiHeadAttentionSpatial(_PostNormBase):
    """modules/layers/transformers.py:158-240, spatial_multihead=True, spatial_attn_fusion 'mul' (what the decoder instantiates,
    query_encoder.py:411-416: softmax(log(clamp(relu(W loc), 1e-6)) + qk)) or 'bias' (softmax(W loc + qk), :196-203, :228-230).
    'add', 'ctx' and 'cond' belong to model families outside the query-decoder path (TransformerSpatialDecoderLayer) and raise."""

    def __init__(self, d_model, n_head, dropout=0.1, spatial_multihead=True, spatial_dim=5, spatial_attn_fusion="mul"):
        super().__init__()
        if spatial_attn_fusion not in ("mul", "bias") or not spatial_multihead or spatial_dim != 5:
            raise NotImplementedError("spatial_attn_fusion 'mul' / 'bias', multihead, spatial_dim=5 are implemented "
                                      "(the decoder instantiates 'mul': query_encoder.py:411-416)")
        self.spatial_attn_fusion = spatial_attn_fusion
        assert d_model % n_head == 0
        self.n_head, self.d_model = n_head, d_model
        self.w_qs, self.w_ks, self.w_vs = (nn.Linear(d_model, d_model) for _ in range(3))
        self.fc = nn.Linear(d_model, d_model)
        self.pairwise_loc_fc = nn.Linear(spatial_dim, n_head)

    def forward(self, q, k, v, pairwise_locs, key_padding_mask=None, q_pos=None, k_pos=None, residual=None, res_drop=None):
        """Returns only the output (the reference also returns the fused attention map, which the decoder
        discards, query_encoder.py:447).  q_pos/k_pos are added to q/k inside the projection kernels; residual / res_drop: the
        pre-norm layer's `residual + dropout(.)` in fc's epilogue."""
        ct = self.ct_q   # as SelfAttentionLayer: fp32-grade projections, exact-f32 attention core
        qh = ops.linear(q, self.w_qs.weight, self.w_qs.bias, x2=q_pos, ct=ct)
        kh = ops.linear(k, self.w_ks.weight, self.w_ks.bias, x2=k_pos, ct=ct)
        vh = ops.linear(v, self.w_vs.weight, self.w_vs.bias, ct=ct)
        if self.spatial_attn_fusion == "mul":
            bias = ops.spatial_bias(pairwise_locs, self.pairwise_loc_fc.weight, self.pairwise_loc_fc.bias)
        else:   # 'bias': the plain 5 -> H projection as the additive term (a [B L T, 5] x [5, H] product at fp32 grade)
            Bq, Lq, Tk, _ = pairwise_locs.shape
            pl = ops.linear(pairwise_locs.reshape(Bq * Lq * Tk, -1).contiguous(), self.pairwise_loc_fc.weight,
                            self.pairwise_loc_fc.bias, ct=F32)
            bias = pl.view(Bq, Lq, Tk, self.n_head).permute(0, 3, 1, 2).contiguous()
        o = ops.attention(qh, kh, vh, H=self.n_head, ct=ops.sa_ct(self.ct), kpm=key_padding_mask, bias=bias)
        return ops.linear(o, self.fc.weight, self.fc.bias, ct=ct, residual=residual, drop=res_drop)


class SpatialSelfAttentionLayer(_PostNormBase):
    """query_encoder.py:402-483 (pre-norm :453-468 on the modular path -- NB its VALUE input is the un-normalised tgt)."""

    def __init__(self, d_model, nhead, dropout=0.0, activation="relu", normalize_before=False, batch_first=False,
                 spatial_multihead=True, spatial_dim=5, spatial_attn_fusion="mul"):
        super().__init__()
        self.normalize_before = bool(normalize_before)
        self.self_attn = MultiHeadAttentionSpatial(d_model, nhead, dropout=dropout, spatial_multihead=spatial_multihead,
                                                   spatial_dim=spatial_dim, spatial_attn_fusion=spatial_attn_fusion)
        self.norm = nn.LayerNorm(d_model)
        self.dropout_p = float(dropout)   # residual dropout only: MultiHeadAttentionSpatial never drops its weights
        _xavier(self)

    def forward(self, tgt, attn_mask=None, tgt_key_padding_mask=None, query_pos=None, pairwise_locs=None, _drop=None):
        ctx = self._drop_ctx(tgt.device, _drop)
        if self.normalize_before:
            t2 = ops.add_layernorm(None, [tgt], [self.norm.weight], [self.norm.bias], eps=self.norm.eps)
            return self.self_attn(t2, t2, tgt, pairwise_locs, key_padding_mask=tgt_key_padding_mask, q_pos=query_pos, k_pos=query_pos,
                                  residual=tgt, res_drop=self._drop(ctx, ops.DROP_SA_RES, tgt.device))
        o = self.self_attn(tgt, tgt, tgt, pairwise_locs, key_padding_mask=tgt_key_padding_mask, q_pos=query_pos,
                           k_pos=query_pos)
        return ops.add_layernorm(tgt, [o], [self.norm.weight], [self.norm.bias], eps=self.norm.eps,
                                 drop=self._drop(ctx, ops.DROP_SA_RES, tgt.device))


class FFNLayer(_PostNormBase):
    """query_encoder.py:354-399 (pre-norm :390-394 on the modular path)."""

    def __init__(self, d_model, dim_feedforward=2048, dropout=0.0, activation="relu", normalize_before=False):
        super().__init__()
        self.normalize_before = bool(normalize_before)
        if activation not in ("relu", "gelu"):
            raise RuntimeError(f"activation function currently support relu/gelu, not {activation}")
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm = nn.LayerNorm(d_model)
        self.activation = activation
        self.dropout_p = float(dropout)
        _xavier(self)

    def forward(self, tgt, _drop=None):
        ct = self.ct_q   # split-bf16 in 'bf16' mode: the ReLU kink is hit at fp32 grade, the hidden activations stay fp32
        ctx = self._drop_ctx(tgt.device, _drop)
        if self.normalize_before:   # tgt + dropout(linear2(dropout(act(linear1(norm(tgt))))))
            t2 = ops.add_layernorm(None, [tgt], [self.norm.weight], [self.norm.bias], eps=self.norm.eps)
            h = ops.linear(t2, self.linear1.weight, self.linear1.bias, ct=ct, act=self.activation, out_dtype=ops.act_dtype(ct),
                           drop=self._drop(ctx, ops.DROP_FFN_INNER, tgt.device))
            return ops.linear(h, self.linear2.weight, self.linear2.bias, ct=ct, residual=tgt,
                              drop=self._drop(ctx, ops.DROP_FFN_RES, tgt.device))
        h = ops.linear(tgt, self.linear1.weight, self.linear1.bias, ct=ct, act=self.activation,
                       out_dtype=ops.act_dtype(ct), drop=self._drop(ctx, ops.DROP_FFN_INNER, tgt.device))
        # linear2's K range is split into 4 partial GEMMs whose fp32 outputs the LayerNorm kernel adds in a fixed order
        # (same rounding as the fused executor, which runs the 4 parts as one grouped launch: fused.py)
        F_ = h.shape[-1]
        KS = 4 if F_ % (4 * 64) == 0 else 1
        Fk = F_ // KS
        ys = [ops.linear(h[..., k * Fk:(k + 1) * Fk], self.linear2.weight[:, k * Fk:(k + 1) * Fk],
                         self.linear2.bias if k == 0 else None, ct=ct) for k in range(KS)]
        return ops.add_layernorm(tgt, ys, [self.norm.weight], [self.norm.bias], eps=self.norm.eps,
                                 drop=self._drop(ctx, ops.DROP_FFN_RES, tgt.device), sum_branches=True)


def memory_keep_coef(B: int, M: int, p: float, device, keep=None) -> torch.Tensor:
    """Training-time memory dropout of the parallel cross-attention (query_encoder.py:145-151): per (scene, memory)
    Bernoulli keep with at least one memory kept per scene (a scene that drew none keeps all), returned as the weights
    of the masked mean, [M, B] fp32.  ``keep`` [B, M] bool overrides the draw."""
    if keep is None:
        keep = torch.rand(B, M, device=device) > p
    keep = keep.to(device=device, dtype=torch.bool)
    keep = torch.logical_or(keep, keep.sum(1, keepdim=True) == 0)
    return (keep / keep.sum(1, keepdim=True)).t().contiguous().float()


class QueryEncoderLayer(_PostNormBase):
    """query_encoder.py:96-181."""

    def __init__(self, d_model, nhead, memories, dim_feedforward=2048, dropout=0.1, activation="relu", prenorm=False,
                 spatial_selfattn=False, structure="mixed", memory_dropout=0, drop_memories_test=[]):
        super().__init__()
        if spatial_selfattn:
            self.self_attn = SpatialSelfAttentionLayer(d_model, nhead, dropout=dropout, activation=activation,
                                                       normalize_before=prenorm, batch_first=True)
        else:
            self.self_attn = SelfAttentionLayer(d_model, nhead, dropout=dropout, activation=activation,
                                                normalize_before=prenorm, batch_first=True)
        ca = CrossAttentionLayer(d_model, nhead, dropout=dropout, activation=activation, normalize_before=prenorm,
                                 batch_first=True)
        self.cross_attn_list = layer_repeat(ca, len(memories))
        self.memory2ca = {m: c for m, c in zip(memories, self.cross_attn_list)}
        self.ffn = FFNLayer(d_model, dim_feedforward, dropout=dropout, activation=activation, normalize_before=prenorm)
        self.structure = structure
        self.memories = list(memories)
        self.memory_dropout = memory_dropout
        self.drop_memories_test = list(drop_memories_test)
        self.dropout_p = float(dropout)
        if structure == "gate":
            self.gate_proj = nn.Linear(d_model, d_model)

    def forward(self, query, input_dict, pairwise_locs=None, _drop=None, _mem_keep=None):
        """``_mem_keep`` [B, len(scene memories)] bool: externally drawn training-time memory-dropout mask (tests feed
        the same mask to the oracle); None -> drawn here with torch.rand as the reference does (:145)."""
        _, query_masks, query_pos = input_dict["query"][:3]
        dctx = self._drop_ctx(query.device, _drop)
        B, H = query.shape[0], self.cross_attn_list[0].nhead if len(self.cross_attn_list) else 1
        row_open = input_dict.get("_attn_row_open")

        def ca_args(memory):
            feat, mask, pos = input_dict[memory][:3]
            kpm, am = _split_mask(mask, B, H)
            return dict(memory=feat, attn_mask=am, memory_key_padding_mask=kpm, pos=pos, query_pos=query_pos,
                        row_open=row_open if am is not None else None)

        # dropout sites of the cross-attention sublayers: memory slot = position in the list handed to parallel_ca
        # (what the fused executor stacks), 4 + position for sequential_ca
        def sequential_ca(q, memories):
            for j, m in enumerate(memories):
                q = self.memory2ca[m](q, **ca_args(m), _drop=dctx, _m=4 + j)
            return q

        def parallel_ca(q, memories):
            assert "prompt" not in memories
            cas = [self.memory2ca[m] for m in memories]
            coef = None
            if self.training and self.memory_dropout > 0.0:  # query_encoder.py:145-151
                coef = memory_keep_coef(B, len(memories), self.memory_dropout, q.device, _mem_keep)  # [M,B]
            if cas and cas[0].normalize_before:
                # pre-norm (not a shipped configuration): every memory's layer output is q + dropout(branch(norm(q))); their (masked)
                # mean (query_encoder.py:145-152) has no LayerNorm to merge into -- a plain weighted sum of the M outputs
                full = [ca(q, **ca_args(m), _drop=dctx, _m=j) for j, (ca, m) in enumerate(zip(cas, memories))]
                if coef is None:
                    return torch.stack(full, 0).mean(0)
                return (torch.stack(full, 0) * coef[:, :, None, None]).sum(0)
            outs = [ca.branch(q, **ca_args(m), _drop=dctx, _m=j) for j, (ca, m) in enumerate(zip(cas, memories))]
            return ops.add_layernorm(q, outs, [c.norm.weight for c in cas], [c.norm.bias for c in cas],
                                     eps=cas[0].norm.eps, coef=coef,
                                     drop=cas[0]._drop(dctx, ops.DROP_CA_RES, q.device) if cas else None)

        memories = self.memories if self.training else [m for m in self.memories if m not in self.drop_memories_test]
        if self.structure == "sequential":
            query = sequential_ca(query, memories)
        elif self.structure == "parallel":
            query = parallel_ca(query, memories)
        elif self.structure == "mixed":
            query = parallel_ca(query, [m for m in memories if m != "prompt"])
            query = sequential_ca(query, ["prompt"])
        elif self.structure == "gate":
            prompt = sequential_ca(query, ["prompt"])
            gate = ops.linear(prompt, self.gate_proj.weight, self.gate_proj.bias, ct=self.ct_q)
            update = parallel_ca(query, [m for m in self.memories if m != "prompt"])
            query = ops.gate_mix(query, update, gate)
        else:
            raise NotImplementedError(f"Unknow structure type: {self.structure}")

        if isinstance(self.self_attn, SpatialSelfAttentionLayer):
            query = self.self_attn(query, tgt_key_padding_mask=query_masks, query_pos=query_pos,
                                   pairwise_locs=pairwise_locs, _drop=dctx)
        else:
            query = self.self_attn(query, tgt_key_padding_mask=query_masks, query_pos=query_pos, _drop=dctx)
        return self.ffn(query, _drop=dctx)


class QueryMaskEncoder(nn.Module):
    """query_encoder.py:52-94.  forward(input_dict, pairwise_locs, mask_head=None) ->
    (query, predictions_class, predictions_mask); mutates input_dict[memory][1] like the reference, except that the
    self-mask is kept as [B,Nq,Ns] (head broadcast in-kernel) and fully-masked rows are opened through a [B,Nq] flag
    (input_dict['_attn_row_open']) instead of rewriting / repeat_interleaving the mask."""

    def __init__(self, cfg, memories=[], memory_dropout=0.0, hidden_size=768, num_attention_heads=12, num_layers=4,
                 share_layer=False, spatial_selfattn=False, structure="sequential", drop_memories_test=[],
                 use_self_mask=False, num_blocks=1, activation="relu", compute="bf16"):
        super().__init__()
        self.spatial_selfattn = spatial_selfattn
        layer = QueryEncoderLayer(hidden_size, num_attention_heads, memories, spatial_selfattn=spatial_selfattn,
                                  structure=structure, memory_dropout=memory_dropout,
                                  drop_memories_test=drop_memories_test, activation=activation)
        self.unified_encoder = layer_repeat(layer, num_layers, share_layer)
        self.apply(_init_weights_bert)
        self.memory_dropout = memory_dropout
        self.scene_meomories = [x for x in memories if x != "prompt"]
        self.drop_memories_test = drop_memories_test
        self.use_self_mask = use_self_mask
        self.num_heads = num_attention_heads
        self.num_blocks = num_blocks
        # test hook: callable(app, B, M, device) -> [B, M] bool keep-mask of layer application `app` (None: torch.rand)
        self.memory_keep_hook = None
        self.fused = True          # use the fused executor (fused.py) whenever the configuration allows it
        self._fused_final = None   # (cls, mask_logits) of the trailing mask-head call computed by the fused path
        self._drop_base, self._drop_epoch = DROP_BASE_ENCODER, -1
        set_compute(self, compute)

    def _try_fused(self, input_dict, pairwise_locs, mask_head):
        """Fused executor for structure='parallel' with `mask_head` either None or a functools.partial of a
        MaskHeadSegLevel (what Query3DUnified builds).  Returns None when the configuration is not covered."""
        from . import fused as F
        mh_mod, kw = None, {}
        if mask_head is not None:
            func = getattr(mask_head, "func", None)
            owner = func if isinstance(func, MaskHeadSegLevel) else getattr(func, "__self__", None)
            if not (isinstance(mask_head, partial) and isinstance(owner, MaskHeadSegLevel) and not mask_head.args):
                return None
            mh_mod, kw = owner, dict(mask_head.keywords)
            kw.pop("keys", None)
        try:
            query, pcls, pmask = F.fused_decoder(self, input_dict, pairwise_locs, mh_mod,
                                                 kw.get("seg_fts_for_match"), kw.get("seg_masks"),
                                                 kw.get("offline_attn_masks"), kw.get("skip_prediction", False))
        except NotImplementedError:
            return None
        if mh_mod is not None:
            self._fused_final = (pcls[-1], pmask[-1])
            pcls, pmask = pcls[:-1], pmask[:-1]
        return query, pcls, pmask

    def forward(self, input_dict, pairwise_locs, mask_head=None):
        self._fused_final = None
        mh_owner = None
        if isinstance(mask_head, partial):
            func = mask_head.func
            mh_owner = func if isinstance(func, MaskHeadSegLevel) else getattr(func, "__self__", None)
            mh_owner = mh_owner if isinstance(mh_owner, MaskHeadSegLevel) else None
        n_app = self.num_blocks * len(self.unified_encoder)
        if self.training:
            begin_dropout_step(self, input_dict["query"][0].device, [mh_owner] if mh_owner is not None else ())
        if mh_owner is not None:
            mh_owner._drop_call = n_app   # the trailing call the model makes after the encoder
        if self.fused:
            out = self._try_fused(input_dict, pairwise_locs, mask_head)
            if out is not None:
                return out
        predictions_class, predictions_mask = [], []
        query = input_dict["query"][0]
        voxel_feat = input_dict["voxel"][0] if "voxel" in input_dict.keys() else None
        attn_mask = None
        for _block in range(self.num_blocks):
            for i, layer in enumerate(self.unified_encoder):
                app = _block * len(self.unified_encoder) + i
                if mask_head is not None:
                    if mh_owner is not None:
                        mh_owner._drop_call = app
                    output_class, outputs_mask, attn_mask = mask_head(query)
                    predictions_class.append(output_class)
                    predictions_mask.append(outputs_mask)
                if self.use_self_mask:
                    if attn_mask.shape[0] != query.shape[0]:
                        attn_mask = attn_mask.view(query.shape[0], -1, *attn_mask.shape[1:])[:, 0]
                    input_dict["_attn_row_open"] = ops.mask_row_all(attn_mask)
                    for memory in input_dict.keys():
                        if memory in ("query", "prompt") or memory.startswith("_"):
                            continue
                        input_dict[memory][1] = attn_mask
                if isinstance(voxel_feat, list):
                    input_dict["voxel"][0] = voxel_feat[i]
                mk = None
                if self.training and layer.memory_dropout > 0.0 and self.memory_keep_hook is not None:
                    mk = self.memory_keep_hook(app, query.shape[0], len(layer.memories), query.device)
                query = layer(query, input_dict, pairwise_locs, _drop=(self._drop_base, app) if self.training else None,
                              _mem_keep=mk)
        if mh_owner is not None:
            mh_owner._drop_call = n_app
        return query, predictions_class, predictions_mask


class QueryEncoder(nn.Module):
    """query_encoder.py:11-49 (no mask head / self mask)."""

    def __init__(self, cfg, memories=[], memory_dropout=0.0, hidden_size=768, num_attention_heads=12, num_layers=4,
                 share_layer=False, spatial_selfattn=False, structure="sequential", drop_memories_test=[],
                 compute="bf16"):
        super().__init__()
        self.spatial_selfattn = spatial_selfattn
        layer = QueryEncoderLayer(hidden_size, num_attention_heads, memories, spatial_selfattn=spatial_selfattn,
                                  structure=structure)
        self.unified_encoder = layer_repeat(layer, num_layers, share_layer)
        self.apply(_init_weights_bert)
        self.memory_dropout = memory_dropout
        self.scene_meomories = [x for x in memories if x != "prompt"]
        self.drop_memories_test = drop_memories_test
        self.memory_drop_hook = None   # test hook: callable(memory, B, device) -> [B] bool drop-mask (None: torch.rand)
        self._drop_base, self._drop_epoch = DROP_BASE_ENCODER, -1
        set_compute(self, compute)

    def dropout_memory(self, input_dict):
        """query_encoder.py:26-37: zero the features AND the position rows of dropped scenes -- training: per (scene,
        memory) Bernoulli(memory_dropout); eval: every scene of the memories in ``drop_memories_test``.  The reference
        writes in place, and the scene memories share ONE position tensor (query3d_unified.py:124-153), so a scene dropped
        for any memory loses its positions for all memories that share the tensor: reproduced by zeroing shared position
        tensors once with the union of their memories' masks.  (Out of place here: the inputs may require grad.)"""
        masks = {}
        for memory in self.scene_meomories:
            feat = input_dict[memory][0]
            B, dev = feat.shape[0], feat.device
            if self.training:
                dm = self.memory_drop_hook(memory, B, dev) if self.memory_drop_hook is not None \
                    else torch.rand(B, device=dev) < self.memory_dropout
            else:
                dm = torch.full((B,), memory in self.drop_memories_test, dtype=torch.bool, device=dev)
            masks[memory] = dm.to(device=dev, dtype=torch.bool)
        pos_union = {}
        for memory in self.scene_meomories:
            pos = input_dict[memory][2]
            if pos is not None:
                k = id(pos)
                pos_union[k] = masks[memory] if k not in pos_union else (pos_union[k] | masks[memory])
        new_pos = {}
        for memory in self.scene_meomories:
            feat, mask, pos = input_dict[memory][:3]
            keep = masks[memory].logical_not().to(feat.dtype)[:, None, None]
            if pos is not None:
                k = id(pos)
                if k not in new_pos:
                    new_pos[k] = pos * pos_union[k].logical_not().to(pos.dtype)[:, None, None]
                pos = new_pos[k]
            input_dict[memory] = [feat * keep, mask, pos] + list(input_dict[memory][3:])

    def forward(self, input_dict, pairwise_locs):
        if (self.training and self.memory_dropout > 0) or (not self.training and self.drop_memories_test):
            self.dropout_memory(input_dict)
        query = input_dict["query"][0]
        if self.training:
            begin_dropout_step(self, query.device)
        voxel_feat = input_dict["voxel"][0] if "voxel" in input_dict.keys() else None
        for i, layer in enumerate(self.unified_encoder):
            if isinstance(voxel_feat, list):
                input_dict["voxel"][0] = voxel_feat[i]
            query = layer(query, input_dict, pairwise_locs, _drop=(self._drop_base, i) if self.training else None)
        return query


# ------------------------------------------------------------------------------------------------ heads
class MaskPredictionLayer(nn.Module):
    """mask_head.py:46-57 (parameter container; the einsum runs inside ops.mask_logits)."""

    def __init__(self, hidden_size):
        super().__init__()
        self.q_proj = nn.Linear(hidden_size, hidden_size)
        self.k_proj = nn.Linear(hidden_size, hidden_size, False)


class MaskHeadSegLevel(_PostNormBase):
    """modules/heads/mask_head.py:11-44."""

    def __init__(self, cfg, hidden_size, num_targets, memories_for_match=["voxel"], filter_out_classes=None,
                 dropout=0.1):
        super().__init__()
        self.cls_head = get_mlp_head(hidden_size, hidden_size, num_targets, dropout=dropout)
        self.filter_out_classes = filter_out_classes
        memories_for_match = [m for m in memories_for_match if m in ("voxel", "mv", "pc")]
        self.mask_pred_list = layer_repeat(MaskPredictionLayer(hidden_size), len(memories_for_match))
        self.num_targets = num_targets
        self.dropout_p, self._drop_base = float(dropout), DROP_BASE_MASK_HEAD
        # reference quirk: filter_out_classes=None makes x[..., None] = -inf overwrite every logit (mask_head.py:28)
        foc = list(range(num_targets)) if filter_out_classes is None else list(filter_out_classes)
        self.register_buffer("_foc_cols", torch.tensor(foc, dtype=torch.int32), persistent=False)
        flags = torch.zeros(num_targets, dtype=torch.int32)
        flags[[c for c in foc if 0 <= c < num_targets]] = 1          # the same set as per-column flags (csrc/chain_mh.hip)
        self.register_buffer("_foc_flags", flags, persistent=False)

    def project_keys(self, seg_fts_for_match):
        """k_proj of every matching memory (rows of padded segments zeroed) + the masked-mean denominators.
        Layer-invariant unless the voxel memory is multi-scale: callers may compute it once per forward."""
        # split-bf16 in 'bf16' mode (fp32 keys): the mask logits decide the self-masks of every following layer, and a
        # single-bf16 key projection alone costs 1.4e-3 of their scale (profiles/parity_r02.txt); the projection is
        # layer-invariant (computed once per forward)
        ct = self.ct_q
        n = len(self.mask_pred_list)   # mask_head.py:31: zip() truncates to the shorter of the two lists
        seg_fts_for_match = list(seg_fts_for_match)[:n]
        keys = [ops.linear(feat, mp.k_proj.weight, None, ct=ct, out_dtype=ops.act_dtype(ct),
                           row_mask=mask.logical_not())
                for (feat, mask, _pos), mp in zip(seg_fts_for_match, self.mask_pred_list)]
        inv_den = ops.mask_inv_den([f[1] for f in seg_fts_for_match])
        return keys, inv_den

    def forward(self, query, seg_fts_for_match, seg_masks, offline_attn_masks=None, skip_prediction=False, keys=None):
        if skip_prediction:
            return None, None, offline_attn_masks
        ct = self.ct_q
        cls_logits = mlp_head_forward(self.cls_head, query, self.ct,
                                      drop=self._drop(self._head_ctx(query.device), ops.DROP_MLP_HEAD, query.device))
        if self._foc_cols.numel():
            cls_logits = ops.fill_cols(cls_logits, self._foc_cols, float("-inf"))
        k_list, inv_den = keys if keys is not None else self.project_keys(seg_fts_for_match)
        q_list = [ops.linear(query, mp.q_proj.weight, mp.q_proj.bias, ct=ct, out_dtype=ops.act_dtype(ct))
                  for mp in self.mask_pred_list[:len(k_list)]]
        mask_logits, attn_mask = ops.mask_logits(k_list, q_list, inv_den, seg_masks, ct=ct)
        if offline_attn_masks is not None:
            attn_mask = offline_attn_masks
        return cls_logits, mask_logits, attn_mask


class GroundHeadV1(_PostNormBase):
    """modules/heads/grounding_head.py:7-40: the grounding logits plus three auxiliary classification heads (text token 0, object
    embeddings, pre-fusion object embeddings), each a get_mlp_head."""

    def __init__(self, cfg, input_size=768, hidden_size=768, sem_cls_size=607, dropout=0.3, detach_all_aux_loss=False):
        super().__init__()
        self.og3d_head = get_mlp_head(input_size, hidden_size, 1, dropout=dropout)
        self.txt_clf_head = get_mlp_head(input_size, hidden_size, sem_cls_size, dropout=dropout)
        self.obj3d_clf_head = get_mlp_head(input_size, hidden_size, sem_cls_size, dropout=dropout)
        self.obj3d_clf_pre_head = get_mlp_head(input_size, hidden_size, sem_cls_size, dropout=dropout)
        self.detach_all_aux_loss = detach_all_aux_loss
        self.dropout_p, self._drop_base = float(dropout), DROP_BASE_GROUND_HEAD

    def forward(self, txt_embeds, obj_embeds, obj_pre_embeds, obj_masks, **kwargs):
        dev = obj_embeds.device
        ctx = self._head_ctx(dev)
        dr = lambda m: self._drop(ctx, ops.DROP_MLP_HEAD, dev, m)
        og3d = mlp_head_forward(self.og3d_head, obj_embeds, self.ct, fill_flag=obj_masks.logical_not(), fill_value=float("-inf"),
                                drop=dr(0)).squeeze(2)
        if self.detach_all_aux_loss:
            txt_embeds, obj_embeds, obj_pre_embeds = txt_embeds.detach(), obj_embeds.detach(), obj_pre_embeds.detach()
        txt = mlp_head_forward(self.txt_clf_head, txt_embeds[:, 0].contiguous(), self.ct, drop=dr(1))
        obj = mlp_head_forward(self.obj3d_clf_head, obj_embeds, self.ct, drop=dr(2))
        pre = mlp_head_forward(self.obj3d_clf_pre_head, obj_pre_embeds, self.ct, drop=dr(3))
        return txt, obj, pre, og3d


class GroundHead(_PostNormBase):
    """modules/heads/grounding_head.py:43-55."""

    def __init__(self, cfg, input_size=768, hidden_size=768, dropout=0.3):
        super().__init__()
        self.og3d_head = get_mlp_head(input_size, hidden_size, 1, dropout=dropout)
        self.dropout_p, self._drop_base = float(dropout), DROP_BASE_GROUND_HEAD

    def forward(self, obj_embeds, obj_masks=None, **kwargs):
        flag = obj_masks.logical_not() if obj_masks is not None else None
        dev = obj_embeds.device
        return mlp_head_forward(self.og3d_head, obj_embeds, self.ct, fill_flag=flag, fill_value=float("-inf"),
                                drop=self._drop(self._head_ctx(dev), ops.DROP_MLP_HEAD, dev)).squeeze(2)


T5_ARCH = {   # architectures of the HF checkpoints the reference names (used only when the checkpoint is not on disk)
    "t5-small": dict(vocab_size=32128, d_model=512, d_kv=64, d_ff=2048, num_layers=6, num_decoder_layers=6, num_heads=8,
                     decoder_start_token_id=0, pad_token_id=0, eos_token_id=1),
    "t5-base": dict(vocab_size=32128, d_model=768, d_kv=64, d_ff=3072, num_layers=12, num_decoder_layers=12, num_heads=12,
                    decoder_start_token_id=0, pad_token_id=0, eos_token_id=1),
}


class T5(_PostNormBase):
    """modules/heads/generation_head.py:8-30.  The in-repo part -- ``input_proj`` = Linear(d -> d_model) + LayerNorm --
    runs on the HIP kernels, and so does the teacher-forced decoder body (``body='hip'``, pq3d_amd/t5.py: the
    third-party HF ``T5ForConditionalGeneration`` decoder restated on this package's ops, reading the HF module's
    parameters) and greedy generation (labels=None: KV-cache decode step in a HIP graph, t5.GreedyDecoder);
    ``body='hf'`` runs the stock HF model exactly as the reference uses it (encoder bypassed through
    ``encoder_outputs``, cross-attending to the N_q query tokens).
    ``variant`` is loaded with ``from_pretrained`` when it is available locally; without network the same architecture
    is built with random weights (``T5_ARCH`` or an explicit ``hf_config`` dict) -- state_dict keys are identical, so
    a reference checkpoint loads over it."""

    def __init__(self, cfg, variant="t5-small", input_size=768, use_projection=True, hf_config=None, body="hip", **kwargs):
        super().__init__()
        assert body in ("hip", "hf")
        self.body = body   # 'hip': teacher-forced decoder on this package's kernels (pq3d_amd/t5.py); 'hf': stock HF forward
        from transformers import T5Config, T5ForConditionalGeneration   # third-party body (transformers, requirements.txt:63)
        if hf_config is not None:
            self.model = T5ForConditionalGeneration(T5Config(**dict(hf_config)))
        else:
            try:
                self.model = T5ForConditionalGeneration.from_pretrained(variant, local_files_only=True)
            except Exception:  # noqa: BLE001 -- not cached and no network: same architecture, random init
                if variant not in T5_ARCH:
                    raise
                self.model = T5ForConditionalGeneration(T5Config(**T5_ARCH[variant]))
        self.model.config.update(kwargs)
        self._greedy = {}
        hidden_size = self.model.config.d_model
        self.use_projection = use_projection
        if use_projection:
            self.input_proj = nn.Sequential(nn.Linear(input_size, hidden_size), nn.LayerNorm(hidden_size))
        else:
            assert input_size == hidden_size, "input_feat_size should be equal to hidden_size!"

    def forward(self, query_embeds, attention_masks, labels=None):
        from transformers.modeling_outputs import BaseModelOutput
        if self.use_projection:
            query_embeds = linear_ln_forward(self.input_proj, query_embeds, self.ct)
        if labels is not None and self.body == "hip":
            from . import t5
            if self.training:   # standalone use: make sure every call draws from a fresh RNG epoch (a parent model's
                rng = ops.drop_rng(query_embeds.device)   # begin_dropout_step marks this module as managed instead)
                if self._drop_managed != rng.epoch:
                    if rng.epoch == self._drop_epoch:
                        rng.advance()
                    self._drop_epoch = rng.epoch
            return t5.decoder_logits(self.model, query_embeds, attention_masks, labels, self.ct, self.training)
        if labels is None and self.body == "hip":   # greedy generation: KV-cache decode step replayed from a HIP graph
            from . import t5
            B, N, _ = query_embeds.shape
            key = (B, N, t5.new_token_budget(self.model), self.ct, str(query_embeds.device), self.model.shared.weight.data_ptr())
            if key not in self._greedy:
                self._greedy.clear()                # one resident decoder (static caches + graphs) at a time
                self._greedy[key] = t5.GreedyDecoder(self.model, B, N, self.ct, key[2], query_embeds.device)
            return self._greedy[key](query_embeds, attention_masks)
        enc = BaseModelOutput(last_hidden_state=query_embeds)
        if labels is not None:
            return self.model(encoder_outputs=enc, attention_mask=attention_masks, labels=labels).logits
        outputs = self.model.generate(encoder_outputs=enc, attention_mask=attention_masks, do_sample=False)
        return outputs[:, 1:]   # remove the decoder start token (generation_head.py:29)


# ------------------------------------------------------------------------------------------------ input side
class ObjectEncoder(_PostNormBase):
    """modules/vision/object_encoder.py:15-79, projection path (backbone='none')."""

    def __init__(self, cfg, backbone="none", input_feat_size=768, hidden_size=768, freeze_backbone=False,
                 use_projection=False, tgt_cls_num=607, pretrained=None, dropout=0.1, use_cls_head=True):
        super().__init__()
        self.freeze_backbone = freeze_backbone
        if backbone == "pointnet++":       # object_encoder.py:22-28: the frozen point tokenizer on the HIP kernels
            from .pointnetpp import POINTNETPP_TOKENIZER, PointNetPP
            if not freeze_backbone:
                raise NotImplementedError("backbone='pointnet++' is provided as the frozen tokenizer only "
                                          "(freeze_backbone=True, as configs/unified_tasks_sceneverse.yaml:147-148 uses it)")
            self.backbone = PointNetPP(**{k: [list(x) if isinstance(x, list) else x for x in v]
                                          for k, v in POINTNETPP_TOKENIZER.items()})
        elif backbone != "none":
            raise NotImplementedError(f"backbone {backbone!r}")
        if use_cls_head:
            self.cls_head = get_mlp_head(input_feat_size, input_feat_size // 2, tgt_cls_num, dropout=0.3)
        self.dropout_p, self.cls_dropout_p, self._drop_base = float(dropout), 0.3, DROP_BASE_OBJ_ENC
        self.use_projection = use_projection
        if use_projection:
            self.input_feat_proj = nn.Sequential(nn.Linear(input_feat_size, hidden_size), nn.LayerNorm(hidden_size))
        else:
            assert input_feat_size == hidden_size, "input_feat_size should be equal to hidden_size!"
        self.apply(_init_weights_bert)
        if pretrained:                     # object_encoder.py:41-53 (key mapping of the three shipped checkpoints)
            state_dict = {}
            for k, v in torch.load(pretrained, map_location="cpu").items():
                if k[0] in ["0", "2", "4"]:
                    k = "cls_head." + k
                k = k.replace("vision_encoder.vis_cls_head.", "cls_head.").replace("point_cls_head.", "cls_head.")
                state_dict[k.replace("point_feature_extractor.", "backbone.")] = v
            self.load_state_dict(state_dict, strict=False)

    def forward(self, obj_feats, data_dict=None, **kwargs):
        if hasattr(self, "backbone"):      # object_encoder.py:56-69: [B, O, P, 3+C] point clouds -> [B, O, 768], frozen
            self.backbone.compute = self.compute
            for m in self.backbone.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
            B_, O_ = obj_feats.shape[:2]
            with torch.no_grad():
                obj_feats = self.backbone(obj_feats.flatten(0, 1)).view(B_, O_, -1)
        obj_embeds = linear_ln_forward(self.input_feat_proj, obj_feats, self.ct) if self.use_projection else obj_feats
        dev = obj_feats.device
        ctx = self._head_ctx(dev, max(self.dropout_p, self.cls_dropout_p if hasattr(self, "cls_head") else 0.0))
        if self.dropout_p > 0:   # object_encoder.py:72-73
            obj_embeds = ops.dropout(obj_embeds, self._drop(ctx, ops.DROP_ENC_OUT, dev))
        if hasattr(self, "cls_head"):
            return obj_embeds, mlp_head_forward(self.cls_head, obj_feats, self.ct,
                                                drop=self._drop(ctx, ops.DROP_MLP_HEAD, dev, p=self.cls_dropout_p))
        return obj_embeds


class PCDMask3DSegLevelEncoder(_PostNormBase):
    """modules/vision/pcd_mask3d_encoder.py:114-154 WITHOUT its sparse-convolution backbone (Res16UNet34C on
    MinkowskiEngine: SURVEY 2 'out of scope').  Everything after the backbone is here, with the reference's parameter
    names (``feat_proj_list.{i}.{0,1}.{weight,bias}``: a checkpoint's post-backbone weights load; ``backbone.*`` keys are
    not this module's): for every level in ``hlevels + [4]`` the level's voxel features are up-sampled to full resolution
    (``pooltr`` x (4 - hlevel), :127-131), mean-pooled per segment (``scatter_mean(..., dim_size=max_seg)``, :149) and
    projected (Linear + LayerNorm + Dropout, :122-126); the result is the multi-scale LIST the decoder indexes per layer
    (query_encoder.py:90-91) and whose last entry the mask head matches against (query3d_unified.py:163-165).

    forward(pyramid, point2segment, max_seg): ``pyramid[i] = (feats, parents)`` for level ``hlevels[i]`` -- ``feats[b]``
    [N_coarse_b, C_level] the backbone's decomposed features of scene b at that level, ``parents[b]`` [N_b] int64 the
    coarse row of every full-resolution voxel (ops.compose_parents / ops.parents_from_coords; identity for level 4) --
    which is what the backbone's ``aux`` list + coordinate maps provide.  The up-sampled [N, C] intermediate is never
    materialised (ops.upsample_scatter_mean).  Third-party semantics (MinkowskiEngine pooling, torch_scatter): parity
    unpinned by execution, pinned to the oracle's restatement (tests/test_gpu_ops.py)."""

    PLANES = (256, 256, 128, 96, 96)    # Res16UNet34C.PLANES[-5:] (res16unet.py:391)

    def __init__(self, cfg, backbone_kwargs=None, hidden_size=768, hlevels=(0, 1, 2, 3), freeze_backbone=False, dropout=0.1,
                 sizes=None):
        super().__init__()
        self.sizes = tuple(sizes) if sizes is not None else self.PLANES
        self.hlevels = list(hlevels) + [4]      # 4 is for the last level, always used for mask seg features (:118)
        self.feat_proj_list = nn.ModuleList([nn.Sequential(nn.Linear(self.sizes[h], hidden_size), nn.LayerNorm(hidden_size),
                                                           nn.Dropout(dropout)) for h in self.hlevels])
        self.dropout_p, self._drop_base = float(dropout), DROP_BASE_OBJ_ENC + (7 << 12)

    def forward(self, pyramid, point2segment, max_seg, batch_size=None):
        """Two input forms.  LIST form (the reference's decomposed features): ``pyramid[i] = ([feats_b], [parents_b])``,
        ``point2segment = [ids_b]``.  BATCHED form (what a sparse tensor's ``.F`` already is): ``pyramid[i] = (feats [sum N_c, C],
        parents [sum N] -> rows of the concatenated coarse level)``, ``point2segment`` ONE int64 tensor [sum N] with scene b's
        ids offset by ``b * max_seg`` and ``batch_size`` given: one sort for the whole batch, one launch per level."""
        assert len(pyramid) == len(self.hlevels), "one (features, parents) entry per level in hlevels + [4]"
        out = []
        if torch.is_tensor(point2segment):
            assert batch_size is not None, "the batched form needs batch_size (ids are offset by b * max_seg)"
            dev = point2segment.device
            ctx = self._head_ctx(dev)
            plan = ops.SegmentPlan(point2segment, int(batch_size) * int(max_seg))
            for i, ((feats, parents), proj) in enumerate(zip(pyramid, self.feat_proj_list)):
                pooled = ops.upsample_scatter_mean(feats, parents, point2segment, plan.S, plan=plan)
                y = linear_ln_forward(proj, pooled.view(int(batch_size), int(max_seg), -1), self.ct)
                if self.dropout_p > 0:
                    y = ops.dropout(y, self._drop(ctx, ops.DROP_ENC_OUT, dev, m=i))
                out.append(y)
            return out
        dev = point2segment[0].device
        ctx = self._head_ctx(dev)
        # the voxel -> segment ids are sorted once per scene; all 5 levels (and their gradients) reduce over that grouping
        plans = [ops.SegmentPlan(p2s, int(max_seg)) for p2s in point2segment]
        for i, ((feats, parents), proj) in enumerate(zip(pyramid, self.feat_proj_list)):
            pooled = torch.stack([ops.upsample_scatter_mean(f, par, p2s, int(max_seg), plan=pl)
                                  for f, par, p2s, pl in zip(feats, parents, point2segment, plans)])   # [B, max_seg, C_level]
            y = linear_ln_forward(proj, pooled, self.ct)
            if self.dropout_p > 0:
                y = ops.dropout(y, self._drop(ctx, ops.DROP_ENC_OUT, dev, m=i))
            out.append(y)
        return out


class PositionEmbeddingCoordsSine(nn.Module):
    """position_embedding.py:46-179, pos_type='fourier', normalize=True (buffer gauss_B [3, d_pos/2])."""

    def __init__(self, d_pos, d_in=3, gauss_scale=1.0):
        super().__init__()
        assert d_pos % 2 == 0
        self.register_buffer("gauss_B", torch.empty(d_in, d_pos // 2).normal_() * gauss_scale)

    def forward(self, xyz, input_range):
        return ops.fourier(xyz, input_range[0], input_range[1], self.gauss_B)  # [B,N,d] (already permuted)


class CoordinateEncoder(_PostNormBase):
    """model/query3d_unified.py:15-27."""

    def __init__(self, hidden_size, use_projection=True):
        super().__init__()
        self.pos_enc = PositionEmbeddingCoordsSine(d_pos=hidden_size)
        if use_projection:
            self.feat_proj = nn.Sequential(nn.Linear(hidden_size, hidden_size), nn.LayerNorm(hidden_size))

    def forward(self, coords, input_range):
        pos = self.pos_enc(coords, input_range)
        if hasattr(self, "feat_proj"):
            pos = linear_ln_forward(self.feat_proj, pos, self.ct)
        return pos


def calc_pairwise_locs(obj_centers, obj_whls=None, eps=1e-10, pairwise_rel_type="center", spatial_dist_norm=True,
                       spatial_dim=5):
    """modules/utils.py:38-87 for the configuration on the path."""
    if pairwise_rel_type != "center" or not spatial_dist_norm or spatial_dim != 5:
        raise NotImplementedError("only pairwise_rel_type='center', spatial_dist_norm=True, spatial_dim=5")
    return ops.pairwise_locs(obj_centers.float(), eps)
