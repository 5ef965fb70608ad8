#!/usr/bin/env python3
"""CPU analysis (test infrastructure; imports the oracle): what a split-bf16 ('bf16x3') FORWARD on the key/value side buys the
parameter gradients when the BACKWARD products stay single-bf16 -- the decision behind round 6's compute mode 'bf16x3'.

The fp32 oracle runs with (i) optional bf16 rounding at the five forward sites of every cross-attention (K, V, Q, P, O: the 'bf16'
mode), and (ii) an emulation of the HIP backward: every Linear's input / weight gradient from bf16-rounded operands, the
cross-attention core's backward as attn_bwd_resident forms it (P recomputed from bf16 Q, K and the saved log-sum-exp,
dP = dO V^T on bf16, delta = rowsum(dO * O), dS / P rounded to bf16 in front of the dQ / dK / dV products, dK / dV stored in bf16).
Printed: forward error of the final queries and the worst / median per-parameter gradient error against the plain fp32 oracle.
usage: python tools/probes/x3_grad_emul.py [B] [Ns]"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pq3d_oracle as O  # noqa: E402
from tests import util  # noqa: E402


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


class LinB(torch.autograd.Function):
    """fp32 forward (optionally bf16-rounded operands), backward products on bf16-rounded operands."""
    @staticmethod
    def forward(ctx, x, w, b, fwd_round, bwd_round, out_round):
        ctx.save_for_backward(x, w)
        ctx.has_b, ctx.bwd_round, ctx.out_round = b is not None, bwd_round, out_round
        xx, ww = (bf(x), bf(w)) if fwd_round else (x, w)
        y = xx @ ww.t()
        return y + b if b is not None else y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        r = bf if ctx.bwd_round else (lambda t: t)
        if ctx.out_round:   # the gradient arrives as a stored bf16 tensor (dK / dV)
            dy = bf(dy)
        dx = r(dy) @ r(w)
        dw = r(dy).reshape(-1, dy.shape[-1]).t() @ r(x).reshape(-1, x.shape[-1])
        db = dy.reshape(-1, dy.shape[-1]).sum(0) if ctx.has_b else None
        return dx, dw, db, None, None, None


class AttnCore(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, kpm, H, cfg):
        B, Lq, d = q.shape
        dh = d // H
        sp = lambda t: t.view(B, -1, H, dh).permute(0, 2, 1, 3)
        qh, kh, vh = sp(q), sp(k), sp(v)
        fs = cfg["fwd_sites"]
        if "Q" in fs: qh = bf(qh)
        if "K" in fs: kh = bf(kh)
        if "V" in fs: vh = bf(vh)
        sc = 1.0 / math.sqrt(dh)
        s = (qh @ kh.transpose(-1, -2)) * sc
        if kpm is not None:
            s = s.masked_fill(kpm[:, None, None, :], float("-inf"))
        # add_zero_attn: one more key with logit 0 and value 0
        m = torch.clamp(s.max(-1, keepdim=True).values, min=0.0)
        e = torch.exp(s - m)
        l = e.sum(-1, keepdim=True) + torch.exp(-m)
        p = e / l
        lse = m + torch.log(l)
        pp = bf(p) if "P" in fs else p
        o = pp @ vh
        if "O" in fs: o = bf(o)
        ctx.save_for_backward(qh, kh, vh, o, lse)
        ctx.kpm, ctx.cfg, ctx.sc = kpm, cfg, sc
        return o.permute(0, 2, 1, 3).reshape(B, Lq, d)

    @staticmethod
    def backward(ctx, do):
        qh, kh, vh, o, lse = ctx.saved_tensors
        cfg, sc = ctx.cfg, ctx.sc
        B, H, Lq, dh = qh.shape
        doh = do.view(B, Lq, H, dh).permute(0, 2, 1, 3)
        mode = cfg["bwd"]
        if mode == "exact":
            r = lambda t: t
        else:
            r = bf
        doh_r = r(doh)                      # dO is produced in bf16 by the out-projection's input-gradient launch
        q_r, k_r, v_r = r(qh), r(kh), r(vh)
        if mode == "bf16_s3":               # scores recomputed at split-bf16 grade, everything else bf16
            s = (qh @ kh.transpose(-1, -2)) * sc
        else:
            s = (q_r @ k_r.transpose(-1, -2)) * sc
        if ctx.kpm is not None:
            s = s.masked_fill(ctx.kpm[:, None, None, :], float("-inf"))
        p = torch.exp(s - lse)
        dp = doh_r @ v_r.transpose(-1, -2)
        delta = (doh_r * (r(o) if cfg.get("o_bf16", False) else o)).sum(-1, keepdim=True)
        ds = p * (dp - delta) * sc
        p_r, ds_r = r(p), r(ds)
        dv = p_r.transpose(-1, -2) @ doh_r
        dk = ds_r.transpose(-1, -2) @ q_r
        dq = ds_r @ k_r
        mg = lambda t: t.permute(0, 2, 1, 3).reshape(B, -1, H * dh)
        if mode != "exact":
            dq = bf(dq)
            if not cfg.get("dkv_f32", False):
                dk, dv = bf(dk), bf(dv)
        return mg(dq), mg(dk), mg(dv), None, None, None


def run(args, sd, dd, cfg):
    orig_lin, orig_mha = O.linear, O.mha

    def lin(x, w, b=None):
        return LinB.apply(x, w, b, False, cfg["lin_bwd"], False)

    def mha(sd_, p, query, key, value, H, key_padding_mask=None, attn_mask=None, add_zero_attn=False, drop_tag=None, drop_m=0):
        d = query.shape[-1]
        w, b = sd_[p + "in_proj_weight"], sd_[p + "in_proj_bias"]
        kvr = cfg["kv_lin_bwd"]
        q = LinB.apply(query, w[:d], b[:d], False, cfg["lin_bwd"], False)
        k = LinB.apply(key, w[d:2 * d], b[d:2 * d], "K" in cfg["fwd_sites"], kvr, kvr and not cfg.get("dkv_f32", False))
        v = LinB.apply(value, w[2 * d:], b[2 * d:], "V" in cfg["fwd_sites"], kvr, kvr and not cfg.get("dkv_f32", False))
        o = AttnCore.apply(q, k, v, key_padding_mask, H, cfg)
        return LinB.apply(o, sd_[p + "out_proj.weight"], sd_[p + "out_proj.bias"], "O" in cfg["fwd_sites"], cfg["lin_bwd"], False)

    O.linear, O.mha = lin, mha
    try:
        return util.run_oracle(args, sd, dd)
    finally:
        O.linear, O.mha = orig_lin, orig_mha


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    Ns = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    args = dict(B=B, Ns=Ns, Nq=100, d=256, H=8, L=4, memories=["voxel", "mv", "pc"], heads=["ground"], spatial=True,
                structure="parallel", seed=0, data_seed=1234)
    _cfg, _model, sd, dd = util.model_case(args)
    oout, collect, oloss, og = util.run_oracle(args, sd, dd)
    qref = collect[-1]
    scale = float(qref.abs().max())
    names = sorted(n for n in og if "pairwise_loc_fc" not in n)
    gmax = max(float(og[n].norm()) for n in names)
    ALL = {"K", "V", "Q", "P", "O"}
    cases = [
        ("bf16 mode: 5 fwd sites + bf16 bwd", dict(fwd_sites=ALL, bwd="bf16", lin_bwd=True, kv_lin_bwd=True, o_bf16=True)),
        ("x3 fwd + bf16 bwd everywhere", dict(fwd_sites=set(), bwd="bf16", lin_bwd=True, kv_lin_bwd=True)),
        ("x3 fwd + bf16 bwd, scores x3 in bwd", dict(fwd_sites=set(), bwd="bf16_s3", lin_bwd=True, kv_lin_bwd=True)),
        ("x3 fwd + bf16 bwd, dK/dV fp32", dict(fwd_sites=set(), bwd="bf16", lin_bwd=True, kv_lin_bwd=True, dkv_f32=True)),
        ("x3 fwd + exact attn/KV bwd, bf16 query-side bwd", dict(fwd_sites=set(), bwd="exact", lin_bwd=True, kv_lin_bwd=False)),
        ("x3 fwd + bf16 attn/KV bwd, exact query-side bwd", dict(fwd_sites=set(), bwd="bf16", lin_bwd=False, kv_lin_bwd=True)),
    ]
    print(f"config 2 shapes, {B} scenes x {Ns} segments; gradient error = |g - g_fp32| / max(|g_fp32|, f * gmax) per parameter")
    for name, cfg in cases:
        out, col, loss, g = run(args, sd, dd, cfg)
        qe = float((col[-1] - qref).abs().max()) / scale
        e2 = sorted(((float((g[n] - og[n]).norm() / max(float(og[n].norm()), 1e-2 * gmax)), n) for n in names), reverse=True)
        e3 = sorted(((float((g[n] - og[n]).norm() / max(float(og[n].norm()), 1e-3 * gmax)), n) for n in names), reverse=True)
        med = e2[len(e2) // 2][0]
        print(f"  {name:52s} query {qe:.2e}  worst(f=1e-2) {e2[0][0]:.2e} {e2[0][1]}  worst(f=1e-3) {e3[0][0]:.2e}  median {med:.2e}")
        print("       top5 (f=1e-2): " + ", ".join(f"{n.split('unified_encoder.')[-1]}={v:.1e}" for v, n in e2[:5]))


if __name__ == "__main__":
    main()
