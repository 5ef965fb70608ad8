#!/usr/bin/env python3
"""CPU report (test infrastructure; imports the oracle): which single-bf16 rounding site of the cross-attention chain costs what,
end to end, at BASELINE config 2 shapes -- the measurement behind the 'bf16_kv32' decision.

The 'bf16' compute mode rounds five tensors of every cross-attention to bf16: the stored K and V, Q, the
probabilities P in front of P V, and the attention output O in front of the output projection (whose weights are rounded
too).  Everything else of the decoder is fp32-grade (split-bf16).  This script runs the fp32 oracle with bf16 rounding injected
at a chosen SUBSET of those sites and prints max |query - fp32 query| / max |fp32 query| after the 4 layers:
    all five       = the bf16 mode's arithmetic (the HIP kernels measure 6.7e-3 at full config 2),
    all but K, V   = a 'bf16_kv32' variant (K / V kept in fp32: 2x the K/V bytes and an fp32-operand or split-bf16 score product),
    single sites   = each site's own share.
usage: python tests/rounding_sites_report.py [B] [Ns]      (default 2 scenes x 1024 segments: ~1 min on 8 cores)"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pq3d_oracle as O  # noqa: E402  (checker-side analysis only)
from pq3d_amd import synth  # noqa: E402
from pq3d_amd.model import Query3DUnified, make_cfg  # noqa: E402

SITES = ("K", "V", "Q", "P", "O")


def bf(x):
    return x.to(torch.bfloat16).to(x.dtype)


def make_mha(sites):
    def mha(sd, p, query, key, value, H, key_padding_mask=None, attn_mask=None, add_zero_attn=False, drop_tag=None, drop_m=0):
        d = query.shape[-1]
        dh = d // H
        w, b = sd[p + "in_proj_weight"], sd[p + "in_proj_bias"]
        q = O._split_heads(O.linear(query, w[:d], b[:d]), H)
        k = O._split_heads(O.linear(key, w[d:2 * d], b[d:2 * d]), H)
        v = O._split_heads(O.linear(value, w[2 * d:], b[2 * d:]), H)
        if "Q" in sites: q = bf(q)
        if "K" in sites: k = bf(k)
        if "V" in sites: v = bf(v)
        B, _, Lq, _ = q.shape
        Lk = k.shape[2]
        s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(dh))
        if attn_mask is not None:
            s = s.masked_fill(attn_mask.view(B, -1, Lq, Lk), O.NEG_INF)
        if key_padding_mask is not None:
            s = s.masked_fill(key_padding_mask[:, None, None, :], O.NEG_INF)
        if add_zero_attn:
            s = torch.cat([s, s.new_zeros(B, H, Lq, 1)], dim=-1)
            v = torch.cat([v, v.new_zeros(B, H, 1, dh)], dim=2)
        a = torch.softmax(s, dim=-1)
        if "P" in sites: a = bf(a)
        o = O._merge_heads(a @ v)
        if "O" in sites: o = bf(o)
        return O.linear(o, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])
    return mha


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    Ns = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    mem = ["voxel", "mv", "pc"]
    kw = dict(d=256, H=8, L=4, memories=mem, heads=[], spatial=True, structure="parallel", use_self_mask=False)
    model = Query3DUnified(make_cfg(**kw), compute="fp32")
    sd = synth.fill_module(model, 0)
    dd = synth.synth_data_dict(B, Ns, 100, {m: 256 for m in mem}, seed=1234)
    ocfg = dict(memories=mem, heads=[], hidden_size=256, num_heads=8, num_layers=4, structure="parallel", spatial_selfattn=True,
                use_self_mask=False, filter_out_classes=[0, 2])
    orig = O.mha
    with torch.no_grad():
        ref = O.query3d_unified_forward(sd, ocfg, dict(dd))["query"]
        scale = float(ref.abs().max())
        rows = [("all five (= 'bf16' mode arithmetic)", set(SITES)), ("all but K, V (= bf16_kv32)", {"Q", "P", "O"}),
                ("K, V only", {"K", "V"})] + [(f"{s_} only", {s_}) for s_ in SITES]
        print(f"config 2 shapes, {B} scenes x {Ns} segments, 4 layers; error = max|dq| / max|q_fp32| (max|q| = {scale:.3f})")
        for name, sites in rows:
            O.mha = make_mha(sites)
            try:
                out = O.query3d_unified_forward(sd, ocfg, dict(dd))["query"]
            finally:
                O.mha = orig
            print(f"  {name:40s} {float((out - ref).abs().max()) / scale:.2e}")


if __name__ == "__main__":
    main()
