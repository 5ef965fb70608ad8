#!/usr/bin/env python3
"""bench.py -- decoder forward+backward scenes/sec on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one forward+backward of Query3DUnified (encoders + 4-layer promptable query decoder) on one batch
of synthetic scenes (BASELINE config 2: B=8 scenes/GPU, N_seg=1024, N_q=100, d=256, H=8, L=4, three memories,
parallel cross-attention, spatial self-attention, 2-D padding masks, loss = mean(query)), bf16 MFMA operands with
fp32 accumulation, inputs resident in HBM, followed by the data-parallel gradient exchange (flat-buffer RCCL
all-reduce, N > 1).  The step is captured once into a HIP graph and replayed.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from pq3d_amd import ops, synth  # noqa: E402
from pq3d_amd.model import Query3DUnified, make_cfg  # noqa: E402
from pq3d_amd.parallel import FlatGradAllReducer  # noqa: E402
from pq3d_amd.profiler import KernelTimer  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0

CONFIGS = {
    # name: (B, N_seg, N_q, d, H, L, memories, heads, use_self_mask)
    "c2": dict(B=8, Ns=1024, Nq=100, d=256, H=8, L=4, memories=["voxel", "mv", "pc"], heads=[], use_self_mask=False),
    "c4": dict(B=4, Ns=4096, Nq=200, d=256, H=8, L=4, memories=["voxel", "mv", "pc"], heads=["mask"],
               use_self_mask=True),
    # c5 (BASELINE config 5, per GPU): 6 layers + caption head (T5-small decoder, random init, teacher-forced T_r = 32), all on
    # the HIP kernels; the caption body alone is also timed separately ("t5_body")
    "c5": dict(B=16, Ns=2048, Nq=100, d=256, H=8, L=6, memories=["voxel", "mv", "pc"], heads=["generation"],
               use_self_mask=False, Tr=32),
    # c5p: c5 in the reference's stage-2 SHIPPED decoder configuration (configs/unified_tasks_sceneverse.yaml:113,159-165):
    # memories [mv, pc, voxel, prompt], structure 'mixed' (parallel scene memories, then a sequential prompt
    # cross-attention over T = 32 pre-encoded prompt tokens), training-time memory dropout 0.6 -- on the fused executor
    "c5p": dict(B=16, Ns=2048, Nq=100, d=256, H=8, L=6, memories=["mv", "pc", "voxel", "prompt"], heads=["generation"],
                use_self_mask=False, Tr=32, structure="mixed", T=32, memory_dropout=0.6),
    "c1": dict(B=2, Ns=128, Nq=16, d=64, H=4, L=1, memories=["voxel"], heads=[], use_self_mask=False, spatial=False,
               structure="sequential"),
    # s1 / s2: the decoder sizes the reference SHIPS, on synthetic inputs of the shipped shapes.
    # s1 = stage 1, configs/instseg_sceneverse.yaml:57,95,140-146: per-GPU batch 4, hidden 768, 12 heads (d_h = 64), 4 layers
    #      re-traversed by num_blocks = 3 (12 layer applications, 13 mask-head calls over the 3 memories, 201 targets), 3-D
    #      self-masks, 120 queries (:44); 2048 segments per scene (the segment count is data-dependent; ScanNet scenes have
    #      1-3 k); ObjectEncoder inputs 768-d
    "s1": dict(B=4, Ns=2048, Nq=120, d=768, H=12, L=4, num_blocks=3, memories=["voxel", "mv", "pc"], heads=["mask"],
               use_self_mask=True),
    # s2 = stage 2, configs/unified_tasks_sceneverse.yaml:62,85,113,120,159-171: per-GPU batch 128 scenes of <= 80 objects
    #      (queries AND memory rows are the objects), memories [mv, pc, voxel, prompt], structure 'mixed' with T = 32 prompt
    #      tokens, memory dropout 0.6, 6-D locations, offline voxel features 128-d, ground head (hidden 384)
    "s2": dict(B=128, Ns=80, Nq=80, d=768, H=12, L=4, memories=["mv", "pc", "voxel", "prompt"], heads=["ground"],
               use_self_mask=False, structure="mixed", T=32, memory_dropout=0.6, dim_loc=6, ground_hidden=384,
               d_in={"mv": 768, "pc": 768, "voxel": 128}),
}


def step_flops(c) -> float:
    """SURVEY §8d closed form (multiply-add = 2 FLOP; STEP = 3 x FWD)."""
    B, Ns, Nq, d, H, L = c["B"], c["Ns"], c["Nq"], c["d"], c["H"], c["L"]
    nb = c.get("num_blocks", 1)
    mems = [m for m in c["memories"] if m != "prompt"]
    M = len(mems)
    F, C = 2048, 201
    ca = M * (2 * B * d * d * (2 * Nq + 2 * Ns) + 4 * B * Nq * (Ns + 1) * d)
    if "prompt" in c["memories"]:   # the sequential prompt cross-attention of structure 'mixed' (T prompt tokens)
        ca += 2 * B * d * d * (2 * Nq + 2 * c["T"]) + 4 * B * Nq * (c["T"] + 1) * d
    sa = 8 * B * Nq * d * d + 4 * B * Nq * Nq * d + (2 * B * Nq * Nq * 5 * H if c.get("spatial", True) else 0)
    ffn = 4 * B * Nq * d * F
    mh = 2 * B * Nq * (d * d + d * C) + M * (2 * B * d * d * (Nq + Ns) + 2 * B * Ns * Nq * d)
    d_in = c.get("d_in", {})
    enc = sum(2 * B * Ns * d_in.get(m, d) * d for m in mems) + 2 * B * (Nq + Ns) * d * d + 2 * B * (Nq + Ns) * 3 * (d // 2)
    calls = (L * nb + 1) if "mask" in c["heads"] else 0
    return 3.0 * (L * nb * (ca + sa + ffn) + calls * mh + enc)


def build(c, compute, device, seed):
    d_in = {m: c.get("d_in", {}).get(m, c["d"]) for m in c["memories"]}
    cfg = make_cfg(d=c["d"], H=c["H"], L=c["L"], memories=c["memories"], heads=c["heads"], d_in=d_in,
                   spatial=c.get("spatial", True), structure=c.get("structure", "parallel"),
                   use_self_mask=c["use_self_mask"], C=201, foc=[0, 2], memory_dropout=c.get("memory_dropout", 0.0),
                   num_blocks=c.get("num_blocks", 1), dim_loc=c.get("dim_loc", 3), ground_hidden=c.get("ground_hidden"))
    model = Query3DUnified(cfg, compute=compute)
    sd = synth.fill_module(model, 0)
    model.to(device)
    dd = synth.synth_data_dict(c["B"], c["Ns"], c["Nq"], d_in, seed=seed, memories=c["memories"], prompt_len=c.get("T", 0),
                               d_model=c["d"], loc_dim=c.get("dim_loc", 3))
    if "generation" in c["heads"]:
        g = torch.Generator().manual_seed(seed)
        dd["response"] = torch.randint(2, 32000, (c["B"], c["Tr"]), generator=g)
    return model, sd, dd


def loss_fn(out, heads):
    q = out["query_embeds"] if "query_embeds" in out else out["query"]
    if q.is_cuda and "mask" in heads:
        # mean(query) + sum over prediction layers of mean(clamp(mask_logits, -50)) + mean(class logits of the kept
        # classes): one launch forward, one backward (ops.mean_many) instead of ~14 framework launches per layer
        xs = [q] + list(out["predictions_mask"]) + list(out["predictions_class"])
        modes = ["plain"] + ["clamp_min"] * len(out["predictions_mask"]) + ["finite"] * len(out["predictions_class"])
        loss = ops.mean_many(xs, modes, clamp_min=-50.0)
    else:
        loss = ops.mean_all(q) if q.is_cuda else q.mean()   # the CPU-baseline leg has host tensors
        if "mask" in heads:
            for cl, m in zip(out["predictions_class"], out["predictions_mask"]):
                loss = loss + m.clamp(min=-50.0).mean() + torch.where(torch.isfinite(cl), cl, torch.zeros_like(cl)).mean()
    if "generation" in heads:   # generation_loss: token cross-entropy of the teacher-forced logits
        lg = out["generation_logits"]
        if lg.is_cuda:   # the 32128-way token cross-entropy on the path's own kernels (one workgroup per row)
            from pq3d_amd.losses import cross_entropy_rows
            loss = cross_entropy_rows(lg, out["generation_label"], add=loss)
        else:
            loss = loss + torch.nn.functional.cross_entropy(lg.flatten(0, 1).float(), out["generation_label"].flatten())
    return loss


def cpu_baseline(c, sd, dd, steps, warmup, hf_body=None):
    """The oracle (CPU restatement of the reference path, pinned to the reference's outputs by the golden
    fixtures) timed on this host's cores: forward+backward, fp32, dropout 0 -- same workload, bounded sample.
    Caption head (config 5): the oracle restates the in-repo part (input_proj); the T5 body is the stock HF model on the
    CPU, exactly what the reference's head calls (generation_head.py: encoder_outputs = projected queries)."""
    from oracle import pq3d_oracle as O  # CPU baseline leg only
    ocfg = dict(memories=c["memories"], heads=c["heads"], hidden_size=c["d"], num_heads=c["H"], num_layers=c["L"],
                structure=c.get("structure", "parallel"), spatial_selfattn=c.get("spatial", True),
                use_self_mask=c["use_self_mask"], filter_out_classes=[0, 2], num_blocks=c.get("num_blocks", 1),
                dim_loc=c.get("dim_loc", 3))
    sdo = {k: v.clone().requires_grad_(v.dtype.is_floating_point and not k.endswith("gauss_B") and
                                       not k.startswith("generation_head.model.")) for k, v in sd.items()}

    def one_step():
        for v in sdo.values():
            v.grad = None
        if hf_body is not None:
            hf_body.zero_grad(set_to_none=True)
        t0 = time.perf_counter()
        out = O.query3d_unified_forward(sdo, ocfg, dict(dd))
        loss = loss_fn(out, [h for h in c["heads"] if h != "generation"])
        if hf_body is not None:
            from transformers.modeling_outputs import BaseModelOutput
            res = hf_body(encoder_outputs=BaseModelOutput(last_hidden_state=out["generation_input"]),
                          attention_mask=dd["query_pad_masks"].long(), labels=dd["response"])
            loss = loss + res.loss
        loss.backward()
        return time.perf_counter() - t0

    # torch's CPU backend is not fastest with every hardware thread on a many-core host (tiny ops; at 256 threads one
    # step took 247 s on a 256-CPU box vs 0.25 s at 16): probe thread counts in ASCENDING order (1 warm-up + 2 timed
    # steps each), stop as soon as more threads stop helping, never exceed 64, and keep the whole probe under ~60 s
    ncpu = os.cpu_count() or 1
    cands = sorted({n for n in (4, 8, 16, 32, 64) if n <= ncpu} or {ncpu})
    probe = {}
    t_probe = time.perf_counter()
    for n in cands:
        torch.set_num_threads(n)
        one_step()
        probe[n] = min(one_step(), one_step())
        if probe[n] > 1.25 * min(probe.values()) or time.perf_counter() - t_probe > 60.0:
            break
    best = min(probe, key=probe.get)
    torch.set_num_threads(best)
    steps = max(4, min(steps, int(25.0 / probe[best])))   # bounded sample: ~25 s of CPU work at the big configurations
    for _ in range(warmup):
        one_step()
    times = sorted(one_step() for _ in range(steps))
    med = times[len(times) // 2]
    return {"value": c["B"] / med, "unit": "scenes/s", "cores": best, "kind": "port", "ms_per_step": med * 1e3,
            "host_cpus": ncpu, "threads_tried_ms": {str(k): round(v * 1e3, 1) for k, v in probe.items()},
            "sample": f"{steps} timed fwd+bwd steps (median) of the same {c['B']}-scene batch after {warmup} warm-up, "
                      f"fp32, torch CPU ops, dropout 0, best of the thread counts tried"}


# The headline mode is the one that meets north_star's tolerance ("within 1e-3 bf16") END TO END: 'bf16x3' = bf16 MFMA operands
# everywhere, hi + lo bf16 pairs where single bf16 would not hold 1e-3.  'bf16' (single-bf16 key/value side: ~12 %
# faster at config 2, 3e-3 .. 7e-3 end to end) and 'fp32' are timed beside it in `parity_modes`.
DEFAULT_COMPUTE = "bf16x3"
FAMILY_KERNELS = {   # GPU kernels behind a C-ABI entry point (rocprofv3 / PMC kernel names, template arguments stripped)
    "pq3d_gemm": ["gemm_wk_kernel", "gemm_fast_kernel", "gemm_slow_kernel", "gemm_nt128_kernel", "gemm_tt128_kernel", "gemm_wktt_kernel",
                  "gemm_cv128_kernel"],
    "pq3d_gemm_tt_multi": ["gemm_tt_multi_kernel"],
    "pq3d_attn_fwd": ["attn_fwd_resident_kernel", "attn_fwd_x3_kernel", "attn_sa_fwd_kernel", "attn_small_fwd_kernel", "attn_fwd_kernel",
                      "attn_fwd_combine_kernel"],
    "pq3d_attn_bwd": ["attn_bwd_resident_kernel", "attn_sa_bwd_kernel", "attn_small_bwd_kernel", "attn_bwd_dq_kernel", "attn_bwd_dkv_kernel",
                      "attn_dq_combine_kernel"],
    "pq3d_add_ln_fwd": ["add_ln_fwd_kernel"], "pq3d_add_ln_bwd": ["add_ln_bwd_kernel"],
    "pq3d_chain_ffn_fwd": ["chain_ffn_fwd_kernel"], "pq3d_chain_ca_fwd": ["chain_ca_fwd_kernel"],
    "pq3d_chain_ffn_bwd": ["chain_ffn_bwd_kernel"], "pq3d_chain_sa_bwd": ["chain_sa_bwd_kernel"],
    "pq3d_chain_mh_fwd": ["chain_mh_fwd_kernel"], "pq3d_chain_mh_bwd": ["chain_mh_bwd_kernel"],
}


def kernel_for(entry, key):
    """The GPU kernel a timed C-ABI call (entry point, shape key of pq3d_amd/ops.py's timers) launches -- the rules of the
    dispatchers in csrc/ (gemm.hip pq3d_gemm, attention.hip launch_fwd / launch_bwd) restated for the shapes of the BASELINE
    configurations.  Used to attach the ALGORITHMIC bytes / FLOPs of the eager profiled pass to the kernels of the in-graph trace;
    every use is verified against the trace's launch counts (`mapping_verified`), so a wrong rule cannot silently price a kernel."""
    import re
    fam = FAMILY_KERNELS.get(entry, [])
    if len(fam) == 1:
        return fam[0]
    g = lambda pat, d=0: (lambda m: int(m.group(1)) if m else d)(re.search(pat, key))
    if entry in ("pq3d_attn_fwd", "pq3d_attn_bwd"):
        fwd = entry.endswith("fwd")
        Lq, Lk, dh, ct = g(r"Lq(\d+)"), g(r"Lk(\d+)"), g(r"dh(\d+)"), g(r"ct(\d+)")
        if ct == 2:   # split-bf16: self-attention kernels (<= 240 tokens) / the key/value-plane forward
            if Lq <= 240 and Lk <= 240:
                return "attn_sa_fwd_kernel" if fwd else "attn_sa_bwd_kernel"
            return "attn_fwd_x3_kernel" if fwd else None
        if ct == 1:
            if fwd:
                return "attn_fwd_resident_kernel" if (Lq <= 128 <= Lk and dh == 32 and "m3" not in key) else "attn_fwd_kernel"
            return "attn_bwd_resident_kernel" if (Lq <= 256 and Lk >= 128 and dh in (32, 64)) else "attn_bwd_dkv_kernel"
        return "attn_small_fwd_kernel" if fwd else "attn_small_bwd_kernel"
    if entry == "pq3d_gemm":
        M, N, K, ct, sk = g(r"M(\d+)"), g(r"N(\d+)"), g(r"K(\d+)g"), g(r"ct(\d+)"), g(r"s(\d+)ct", 1)
        tt = "TT" in key
        if tt:
            return "gemm_tt128_kernel" if (ct == 1 and K >= 2048 and M % 128 == 0 and N % 128 == 0) else "gemm_wktt_kernel"
        if M <= 2048:
            return "gemm_wk_kernel"
        if ct == 1 and N % 128 == 0 and K % 64 == 0 and sk == 1:
            return "gemm_nt128_kernel"
        return "gemm_fast_kernel"
    return None


def step_bytes(c):
    """SURVEY 8(d) 'Algorithmic bytes (compulsory HBM traffic, forward, ideal fusion, e_b = 2)': per layer weights (4 M d^2 + 4 d^2 + 2 d F)
    e_b + memories M 2 B N_s d e_b + query state (M + 2) 2 B N_q d e_b; per mask-head call (2 M d^2 + d^2 + d C) e_b + M B N_s d e_b
    + 4 B N_s N_q + 4 B N_q C.  STEP = 3 x forward (backward = 2 x forward, the convention of the FLOP count)."""
    B, Ns, Nq, d, L_, F_ = c["B"], c["Ns"], c["Nq"], c["d"], c["L"], 2048
    M = len([m for m in c["memories"] if m != "prompt"])
    eb = 2
    layer = (4 * M * d * d + 4 * d * d + 2 * d * F_) * eb + M * 2 * B * Ns * d * eb + (M + 2) * 2 * B * Nq * d * eb
    fwd = L_ * c.get("num_blocks", 1) * layer
    if "mask" in c["heads"]:
        C_ = c.get("C", 201)
        fwd += (L_ * c.get("num_blocks", 1) + 1) * ((2 * M * d * d + d * d + d * C_) * eb + M * B * Ns * d * eb + 4 * B * Ns * Nq + 4 * B * Nq * C_)
    return 3.0 * fwd


def live_kernel_trace(args, timeout=300):
    """In-graph kernel durations of THIS build on THIS box, measured inside this run: a child `rocprofv3 --kernel-trace -- python
    bench.py --headline-only` of the same config / compute mode (20 replayed steps), its rocpd database read here.  Returns
    {kernel name (template arguments stripped): (launches per step, us per step)}, meta -- or ({}, reason) when rocprofv3 is not
    there / the child failed (the roofline then falls back to this run's own HIP-event times and says so)."""
    import glob
    import re
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    if os.environ.get("PQ3D_BENCH_CHILD") or args.no_live_trace:
        return {}, "disabled"
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {}, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="pq3d_trace_", dir="/tmp")
    cmd = [exe, "--kernel-trace", "-d", tmp, "-o", "t", "--", sys.executable, os.path.abspath(__file__), "--config", args.config,
           "--compute", args.compute, "--dropout", args.dropout, "--steps", "20", "--warmup", "5", "--cpu-steps", "0", "--profile-steps", "1",
           "--headline-only", "--min-time", "0", "--no-live-trace"] + (["--no-graph"] if args.no_graph else [])
    try:
        t0 = time.perf_counter()
        subprocess.run(cmd, env=dict(os.environ, PQ3D_BENCH_CHILD="1", TMPDIR="/tmp"), cwd="/tmp", stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, timeout=timeout, check=False)
        dbs = glob.glob(os.path.join(tmp, "**", "*.db"), recursive=True)
        if not dbs:
            return {}, "rocprofv3 child left no database"
        cur = sqlite3.connect(dbs[0]).cursor()
        rows = cur.execute("select name, count(*), sum(end-start) from kernels group by name").fetchall()
        short = lambda n: re.sub(r"[<(].*$", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))
        import collections
        marks = [r[1] for r in rows if any(m in r[0] for m in ("fourier_pair_kernel", "pairwise_locs_kernel", "mask_not_kernel"))]
        cnt = collections.Counter(marks or [r[1] for r in rows if r[1] >= 3])
        steps = float(cnt.most_common(1)[0][0]) if cnt else 0.0
        if steps <= 0:
            return {}, "no once-per-step kernel in the trace"
        out = {}
        for n, calls, tot in rows:
            k = short(n)
            a = out.get(k, (0.0, 0.0))
            out[k] = (a[0] + calls / steps, a[1] + tot / 1e3 / steps)
        return out, {"source": "live: rocprofv3 --kernel-trace child of this run (same build, same box; 20 replayed + 9 warm-up / eager steps)",
                     "steps": steps, "kernel_us_per_step": sum(v[1] for v in out.values()), "dispatches_per_step": sum(v[0] for v in out.values()),
                     "child_wall_s": round(time.perf_counter() - t0, 1)}
    except Exception as e:  # noqa: BLE001
        return {}, f"{type(e).__name__}: {e}"[:200]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pmc_per_launch(pmc, name):
    """Calibrated memory-side bytes per launch of one kernel from a PMC traffic file (FETCH_SIZE / WRITE_SIZE passes), or None."""
    rows = (pmc or {}).get("kernels", {}).get(name, [])
    launches = sum(r["launches"] for r in rows)
    if not launches:
        return None, None
    cal = pmc.get("calibration") or {}
    cf = 1.0 / cal["fetch_raw_over_true"] if cal.get("fetch_raw_over_true") else 2.0    # guide: x2 for wide reads
    cw = 1.0 / cal["write_raw_over_true"] if cal.get("write_raw_over_true") else 1.0
    fetch = sum(r["fetch_kib"] * 1024 * r["launches"] for r in rows)
    write = sum(r["write_kib"] * 1024 * r["launches"] for r in rows)
    return (fetch * cf + write * cw) / launches, {"fetch_x": round(cf, 3), "write_x": round(cw, 3),
                                                  "from": "calibration launch in the same PMC pass" if cal else
                                                  "MI355X_MICROARCH.md HBM section (no calibration in this file)"}


def kernel_block(name, trace, summ, ps, peak_tflops, pmc, mfma):
    """Roofline entry of ONE GPU kernel: time = its launches in the replayed graph (live trace of this run), work = the ALGORITHMIC
    bytes / FLOPs of the C-ABI calls that launch it (eager profiled pass of this run, pq3d_amd/ops.py's counts), both fractions
    reported, `bound` = the larger; `traffic` = calibrated PMC bytes per launch from the committed counter passes of the
    same build (null when profiles/ holds none for this build)."""
    l_graph, us_step = trace[name]
    recs = [(n, k, v) for (n, k), v in summ.items() if kernel_for(n, k) == name]
    calls = sum(v["calls"] for _n, _k, v in recs) / ps
    fl = sum(v["flops"] for _n, _k, v in recs) / ps
    by = sum(v["bytes"] for _n, _k, v in recs) / ps
    ev_ms = sum(v["ms"] for _n, _k, v in recs) / ps
    # a C-ABI call may launch helper kernels beside its main one (combine, zero): the main kernel's count is what must agree
    verified = bool(recs) and abs(calls - l_graph) <= 0.26 * max(l_graph, 1.0)
    t = us_step * 1e-6
    tf, gbs = fl / t / 1e12, by / t / 1e9
    blk = {"kernel": name, "launches_per_step": round(l_graph, 2), "avg_launch_us": us_step / max(l_graph, 1e-9), "ms_per_step": us_step / 1e3,
           "timing": "in-graph kernel durations of this run (live rocprofv3 --kernel-trace child)",
           "entry_points": sorted({f"{n} {k}" for n, k, _v in recs})[:8], "mapping_verified": verified,
           "launches_per_step_eager_calls": round(calls, 2), "ms_per_step_eager_events": ev_ms,
           "algorithmic_gflop_per_step": fl / 1e9, "algorithmic_bytes_per_launch": by / max(calls, 1e-9),
           "frac_mfma": tf / peak_tflops, "frac_hbm": gbs / PEAK_HBM_GBS,
           "hbm_bytes_basis": "algorithmic bytes (compulsory operand + result bytes of each launch)", "traffic": None}
    if blk["frac_mfma"] >= blk["frac_hbm"]:
        blk.update({"bound": "mfma", "achieved": tf, "peak": peak_tflops, "unit": "TFLOP/s", "frac": blk["frac_mfma"]})
    else:
        blk.update({"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": blk["frac_hbm"]})
    per_launch, corr = pmc_per_launch(pmc, name)
    if per_launch is not None:
        blk.update({"traffic": per_launch, "traffic_correction": corr, "traffic_source": pmc.get("_file"),
                    "traffic_over_algorithmic": per_launch * l_graph / max(by, 1.0), "measured_traffic_gbs": per_launch * l_graph / t / 1e9})
    if mfma is not None:
        rows = mfma.get("kernels", {}).get(name, [])
        n = sum(r["launches"] for r in rows)
        if n:
            cyc = sum(r["mfma_busy_cycles"] * r["launches"] for r in rows) / n * l_graph
            blk["mfma_busy"] = cyc / (t * mfma.get("clock_ghz", 2.4) * 1e9 * mfma.get("simds", 1024))
            blk["mfma_busy_source"] = mfma.get("_file")
    return blk


CALIBRATION_BYTES = 100 << 20   # the PMC calibration launch: copy_many_kernel streams this many bytes in and out (16 B / lane)


def committed_profiles(config, compute, ids):
    """The committed evidence of this config / compute mode under profiles/ that belongs to THIS build: a file is used only when the
    `src_sha256` stamped into it (tools/pmc_*_json.py: "_build"; tools/rocprof_summary.py: "# build:" header) equals the running
    build's (pq3d_amd.build.build_ids) -- a stale file is refused and named in `refused`.  Returns (PMC traffic json or None, PMC
    MFMA json or None, {kernel: (launches per step, us per step)} of the committed rocprofv3 kernel stats or {}, info dict)."""
    import glob
    import re
    pdir = os.path.join(ROOT, "profiles")
    sfx = "" if compute == DEFAULT_COMPUTE else f"_{compute}"
    tags = sorted({m.group(1) for f in glob.glob(os.path.join(pdir, f"*_{config}{sfx}.*")) for m in [re.search(r"_(r\d\d)_", f)] if m})
    pmc = mfma = None
    stats, info = {}, {"build": ids, "refused": []}
    src = ids.get("src_sha256")

    def load(path):
        try:
            j = json.load(open(path))
        except (OSError, ValueError):
            return None
        if (j.get("_build") or {}).get("src_sha256") != src:
            info["refused"].append(f"profiles/{os.path.basename(path)} (build {(j.get('_build') or {}).get('src_sha256')} != {src})")
            return None
        j["_file"] = f"profiles/{os.path.basename(path)}"
        return j
    for tag in reversed(tags):
        if pmc is None:
            pmc = load(os.path.join(pdir, f"pmc_traffic_{tag}_{config}{sfx}.json"))
        if mfma is None:
            mfma = load(os.path.join(pdir, f"pmc_mfma_{tag}_{config}{sfx}.json"))
        st = os.path.join(pdir, f"rocprofv3_kernel_stats_{tag}_fused_graph_{config}{sfx}.txt")
        if not stats and os.path.exists(st):
            lines = open(st).read().splitlines()
            mb = re.search(r"src_sha256 (\w+)", lines[0]) if lines else None
            if not mb or mb.group(1) != src:
                info["refused"].append(f"profiles/{os.path.basename(st)} (build {mb.group(1) if mb else None} != {src})")
                continue
            hdr = next((ln for ln in lines[:3] if "steps" in ln), "")
            m = re.search(r"steps (\d+)", hdr)
            steps = float(m.group(1)) if m else None
            for ln in lines:
                mm = re.match(r"^(\S.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", ln)
                if mm and steps:
                    name = re.sub(r"[<(].*$", "", mm.group(1).replace("(anonymous namespace)::", "").replace("void ", ""))
                    c_, tot = stats.get(name, (0.0, 0.0))
                    stats[name] = (c_ + int(mm.group(2)) / steps, tot + float(mm.group(3)) * 1e3 / steps)   # launches, us per step
            info["kernel_stats_file"] = f"profiles/{os.path.basename(st)}"
    return pmc, mfma, stats, info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS) + ["pool"],
                    help="c2 = the BASELINE metric's configuration (default); pool = segment pooling alone (bench_pool.py)")
    ap.add_argument("--pool-nvox", type=int, default=250_000, help="--config pool: voxels per scene")
    ap.add_argument("--pool-pmc", action="store_true", help="--config pool: only the headline launches (counter passes)")
    ap.add_argument("--pool-segments", type=int, default=4096, help="--config pool: segments per scene (max_seg)")
    ap.add_argument("--compute", default=DEFAULT_COMPUTE, choices=["bf16", "bf16x3", "fp32"])
    ap.add_argument("--no-live-trace", action="store_true",
                    help="skip the child rocprofv3 --kernel-trace run that times the kernels of the replayed step for `roofline`")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--min-time", type=float, default=1.0,
                    help="when K timed steps take < 0.5 s, repeat the K-step region until this many seconds are timed in "
                         "total (median over repeats reported; 0 = a single K-step region)")
    ap.add_argument("--max-repeats", type=int, default=400)
    ap.add_argument("--cpu-steps", type=int, default=32, help="timed CPU-baseline steps (0 disables)")
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--dump-kernels", default=None, help="write the per-(entry point, shape) timing table here")
    ap.add_argument("--pmc-calibration", action="store_true",
                    help="after the profiled pass launch ONE known streaming copy (100 MiB in, 100 MiB out, 16 B / lane): "
                         "under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE its raw counters calibrate the traffic numbers")
    ap.add_argument("--dropout", default="off", choices=["off", "reference"],
                    help="headline run: 'off' = p=0 (the parity-checked arithmetic); 'reference' = the reference's "
                         "train-mode probabilities (0.1 in the decoder layers and encoders, 0.1/0.3 in the heads)")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the headline measurement + roofline pass (no eager / dropout / optimizer legs); what the "
                         "rocprofv3 runs under profiles/ use so that the trace holds the headline step only")
    ap.add_argument("--no-optimizer-leg", action="store_true",
                    help="skip the extra full-training-step (with optimizer) timing reported next to the headline")
    ap.add_argument("--no-parity-leg", action="store_true",
                    help="skip the bf16-vs-fp32 compute-type comparison (parity_modes) reported next to the headline (N=1 only)")
    ap.add_argument("--no-dropout-leg", action="store_true",
                    help="skip the extra train-mode-dropout timing reported next to the headline (N=1 only)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, one process per GPU (the reference launches its own
        # ranks too: launch.py:62-64, common/launch_utils.py:50-121) -- re-exec under torch.distributed.run on this node
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__),
                                  *sys.argv[1:]])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: a line for {args.gpus} GPUs must come from "
                         f"{args.gpus} ranks (launch with torch.distributed.run --nproc-per-node {args.gpus}, or plain "
                         f"`python bench.py --gpus {args.gpus}`, which starts them)")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback on the product path)"
    # PQ3D_BENCH_BACKEND=gloo is a test hook (tests/test_gpu_bench_multirank.py): it lets two ranks share ONE GPU so the
    # whole N > 1 flow (sharded inputs, gradient all-reduce, barriers, max-over-ranks timing, rank-0 JSON) is exercised on
    # a single-GPU box.  RCCL itself refuses two ranks on one device; the driver's N > 1 runs use the default 'nccl'.
    backend = os.environ.get("PQ3D_BENCH_BACKEND", "nccl")
    local = local % torch.cuda.device_count() if backend != "nccl" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # PQ3D_BENCH_FORCE_DIST=1 (test hook): run the data-parallel step flow with ONE rank -- a real RCCL communicator of
    # size 1, every bucket's all-reduce issued (side stream, ReduceOp.AVG) and, in the default step mode, captured inside
    # the HIP graph.  A one-GPU box cannot run two RCCL ranks; this is how the 'nccl' branch executes on hardware at all.
    dist_on = world > 1 or os.environ.get("PQ3D_BENCH_FORCE_DIST", "0") == "1"
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    if args.config == "pool":   # segment pooling as its own workload: bench_pool.py
        import bench_pool

        def pool_barrier():
            if dist_on:
                torch.distributed.barrier()
        res = bench_pool.run(args, dev, world, rank, pool_barrier, dist_on)
        if rank == 0:
            print(json.dumps(res))
        if dist_on:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return
    c = dict(CONFIGS[args.config])
    model, sd, dd_cpu = build(c, args.compute, dev, seed=1234 + rank)
    dd = {k: v.to(dev) for k, v in dd_cpu.items()}
    model.train()
    ref_dropout = {m: m.dropout_p for m in model.modules() if hasattr(m, "dropout_p")}   # constructor defaults

    def set_dropout_mode(mode):
        for m, p in ref_dropout.items():
            m.dropout_p = p if mode == "reference" else 0.0

    set_dropout_mode(args.dropout)
    params = [p for p in model.parameters() if p.requires_grad]
    # Gradient buckets in the order they become final ("bucketed per decoder layer in reverse execution order"):
    # [layer L-1, ..., layer 0, mask head, everything else (encoders, heads: complete at the end of backward)].  The fused
    # decoder writes the decoder buckets in place (no pack copy).  In a data-parallel run it reports readiness from inside
    # the backward (enc.grads_ready(tag)): with per-layer flushing (enc.grad_bucket_per_layer) tag = layer index as soon as
    # that layer's gradients are final, and tag = "decoder" when every decoder gradient is -- before the key/value
    # input-gradient products and the encoders' backward, which the all-reduces then overlap.
    enc = model.unified_encoder
    layers = list(enc.unified_encoder)
    lay_ids = [{id(p) for p in l.parameters()} for l in layers]
    mh_ids = {id(p) for p in model.mask_head.parameters()} if hasattr(model, "mask_head") else set()
    dec_ids = set().union(*lay_ids) | mh_ids
    # Bucket layout (PQ3D_BENCH_BUCKETS):
    #   coalesced (default): [output heads] [decoder layers + mask head] [everything else] -- at most 3 collectives per step.
    #     With one-rank RCCL the per-layer layout cost 0.23 ms per step of pure launch / stream-join overhead at config 2 (1.646
    #     vs 1.416 ms, profiles/rccl_one_rank_modes_r04.txt) for an overlap window of ~0.3 ms: a few large messages beat many
    #     small ones on point-to-point xGMI as well.
    #   per_layer: [heads] [layer L-1] ... [layer 0] [mask head] [everything else], launched from inside the backward as each
    #     layer's gradients become final (one_graph / eager modes: the only way to overlap there).
    # The heads' bucket (they run backward FIRST, query3d_unified.py:193-218; config 5: the caption body's 240 MB) is launched
    # when the fused decoder's backward starts (enc.grads_ready('heads')) and reduces under the whole decoder backward.  That is
    # only correct when every head parameter's gradient is written IN PLACE through the arena before that point (round 4: it
    # was not -- grounding head all zeros, caption body 1 % short); FlatGradAllReducer.launch() flushes the deferred weight
    # gradients first, tools/probes/heads_arena_probe.py checks the precondition per configuration and the per-bucket
    # fingerprints in the JSON verify every run.  PQ3D_BENCH_EARLY_HEADS=0 puts the heads back into the last bucket.
    bucket_mode = os.environ.get("PQ3D_BENCH_BUCKETS", "coalesced")
    side_stream = os.environ.get("PQ3D_BENCH_SIDE_STREAM", "0") == "1"   # replayed pieces: collectives through a side stream
    early_heads = os.environ.get("PQ3D_BENCH_EARLY_HEADS", "1") != "0"
    head_ids = set()
    if early_heads:
        for hn in ("generation_head", "ground_head"):
            if hasattr(model, hn):
                head_ids |= {id(p) for p in getattr(model, hn).parameters() if p.requires_grad}
    P = lambda ids: [p for p in params if id(p) in ids]
    if bucket_mode == "per_layer":
        layer_groups = [P(lay_ids[i]) for i in reversed(range(len(layers)))] + [P(mh_ids)]
    else:
        layer_groups = [P(dec_ids)]
    groups = [P(head_ids)] + layer_groups + [[p for p in params if id(p) not in dec_ids and id(p) not in head_ids]]
    keep = [bool(g) for g in groups]
    bucket_of = [sum(keep[:j]) for j in range(len(groups))]            # index after dropping empty groups
    # PQ3D_BENCH_WIRE=bf16: gradient buckets cross the links as bf16 with fp32 accumulation (parallel.FlatGradAllReducer wire_dtype;
    # default fp32 = DDP's arithmetic, trainer/build.py:66-75)
    wire = os.environ.get("PQ3D_BENCH_WIRE", "fp32")
    # PQ3D_BENCH_COMM=native: the buckets go through the C-ABI communicator (pq3d_comm_init / pq3d_allreduce_grads[_wire], csrc/comm.hip)
    # instead of torch.distributed's collectives; torch.distributed still carries the rendezvous and the barriers
    exchange = os.environ.get("PQ3D_BENCH_COMM", "torch")
    comm = None
    if exchange == "native" and dist_on and backend == "nccl":
        from pq3d_amd.parallel import NativeComm
        comm = NativeComm.from_process_group()
    reducer = FlatGradAllReducer(params, groups=[g for g in groups if g], wire_dtype=torch.bfloat16 if wire == "bf16" else None, comm=comm)
    heads_bucket = bucket_of[0] if keep[0] else None
    dec_buckets = [bucket_of[j] for j in range(1, 1 + len(layer_groups)) if keep[j]]
    n_layer_buckets = len(layers) if bucket_mode == "per_layer" else 0
    enc.grad_arena = reducer.slots()
    enc.grad_arena_buffers = list(reducer.flat)   # all of them: the encoders' backward writes its slots in place too
    reducer.force_collectives = dist_on and world == 1
    overlap = dist_on and os.environ.get("PQ3D_BENCH_OVERLAP", "1") != "0"
    forced_mode = os.environ.get("PQ3D_BENCH_STEP_MODE", "")   # "", "one_graph", "two_graph", "graph_then_allreduce", "eager"

    def on_ready(tag):
        """Start the all-reduce of what just became final (side stream; the backward carries on)."""
        if tag == "heads":
            if heads_bucket is not None:
                reducer.launch(heads_bucket)
        elif tag == "decoder":
            for b in dec_buckets:
                reducer.launch(b)
        elif n_layer_buckets:
            reducer.launch(bucket_of[1 + n_layer_buckets - 1 - int(tag)])

    one = torch.ones((), device=dev)

    def fwd_bwd():
        model.zero_grad(set_to_none=True)
        out = model(dict(dd))
        loss = loss_fn(out, c["heads"])
        if enc.grad_arena is not None:
            with ops.grad_arena(enc.grad_arena, enc.grad_arena_buffers, pack_follows=True):   # every slot offered for the whole pass
                loss.backward(gradient=one)   # cached seed gradient: no ones_like fill in the step
        else:
            loss.backward(gradient=one)
        reducer.pack()

    def full_step():
        """forward + backward with the decoder buckets' all-reduces launched from inside the backward, then the rest + join."""
        fwd_bwd()
        reducer.finish()

    class SplitGraphStep:
        """RCCL-independent overlap: the step as SEVERAL captured graphs split where a bucket becomes final -- at the start of
        the fused decoder's backward (enc.grads_ready('heads'): the output heads' gradients; only with an early heads bucket)
        and where every decoder gradient is final (enc.grads_ready('decoder')).  After each piece the finished buckets'
        all-reduces are launched EAGERLY on the side stream and run while the next piece replays (heads: under the whole
        decoder backward; decoder: under the key/value input-gradient products + the encoders' backward + the pack of the last
        bucket); finish() joins.  The capture is split from inside the autograd backward (relaxed capture mode: the backward
        runs on autograd's device thread), all graphs share one memory pool."""

        def __init__(self):
            self.tags = (["heads"] if heads_bucket is not None else []) + ["decoder"]
            self.graphs, self.seen = [torch.cuda.CUDAGraph()], []
            stream = torch.cuda.Stream()
            stream.wait_stream(torch.cuda.current_stream())

            def split(tag):
                if tag not in self.tags or tag in self.seen:
                    return
                self.seen.append(tag)
                self.graphs[-1].capture_end()
                g = torch.cuda.CUDAGraph()
                g.capture_begin(pool=self.graphs[0].pool(), capture_error_mode="relaxed")
                self.graphs.append(g)
            enc.grads_ready, enc.grad_bucket_per_layer = split, False
            import gc
            gc.collect(); torch.cuda.empty_cache()
            with torch.cuda.stream(stream):
                self.graphs[0].capture_begin(capture_error_mode="relaxed")
                try:
                    fwd_bwd()
                finally:
                    self.graphs[-1].capture_end()
            torch.cuda.current_stream().wait_stream(stream)
            enc.grads_ready = None
            if "decoder" not in self.seen:
                raise RuntimeError("the backward never reported 'decoder' readiness (no fused decoder on this configuration)")
            self.evs = [torch.cuda.Event() for _ in self.seen]
            self.buckets = [([heads_bucket] if t == "heads" else list(dec_buckets)) for t in self.seen]

        def __call__(self):
            self.graphs[0].replay()
            if side_stream:
                for k, bs in enumerate(self.buckets):
                    self.evs[k].record()           # the buckets of piece k are final here
                    self.graphs[k + 1].replay()    # enqueued BEFORE the collectives: the device goes straight into the next piece
                    for b in bs:                   # while the host is still issuing the all-reduces (side stream, behind the event)
                        reducer.launch(b, after=self.evs[k])
                reducer.finish()
            else:
                # default: no side stream -- each collective is issued from the current stream's position between two pieces
                # (the backend's communication stream picks up there), the next piece is enqueued behind the CALL, not behind
                # the transfer, and everything is joined once at the end: 2 stream hops per bucket instead of 4
                for k, bs in enumerate(self.buckets):
                    for b in bs:
                        reducer.launch(b, side=False)
                    self.graphs[k + 1].replay()
                reducer.finish(side=False)

    def capture():
        """3 eager steps on a side stream (allocator + autograd warm-up), then capture the step.
        Returns (callable or None, mode).  Data-parallel runs use, in this order: (1) only on request
        (PQ3D_BENCH_STEP_MODE=one_graph) ONE graph with the RCCL all-reduces inside, on the capturing stream at the point
        each layer's bucket becomes final -- no host work per step, but no overlap either (branches of one graph do not run
        concurrently on this runtime), and capturing them on a forked side stream crashes hipStreamEndCapture
        (tools/probes/rccl_capture_probe.py); (2) DEFAULT: TWO graphs split at decoder-gradients-final with the collectives
        launched eagerly in between on a side stream, overlapping graph B (validated over RCCL with a one-rank communicator
        on an MI355X and over gloo with two ranks); (3) forward+backward in one graph, all-reduces after the replay (no
        overlap); (4) eager."""
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        enc.grads_ready, enc.grad_bucket_per_layer = (on_ready if overlap else None), overlap and bucket_mode == "per_layer"
        with torch.cuda.stream(s):
            for _ in range(3):
                full_step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        if args.no_graph or forced_mode == "eager":
            return None, "eager" + ("+overlap(per-layer buckets)" if overlap else "")

        def note(what, e):
            if rank == 0:
                print(f"[bench] {what} failed ({type(e).__name__}: {e})", file=sys.stderr)
            torch.cuda.synchronize()
            reducer._pending, reducer._launched = [], set()

        if overlap and backend == "nccl" and forced_mode == "one_graph":
            try:
                g = torch.cuda.CUDAGraph()
                # thread_local: the process group's watchdog thread may touch the runtime while this thread captures
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    full_step()
                g.replay()                      # one checked replay: asynchronous collective errors surface here, inside
                torch.cuda.synchronize()        # the try, and the run falls back
                return g.replay, f"graph(step+allreduce, {len(dec_buckets)} decoder bucket(s) all-reduced in-stream from inside the backward, buckets={bucket_mode})"
            except Exception as e:  # noqa: BLE001
                note("capture with the collectives inside", e)
        if overlap and forced_mode in ("", "two_graph"):
            try:
                tg = SplitGraphStep()
                tg()
                torch.cuda.synchronize()
                return tg, (f"{len(tg.graphs)} graphs split at {' / '.join(tg.seen)}-gradients-final, "
                            f"{sum(len(b) for b in tg.buckets)} bucket(s) all-reduced eagerly between them (overlapping the next piece), "
                            f"buckets={bucket_mode}")
            except Exception as e:  # noqa: BLE001
                note("two-graph capture", e)
        enc.grads_ready, enc.grad_bucket_per_layer = None, False
        try:
            g = torch.cuda.CUDAGraph()
            # with a process group alive its watchdog thread polls events of earlier collectives (hipEventQuery): under the
            # default 'global' capture mode that aborts the process ("operation not permitted when stream is capturing",
            # seen once in four runs of the one-rank RCCL test) -- captures next to RCCL are thread-local / relaxed
            with torch.cuda.graph(g, **({"capture_error_mode": "thread_local"} if dist_on else {})):
                fwd_bwd()

            def run():
                g.replay()
                reducer.finish(side=side_stream)
            return (run if dist_on else g.replay), "graph(fwd+bwd) then allreduce"
        except Exception as e:  # noqa: BLE001
            note("HIP graph capture", e)
            return None, "eager"

    graph, step_mode = capture()
    if graph is None:
        enc.grads_ready, enc.grad_bucket_per_layer = (on_ready if overlap else None), overlap and bucket_mode == "per_layer"

    def step():
        if graph is not None:
            graph()
        else:
            full_step()

    def barrier():
        if dist_on:
            torch.distributed.barrier()

    def timed_k_steps():
        """EXACTLY args.steps steps bracketed by barrier + synchronize on both sides; max over ranks (seconds)."""
        barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize(); barrier()
        dt = time.perf_counter() - t0
        if dist_on:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    for _ in range(args.warmup):
        step()
    # A K-step region of a ~1.4 ms step is a few tens of ms: one clock ramp or a noisy neighbour moves it by percents and a
    # 1 Hz SMI sampler never sees the GPU busy.  When K steps take < 0.5 s the K-step region is REPEATED
    # (each repeat = exactly K steps with the same brackets) until >= args.min_time seconds are timed in total;
    # ms_per_step = MEDIAN over the repeats, min / max beside it.  steps / warmup keep their meaning.
    dts = [timed_k_steps()]
    if dts[0] < 0.5 and args.min_time > 0:
        repeats = int(min(args.max_repeats, max(2, -(-args.min_time // dts[0]))))
        if dist_on:   # every rank must run the same number of collectives: rank 0's count wins
            r = torch.tensor([repeats], device=dev, dtype=torch.int64)
            torch.distributed.broadcast(r, 0)
            repeats = int(r.item())
        dts += [timed_k_steps() for _ in range(repeats - 1)]
    sdts = sorted(dts)
    dt = sdts[len(sdts) // 2] if len(sdts) % 2 else 0.5 * (sdts[len(sdts) // 2 - 1] + sdts[len(sdts) // 2])
    ms = dt / args.steps * 1e3
    value = c["B"] * world * args.steps / dt
    # the one-launch chains' error word (csrc/chain_common.h: a hand-off that ran past its bound sets it and the launch goes on with
    # stale rows): read ONCE behind the timed region on every rank; set anywhere -> "chain_error": true in the line and exit code 3
    from pq3d_amd import ops as _ops
    chain_err = bool(_ops.chain_error(dev))
    if dist_on:
        ce = torch.tensor([int(chain_err)], device=dev, dtype=torch.int32)
        torch.distributed.all_reduce(ce, op=torch.distributed.ReduceOp.MAX)
        chain_err = bool(int(ce.item()))
    grads_identical = None
    if dist_on:   # after the all-reduce every rank must hold the same (mean) gradient: fingerprint min == max over ranks
        fp = torch.stack([torch.stack([f.double().sum(), f.double().abs().sum()]) for f in reducer.flat]).flatten()
        lo, hi = fp.clone(), fp.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        grads_identical = bool(torch.equal(lo, hi)) and bool(torch.isfinite(fp).all()) and bool((fp[1::2] > 0).all())
        grad_fingerprint = [[float(fp[2 * i]), float(fp[2 * i + 1])] for i in range(len(reducer.flat))]

    result = None
    if rank == 0:
        flops = step_flops(c)
        peak = PEAK_BF16_TFLOPS if args.compute in ("bf16", "bf16x3") else PEAK_F32_TFLOPS
        # per-kernel attribution: eager profiled pass with HIP events on the launch stream (rank 0 alone: no collectives)
        enc.grads_ready, enc.grad_bucket_per_layer = None, False
        with KernelTimer() as kt:
            for _ in range(args.profile_steps):
                fwd_bwd()
        summ = kt.summary()
        if args.pmc_calibration:
            n_cal = CALIBRATION_BYTES // 4
            csrc, cdst = torch.ones(n_cal, device=dev), torch.empty(n_cal, device=dev)
            ops.copy_many([cdst], [csrc])
            torch.cuda.synchronize()
            del csrc, cdst
        ps = max(args.profile_steps, 1)
        tot = sum(v["ms"] for v in summ.values())
        fams = {}
        for (n, _k), v in summ.items():
            f = fams.setdefault(n, dict(ms=0.0, calls=0, flops=0.0, bytes=0.0))
            for kk in ("ms", "calls", "flops", "bytes"):
                f[kk] += v[kk]
        # ---- roofline of the run's real dominant KERNEL: time = in-graph kernel durations of THIS run (a child
        # rocprofv3 --kernel-trace of the same build / box / config), work = the algorithmic bytes / FLOPs of the C-ABI calls that
        # launch the kernel (this run's eager profiled pass), traffic = the committed PMC passes IF they are of this build
        from pq3d_amd.build import build_ids
        ids = build_ids()
        pmc, mfma, cstats, pinfo = committed_profiles(args.config, args.compute, ids)
        trace, tmeta = ({}, "not the single-GPU headline run") if (world > 1 or dist_on) else live_kernel_trace(args)
        timing_src = "live"
        if not trace and cstats:
            trace, tmeta, timing_src = cstats, {"source": f"committed {pinfo.get('kernel_stats_file')} of this build (src_sha256 {ids['src_sha256']})"}, "committed"
        blocks = []
        if trace:
            order = sorted(trace, key=lambda n: -trace[n][1])
            blocks = [kernel_block(n, trace, summ, ps, peak, pmc, mfma) for n in order[:4] if not n.startswith("__amd_rocclr")][:3]
            ktot = sum(v[1] for v in trace.values())
            for b_ in blocks:
                b_["share_of_kernel_time"] = trace[b_["kernel"]][1] / ktot if ktot else 0.0
                if timing_src == "committed":
                    b_["timing"] = "in-graph kernel durations, " + tmeta["source"]
                if cstats and b_["kernel"] in cstats and timing_src == "live":   # the committed summary of the same build must agree
                    b_["committed_profile_avg_launch_us"] = cstats[b_["kernel"]][1] / max(cstats[b_["kernel"]][0], 1e-9)
        else:
            # no trace (rocprofv3 absent / child failed): this run's own HIP-event times per C-ABI entry point, stated as such
            order = sorted(fams, key=lambda n: -fams[n]["ms"])
            for n in order[:3]:
                v = fams[n]
                t = v["ms"] / ps * 1e-3
                tf, gbs = v["flops"] / ps / t / 1e12, v["bytes"] / ps / t / 1e9
                blk = {"kernel": n, "gpu_kernels": FAMILY_KERNELS.get(n, []), "launches_per_step": v["calls"] / ps,
                       "avg_launch_us": v["ms"] / v["calls"] * 1e3, "ms_per_step": v["ms"] / ps,
                       "timing": "HIP events on the launch stream around each C-ABI call, eager profiled pass of this run (includes launch gaps)",
                       "frac_mfma": tf / peak, "frac_hbm": gbs / PEAK_HBM_GBS, "traffic": None}
                blk.update({"bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak} if tf / peak >= gbs / PEAK_HBM_GBS
                           else {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS})
                blocks.append(blk)
        roof = blocks[0] if blocks else {"kernel": None, "bound": "mfma", "achieved": 0.0, "peak": peak, "unit": "TFLOP/s", "frac": 0.0, "traffic": None}
        roof["kernel_ms_per_step_eager_events"] = tot / ps
        roof["trace"] = tmeta
        roof["build"] = ids
        if pinfo["refused"]:
            roof["stale_profiles_refused"] = pinfo["refused"]
        # step level: measured memory-side bytes of ALL kernels of a step (PMC passes of this build x in-graph launch counts) against
        # SURVEY 8(d)'s closed-form compulsory bytes (ideal fusion: nothing but weights, memories and the query state moves)
        step_alg_bytes = step_bytes(c)
        step_traffic = None
        if pmc is not None and trace:
            tb = 0.0
            for n, (lps, _us) in trace.items():
                pl, _c = pmc_per_launch(pmc, n)
                if pl is not None:
                    tb += pl * lps
            step_traffic = {"measured_traffic_bytes": tb, "algorithmic_bytes_survey_8d": step_alg_bytes,
                            "traffic_over_algorithmic": tb / step_alg_bytes, "sustained_gbs": tb / (ms * 1e-3) / 1e9,
                            "source": pmc.get("_file"), "note": "sum over every kernel of the step: calibrated FETCH_SIZE + WRITE_SIZE per "
                            "launch x launches per step of the in-graph trace"}
        else:
            step_traffic = {"measured_traffic_bytes": None, "algorithmic_bytes_survey_8d": step_alg_bytes, "traffic_over_algorithmic": None,
                            "note": "no PMC passes of this build under profiles/ (tools/refresh_profiles.sh writes them)"}
        if args.dump_kernels:
            with open(args.dump_kernels, "w") as f:
                f.write(f"{'entry point':18s} {'shape':44s} {'calls/step':>10s} {'us/launch':>10s} {'ms/step':>8s} "
                        f"{'TFLOP/s':>8s} {'GB/s':>8s}\n")
                for (n, k), v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
                    us = v["ms"] / v["calls"] * 1e3
                    f.write(f"{n:18s} {k:44s} {v['calls'] / args.profile_steps:10.1f} {us:10.1f} "
                            f"{v['ms'] / args.profile_steps:8.3f} {v['flops'] / v['calls'] / us / 1e6:8.2f} "
                            f"{v['bytes'] / v['calls'] / us / 1e3:8.1f}\n")
        result = {
            "metric": "decoder fwd+bwd scenes/sec at (N_seg=1024,N_q=100,d=256,L=4)" if args.config == "c2"
            else f"decoder fwd+bwd scenes/sec ({args.config})",
            "value": value, "unit": "scenes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "repeats": len(dts), "ms_per_step_min": sdts[0] / args.steps * 1e3, "ms_per_step_max": sdts[-1] / args.steps * 1e3,
            "timed_total_s": sum(dts),
            "timing_note": "each repeat = exactly `steps` steps between barrier + synchronize brackets (max over ranks); "
                           "ms_per_step / value = median over `repeats`",
            "dtype": "bf16" if args.compute in ("bf16", "bf16x3") else "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE config {args.config}: B={c['B']} scenes/GPU, N_seg={c['Ns']}, "
                                   f"N_q={c['Nq']}, d={c['d']}, H={c['H']}, L={c['L']}, memories={c['memories']}, "
                                   f"{c.get('structure', 'parallel')} cross-attn + spatial self-attn + FFN2048, heads={c['heads']}, "
                                   f"fwd+bwd+grad-pack{'+RCCL all-reduce' if dist_on else ''}",
                       "global_batch": c["B"] * world, "parallelism": f"dp{world}", "hip_graph": graph is not None,
                       "step_mode": step_mode, "gradient_wire_dtype": wire,
                       "gradient_exchange": ("pq3d_comm (C-ABI, RCCL)" if comm is not None else "torch.distributed (RCCL)") if dist_on else None,
                       "compute_mode": {"bf16x3": "bf16x3: bf16 MFMA operands, fp32 accumulation; the key/value side AND the query side carry "
                                                  "their fp32 tensors as hi + lo bf16 pairs (3 MFMAs per product) in the forward, single-bf16 "
                                                  "backward -- meets north_star's 1e-3 end to end",
                                        "bf16": "bf16: as bf16x3 with a single-bf16 key/value side (3e-3 .. 7e-3 end to end)",
                                        "fp32": "fp32: exact-f32 MFMA"}[args.compute],
                       "dropout": 0.0 if args.dropout == "off" else "reference train mode (0.1 / heads 0.1, 0.3)",
                       # the caption body keeps the HF config's dropout_rate under model.train() in BOTH dropout settings
                       # (pq3d_amd/t5.py: the reference never overrides it) -- extra work inside the timed step, stated here
                       **({"caption_body_dropout": "0.1 (T5 config dropout_rate, live: model.train())"} if "generation" in c["heads"] else {}),
                       "activation": "relu"},
            "step_algorithmic_gflop": flops / 1e9,
            "step_roofline_frac": flops * world / (dt / args.steps) / (peak * 1e12 * world),
            **({"grads_identical_across_ranks": grads_identical, "grad_fingerprint_per_bucket": grad_fingerprint,
                "collective_backend": backend,
                "rccl_ranks": (torch.distributed.get_world_size() if backend == "nccl" else 0)} if dist_on else {}),
            "roofline": roof,
            "roofline_next": blocks[1:],
            "step_traffic": step_traffic,
            "kernel_families_ms_per_step": {k: round(v["ms"] / ps, 4) for k, v in sorted(fams.items())},
        }
        if world == 1 and not dist_on and not args.headline_only:
            # the same step launched eagerly (no HIP graph): what a trainer that cannot capture graphs would see
            def timed_loop(fn, n):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t1) / n * 1e3
            n_e = max(3, args.steps // 5)
            fwd_bwd()
            result["eager_ms_per_step"] = timed_loop(fwd_bwd, n_e)
            # the drop-in path for a trainer that calls model(data_dict) / loss.backward() itself (reference:
            # trainer/query3d_trainer.py:30-45): forward and backward as ONE graph replay each behind an autograd node
            # (pq3d_amd/graphed.py); the loss and the gradient plumbing stay eager
            try:
                from pq3d_amd.graphed import GraphedQuery3D
                for mode in ("direct", "autograd"):
                    enc.grad_arena = None
                    gm = GraphedQuery3D(model, dd, mode=mode)

                    def dropin_step():
                        model.zero_grad(set_to_none=True)
                        loss_fn(gm(dd), c["heads"]).backward()
                    for _ in range(3):
                        dropin_step()
                    result[f"dropin_graphed_{mode}_ms_per_step"] = timed_loop(dropin_step, max(10, args.steps))
                    del gm, dropin_step
                    import gc
                    torch.cuda.synchronize(); gc.collect()   # tear the captured graphs down here, on this thread, device idle
            except Exception as e:  # noqa: BLE001
                result["dropin_graphed_error"] = f"{type(e).__name__}: {e}"[:300]
            enc.grad_arena, enc.grad_arena_buffers = reducer.slots(), list(reducer.flat)
            if args.dropout == "off" and not args.no_dropout_leg:
                # train-mode dropout as the reference trains (reported separately from the parity-checked
                # p=0 headline): masks are generated inside the attention / LayerNorm / GEMM-epilogue kernels
                set_dropout_mode("reference")
                g2, _ = capture()
                run2 = g2 if g2 is not None else fwd_bwd
                for _ in range(args.warmup):
                    run2()
                ms2 = timed_loop(run2, args.steps)
                result["train_mode_dropout"] = {
                    "p": {"decoder_layers": 0.1, "mask_head_cls": 0.1, "ground_head": 0.3, "object_encoders": 0.1},
                    "ms_per_step": ms2, "value": c["B"] / (ms2 * 1e-3), "unit": "scenes/s", "hip_graph": g2 is not None,
                    "note": "parity of the dropout arithmetic: tests/test_gpu_dropout.py (same masks fed to the oracle)"}
                set_dropout_mode(args.dropout)
            if args.compute in ("bf16", "bf16x3") and not args.no_parity_leg:
                # north_star: "within 1e-3 bf16 / 1e-5 fp32".  The three compute modes of THIS workload side by side: step time and the
                # end-to-end distance of the final queries from the exact-f32 mode's (itself within 1e-5 of the CPU oracle at full size:
                # tests/test_gpu_fullsize.py).  The headline mode's entry carries the headline's own time.
                notes = {
                    "bf16": "single-bf16 MFMA operands and bf16 K / V / Q / P / O storage on the key/value side, split-bf16 (fp32-grade) "
                            "query side: per sub-layer <= 1e-3 (tests/test_gpu_sublayer_parity.py), 3e-3 .. 7e-3 end to end -- faster, "
                            "but NOT inside north_star's 1e-3 end to end",
                    "bf16x3": "split-bf16 key/value side as well (hi + lo bf16 planes of K / V, q, P, O: csrc/gemm_x3p.hip, csrc/attn_x3.hip; "
                              "3 MFMAs per product): forward within north_star's 1e-3 of the fp32 oracle end to end and every gradient "
                              "within 2e-2 at full size (tests/test_gpu_fullsize.py::test_bf16x3_fullsize_meets_north_star_tolerance); "
                              "single-bf16 backward",
                    "fp32": "exact-f32 MFMA everywhere (<= 1e-5 of the CPU oracle, tests/test_gpu_fullsize.py)"}
                try:
                    from pq3d_amd.modules import set_compute
                    qk = lambda o: (o["query_embeds"] if "query_embeds" in o else o["query"]).detach().float()
                    qs, times = {}, {args.compute: ms}
                    for mode in ("fp32", "bf16", "bf16x3"):
                        set_compute(model, mode)
                        with torch.no_grad():
                            qs[mode] = qk(model(dict(dd)))
                        if mode != args.compute:
                            gm_, _ = capture()
                            runm = gm_ if gm_ is not None else fwd_bwd
                            for _ in range(args.warmup):
                                runm()
                            times[mode] = timed_loop(runm, max(10, args.steps // 2))
                            del gm_, runm
                    set_compute(model, args.compute)
                    ref = qs["fp32"]
                    result["parity_modes"] = {
                        m_: {"ms_per_step": times[m_], "value": c["B"] / (times[m_] * 1e-3), "unit": "scenes/s",
                             **({"query_err_vs_fp32_mode": float((qs[m_] - ref).abs().max() / ref.abs().max())} if m_ != "fp32" else {}),
                             "headline": m_ == args.compute, "note": notes[m_]} for m_ in ("bf16x3", "bf16", "fp32")}
                    import gc
                    torch.cuda.synchronize(); gc.collect()
                except Exception as e:  # noqa: BLE001
                    result["parity_modes_error"] = f"{type(e).__name__}: {e}"[:300]
                    try:
                        set_compute(model, args.compute)
                    except Exception:  # noqa: BLE001
                        pass
            if not args.no_optimizer_leg:
                # the full training step of the reference's trainer (fwd + bwd + clip_grad_norm_ + AdamW + LR schedule,
                # trainer/query3d_trainer.py:18-28) in ONE HIP graph: pq3d_amd/trainer.py
                from pq3d_amd.trainer import TrainStep
                ts = TrainStep(model, lambda out: loss_fn(out, c["heads"]), lr=1e-4, grad_norm=80.0,
                               sched="warmup_cosine", warmup_steps=0, total_steps=10 ** 6)
                sside = torch.cuda.Stream()
                sside.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(sside):
                    for _ in range(2):
                        ts.step(dd)
                torch.cuda.current_stream().wait_stream(sside)
                torch.cuda.synchronize()
                g3 = None
                if not args.no_graph:
                    try:
                        g3 = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g3):
                            ts.step(dd)
                    except Exception as e:  # noqa: BLE001
                        print(f"[bench] train-step graph capture failed ({type(e).__name__}: {e})", file=sys.stderr)
                        torch.cuda.synchronize()
                        g3 = None
                run3 = g3.replay if g3 is not None else (lambda: ts.step(dd))
                for _ in range(args.warmup):
                    run3()
                ms3 = timed_loop(run3, args.steps)
                result["train_step_with_optimizer"] = {
                    "ms_per_step": ms3, "value": c["B"] / (ms3 * 1e-3), "unit": "scenes/s", "hip_graph": g3 is not None,
                    "optimizer": "AdamW(betas=(0.9,0.98), wd 0.01) + clip_grad_norm_(80) + warmup_cosine, 3 kernels on "
                                 "one flat fp32 buffer", "params": int(ts.flat_p.numel()),
                    "finite": bool(torch.isfinite(ts.flat_p).all())}
        if world == 1 and not dist_on and "generation" in c["heads"] and not args.headline_only:
            # the third-party body alone (fwd + bwd of the head on a detached query), eager
            gh = model.generation_head
            qd = torch.randn(c["B"], c["Nq"], c["d"], device=dev, requires_grad=True)

            def head_only():
                from pq3d_amd.losses import cross_entropy_rows
                cross_entropy_rows(gh(qd, dd["query_pad_masks"], dd["response"]), dd["response"]).backward()
            for _ in range(3):
                head_only()
            result["t5_body"] = {"ms_per_step_eager": timed_loop(head_only, max(3, args.steps // 5)),
                                 "impl": "T5-small decoder (random init, HF parameter layout) restated on the HIP kernels "
                                         "(pq3d_amd/t5.py) + input_proj; teacher-forced, T_r = %d" % c["Tr"],
                                 "params": sum(p.numel() for p in gh.parameters())}
        if world == 1 and not dist_on and args.cpu_steps > 0:
            hf_body = None
            if "generation" in c["heads"]:
                import copy
                hf_body = copy.deepcopy(model.generation_head.model).float().cpu().eval()   # eval: HF dropout off
            result["cpu_baseline"] = cpu_baseline(c, sd, dd_cpu, args.cpu_steps, 2, hf_body)
            if hf_body is not None:
                result["cpu_baseline"]["sample"] += "; caption body = stock HF T5 on the CPU (third-party, as the reference calls it)"
            result["speedup_vs_cpu_baseline"] = value / result["cpu_baseline"]["value"]
        result["chain_error"] = chain_err or bool(_ops.chain_error(dev))   # (the side legs above replay chain launches too)
        print(json.dumps(result))
    if dist_on:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if chain_err or bool(_ops.chain_error(dev)):
        print("[bench] a row-local chain launch timed out in a hand-off: the run is INVALID (see ops.chain_check)", file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
