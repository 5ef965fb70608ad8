"""bench.py --config pool: segment pooling as its own measured workload.

A "step" = the voxel -> segment pooling of one batch of B = 4 scenes at the 5 decoder levels of the sparse backbone
(channels 256, 256, 128, 96, 96; pcd_mask3d_encoder.py:142-150, res16unet.py:391), forward AND backward: the ids of the
batch are sorted once (pq3d_segment_plan), every level is reduced over that grouping through the composed fine -> coarse
index (the up-sampled [N, C] intermediate of the reference never exists), the gradient of the full-resolution level is a row
gather and the gradients of the 4 coarse levels are the same reduction regrouped by the coarse parent.  Inputs are resident
in HBM.  Beside the step, the plain torch_scatter.scatter_mean(feat [N, C], ids) -- every row really streamed from HBM --
is swept over N_vox in {5e4, 2.5e5} per scene and C in {96, 128, 256}; the roofline block is that kernel at N_vox = 2.5e5,
C = 256 on SURVEY 8d's algorithmic bytes N*C*4 + N*8 + S*C*4, timed live with HIP events on the launch stream."""
from __future__ import annotations

import json
import os
import time

import torch

LEVEL_C = (256, 256, 128, 96, 96)      # Res16UNet34C.PLANES[-5:]
PEAK_HBM_GBS = 8000.0


def synth_ids(B, n_vox, max_seg, seed):
    """Segment ids of B scenes (offset by b * max_seg): one floor-like segment with ~8 % of a scene's voxels, the rest with
    geometric sizes; voxel order is NOT sorted by segment (the voxelizer's order)."""
    g = torch.Generator().manual_seed(seed)
    ids = []
    for b in range(B):
        u = torch.rand(n_vox, generator=g)
        s = (u.pow(2.0) * (max_seg - 1)).long() + 1
        s[torch.rand(n_vox, generator=g) < 0.08] = 0
        ids.append(s.clamp_(max=max_seg - 1) + b * max_seg)
    return torch.cat(ids)


def synth_parents(B, n_vox, seed):
    """Fine voxel -> row of level h (h = 0..3: strides 16, 8, 4, 2): surfaces lose ~4x voxels per stride doubling.  Children of
    one coarse voxel are scattered over the fine order (voxel order is hash order in the sparse backend)."""
    g = torch.Generator().manual_seed(seed + 1)
    N = B * n_vox
    parents, n_coarse = [], []
    scramble = torch.randperm(N, generator=g)
    for h in range(4):
        f = 4 ** (4 - h)
        nc = (N + f - 1) // f
        parents.append((scramble // f).contiguous())
        n_coarse.append(nc)
    return parents, n_coarse


def _ev_time(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3     # us


def run(args, dev, world, rank, barrier, dist_on):
    from pq3d_amd import ops
    B, n_vox, max_seg = 4, int(args.pool_nvox), int(args.pool_segments)
    if getattr(args, "pool_pmc", False):
        # counter passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, tools/pool_profile.sh): ONLY the headline launch -- plain
        # scatter_mean forward at N = 4 x 250000, C = 256 -- and its gradient's gather, 5 times each, plus the calibration copy
        n_, s_ = B * 250_000, B * max_seg
        ids_ = synth_ids(B, 250_000, max_seg, 7).to(dev)
        pl = ops.SegmentPlan(ids_, s_)
        x, dy, cnt = torch.randn(n_, 256, device=dev), torch.randn(s_, 256, device=dev), torch.ones(s_, device=dev)
        for _ in range(5):
            pl.reduce(x, None, None, 256, True)
            ops.segment_gather(dy, ids_, cnt)
        n_cal = (100 << 20) // 4
        csrc, cdst = torch.ones(n_cal, device=dev), torch.empty(n_cal, device=dev)
        ops.copy_many([cdst], [csrc])
        torch.cuda.synchronize()
        return {"pool_pmc": True}
    N, S = B * n_vox, B * max_seg
    idx = synth_ids(B, n_vox, max_seg, 1234 + rank).to(dev)
    parents, n_coarse = synth_parents(B, n_vox, 1234 + rank)
    parents = [p.to(dev) for p in parents]
    g = torch.Generator().manual_seed(99 + rank)
    feats = [torch.randn(nc, c, generator=g).to(dev).requires_grad_(True) for nc, c in zip(n_coarse, LEVEL_C[:4])]
    feats.append(torch.randn(N, LEVEL_C[4], generator=g).to(dev).requires_grad_(True))
    douts = [torch.randn(S, c, generator=g).to(dev) for c in LEVEL_C]

    def step():
        plan = ops.SegmentPlan(idx, S)
        outs = [ops.upsample_scatter_mean(feats[h], parents[h], idx, S, plan=plan) for h in range(4)]
        outs.append(ops.scatter_mean(feats[4], idx, S, plan=plan))
        torch.autograd.backward(outs, douts)
        for f in feats:
            f.grad = None

    for _ in range(args.warmup):
        step()

    def timed_k():
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

    dts = [timed_k()]
    if dts[0] < 0.5 and args.min_time > 0:
        reps = int(min(args.max_repeats, max(2, -(-args.min_time // dts[0]))))
        if dist_on:
            r = torch.tensor([reps], device=dev, dtype=torch.int64)
            torch.distributed.broadcast(r, 0)
            reps = int(r.item())
        dts += [timed_k() for _ in range(reps - 1)]
    sd = sorted(dts)
    dt = sd[len(sd) // 2] if len(sd) % 2 else 0.5 * (sd[len(sd) // 2 - 1] + sd[len(sd) // 2])
    if rank != 0:
        return None

    # ---- per-kernel figures (HIP events on the launch stream), rank 0 ---------------------------------------------------
    def alg_bytes(n, s, c):
        return n * c * 4.0 + n * 8.0 + s * c * 4.0

    plan = ops.SegmentPlan(idx, S)
    t_plan = _ev_time(lambda: ops.SegmentPlan(idx, S), 20)
    levels = []
    for h in range(5):
        c = LEVEL_C[h]
        src = feats[h].detach()
        if h < 4:
            t_f = _ev_time(lambda: plan.reduce(src, parents[h], None, c, True), 20)
            pplan = plan.child(parents[h], n_coarse[h])
            inv = torch.ones(S, device=dev)
            t_b = _ev_time(lambda: pplan.reduce(douts[h], idx, inv, c, False, want_count=False), 20)
        else:
            t_f = _ev_time(lambda: plan.reduce(src, None, None, c, True), 20)
            cnt = torch.ones(S, device=dev)
            t_b = _ev_time(lambda: ops.segment_gather(douts[h], idx, cnt), 20)
        levels.append({"level": h, "C": c, "rows_read": int(src.shape[0]), "fwd_us": t_f, "bwd_us": t_b,
                       "fwd_alg_gbs": alg_bytes(N, S, c) / t_f / 1e3, "bwd_alg_gbs": alg_bytes(N, S, c) / t_b / 1e3})
    sweep = []
    roof = None
    bitexact = True
    for nv in (50_000, 250_000):
        n_, s_ = B * nv, S
        ids_ = synth_ids(B, nv, max_seg, 7).to(dev)
        pl = ops.SegmentPlan(ids_, s_)
        for c in (96, 128, 256):
            x = torch.randn(n_, c, device=dev)
            t_f = _ev_time(lambda: pl.reduce(x, None, None, c, True), 30)
            dy = torch.randn(s_, c, device=dev)
            cnt = torch.ones(s_, device=dev)
            t_b = _ev_time(lambda: ops.segment_gather(dy, ids_, cnt), 30)
            a, _ = pl.reduce(x, None, None, c, True)
            b2, _ = pl.reduce(x, None, None, c, True)
            bitexact = bitexact and bool(torch.equal(a, b2))
            row = {"n_vox_per_scene": nv, "N": n_, "S": s_, "C": c, "scatter_mean_fwd_us": t_f,
                   "fwd_alg_gbs": alg_bytes(n_, s_, c) / t_f / 1e3, "fwd_frac_hbm": alg_bytes(n_, s_, c) / t_f / 1e3 / PEAK_HBM_GBS,
                   "scatter_mean_bwd_us": t_b, "bwd_alg_gbs": alg_bytes(n_, s_, c) / t_b / 1e3,
                   "bwd_frac_hbm": alg_bytes(n_, s_, c) / t_b / 1e3 / PEAK_HBM_GBS}
            sweep.append(row)
            if nv == 250_000 and c == 256:
                roof = {"kernel": "segment_reduce_kernel<4,1> (pq3d_segment_reduce: scatter_mean forward, plain rows)",
                        "bound": "hbm", "achieved": row["fwd_alg_gbs"], "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": row["fwd_frac_hbm"], "traffic": None, "avg_launch_us": t_f,
                        "algorithmic_bytes_per_launch": alg_bytes(n_, s_, c),
                        "shape": f"N = {n_} voxels (B = {B} x {nv}), S = {s_} segments, C = {c} fp32",
                        "timing": "HIP events on the launch stream, 30 launches after 3 warm-up, this run",
                        "bit_identical_run_to_run": None}
            del x, dy
    roof["bit_identical_run_to_run"] = bitexact
    # measured memory-side traffic of that very launch: committed rocprofv3 --pmc passes (tools/pool_profile.sh), corrected by
    # the calibration copy of the same pass as MI355X_MICROARCH.md's HBM section prescribes
    import glob
    pmcs = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic_r*_pool.json")))
    if pmcs:
        try:
            pj = json.load(open(pmcs[-1]))
            rows = pj["kernels"].get("segment_reduce_kernel", [])
            cal = pj.get("calibration") or {}
            cf = 1.0 / cal["fetch_raw_over_true"] if cal.get("fetch_raw_over_true") else 2.0
            cw = 1.0 / cal["write_raw_over_true"] if cal.get("write_raw_over_true") else 1.0
            if rows:
                r0 = max(rows, key=lambda r: r["launches"])
                roof["traffic"] = (r0["fetch_kib"] * cf + r0["write_kib"] * cw) * 1024
                roof["traffic_over_algorithmic"] = roof["traffic"] / roof["algorithmic_bytes_per_launch"]
                roof["traffic_source"] = "profiles/" + os.path.basename(pmcs[-1])
                roof["traffic_correction"] = {"fetch_x": round(cf, 3), "write_x": round(cw, 3)}
        except (ValueError, KeyError):
            pass
    step_bytes = sum(2 * alg_bytes(N, S, c) for c in LEVEL_C)
    result = {
        "metric": "segment pooling fwd+bwd scenes/sec (5 backbone levels, B=4)", "value": B * world * args.steps / dt,
        "unit": "scenes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "repeats": len(dts), "ms_per_step_min": sd[0] / args.steps * 1e3, "ms_per_step_max": sd[-1] / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"segment pooling (SURVEY 8a row 15 / 8f-2): B={B} scenes/GPU x {n_vox} voxels, {max_seg} segments "
                               f"per scene, levels C={list(LEVEL_C)}, plan (sort) + 5 level means fwd + 5 gradients per step",
                   "global_batch": B * world, "parallelism": f"dp{world}", "hip_graph": False},
        "step_algorithmic_bytes": step_bytes, "step_alg_gbs": step_bytes / (dt / args.steps) / 1e9,
        "plan_us": t_plan, "levels": levels, "scatter_mean_sweep": sweep, "roofline": roof,
    }
    if args.cpu_steps > 0 and world == 1:
        from oracle import pq3d_oracle as O   # CPU baseline leg only: the oracle's restatement of torch_scatter's definition
        ncpu = os.cpu_count() or 1
        torch.set_num_threads(min(32, ncpu))
        idx_c = idx.cpu()
        par_c = [p.cpu() for p in parents]
        f_c = [f.detach().cpu().requires_grad_(True) for f in feats]
        d_c = [d.cpu() for d in douts]

        def cpu_step():
            t0 = time.perf_counter()
            outs = [O.multiscale_segment_pool(f_c[h], par_c[h], idx_c, S) for h in range(4)] + [O.scatter_mean(f_c[4], idx_c, S)]
            torch.autograd.backward(outs, d_c)
            for f in f_c:
                f.grad = None
            return time.perf_counter() - t0
        cpu_step()
        t1 = cpu_step()
        n = max(2, min(args.cpu_steps, int(15.0 / max(t1, 1e-3))))
        ts = sorted(cpu_step() for _ in range(n))
        med = ts[len(ts) // 2]
        result["cpu_baseline"] = {"value": B / med, "unit": "scenes/s", "cores": torch.get_num_threads(), "kind": "port",
                                  "ms_per_step": med * 1e3, "host_cpus": ncpu,
                                  "sample": f"{n} timed fwd+bwd steps (median) of the same {B}-scene batch, torch CPU index_add_ / "
                                            f"bincount / index restatement of torch_scatter + the materialised up-sampling"}
        result["speedup_vs_cpu_baseline"] = result["value"] / result["cpu_baseline"]["value"]
    return result
