#!/usr/bin/env python3
"""Timing of the set criterion at stage-1-like sizes: HIP path (pq3d_amd/losses.py; device part + the
host LSA it shares with the reference) vs the CPU oracle of the reference's loop.  Prints one JSON line.
    python tools/bench_criterion.py [--B 4 --Ns 4096 --Nq 200 --layers 13 --inst 60]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from pq3d_amd import synth
from pq3d_amd.losses import HungarianMatcher, SetCriterion


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=4); ap.add_argument("--Ns", type=int, default=4096)
    ap.add_argument("--Nq", type=int, default=200); ap.add_argument("--layers", type=int, default=13)
    ap.add_argument("--inst", type=int, default=60); ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--cpu-steps", type=int, default=2)
    a = ap.parse_args()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    r = np.random.default_rng(0)
    seg_len = [a.Ns] + [int(x) for x in r.integers(a.Ns // 2, a.Ns, a.B - 1)]
    n_inst = [int(x) for x in r.integers(a.inst // 2, a.inst + 1, a.B)]
    masks, logits, labels, seg = synth.criterion_inputs(seed=1, B=a.B, Ns=a.Ns, Nq=a.Nq, C=201, n_layers=a.layers,
                                                        seg_len=seg_len, n_inst=n_inst)
    W = dict(cost_class=2.0, cost_mask=5.0, cost_dice=2.0)
    crit = SetCriterion(num_classes=200, matcher=HungarianMatcher(num_points=-1, **W), weight_dict={}, losses=["labels", "masks"],
                        num_points=-1, class_weights=-1, ignore_label=-100)
    dm = [m.cuda().requires_grad_(True) for m in masks]
    dl = [l.cuda().requires_grad_(True) for l in logits]

    def step():
        for t in dm + dl:
            t.grad = None
        losses, _ = crit(dm, dl, labels, seg)
        sum(losses.values()).backward()

    for _ in range(2):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize(); gpu_ms = (time.perf_counter() - t0) / a.steps * 1e3
    # host share: the LSA alone on the same cost matrices
    from scipy.optimize import linear_sum_assignment
    host = [np.random.default_rng(i).random((a.Nq, n_inst[i % a.B])).astype(np.float32) for i in range(a.layers * a.B)]
    t0 = time.perf_counter()
    for c in host:
        linear_sum_assignment(c)
    lsa_ms = (time.perf_counter() - t0) * 1e3
    res = {"workload": f"set criterion, {a.layers} prediction layers x B={a.B} scenes, Ns={a.Ns}, Nq={a.Nq}, C=201, "
                       f"targets/scene {n_inst}", "hip_ms_per_step_fwd_bwd": gpu_ms,
           "host_lsa_ms_random_costs": lsa_ms, "cpu_threads": torch.get_num_threads()}
    if a.cpu_steps:
        from oracle import loss_oracle as LO   # CPU baseline leg only
        om = [m.clone().requires_grad_(True) for m in masks]
        ol = [l.clone().requires_grad_(True) for l in logits]

        def cpu_step():
            for t in om + ol:
                t.grad = None
            losses, _ = LO.set_criterion(om, ol, labels, seg, num_classes=200, **W)
            sum(losses.values()).backward()
        cpu_step()
        t0 = time.perf_counter()
        for _ in range(a.cpu_steps):
            cpu_step()
        res["cpu_oracle_ms_per_step_fwd_bwd"] = (time.perf_counter() - t0) / a.cpu_steps * 1e3
        res["speedup_vs_cpu_oracle"] = res["cpu_oracle_ms_per_step_fwd_bwd"] / gpu_ms
    print(json.dumps(res))


if __name__ == "__main__":
    main()
