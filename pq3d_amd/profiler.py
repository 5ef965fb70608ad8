"""Opt-in per-kernel-family timing with HIP events on the launch stream (bench.py's roofline section).

`with KernelTimer() as t:` makes every C-ABI call issued through pq3d_amd._lib record an event pair on torch's
current stream (the stream the kernels are launched on); `t.summary()` synchronises once and aggregates
(calls, total ms, algorithmic FLOPs / bytes) per (entry point, shape key)."""
from __future__ import annotations

from collections import defaultdict
from typing import Dict, List, Tuple

import torch

_active: "KernelTimer | None" = None


class KernelTimer:
    def __init__(self):
        self.records: List[Tuple[str, str, float, float, torch.cuda.Event, torch.cuda.Event]] = []

    def __enter__(self):
        global _active
        _active = self
        return self

    def __exit__(self, *exc):
        global _active
        _active = None

    def summary(self) -> Dict[Tuple[str, str], dict]:
        torch.cuda.synchronize()
        agg: Dict[Tuple[str, str], dict] = defaultdict(lambda: dict(calls=0, ms=0.0, flops=0.0, bytes=0.0))
        for name, key, flops, nbytes, e0, e1 in self.records:
            a = agg[(name, key)]
            a["calls"] += 1
            a["ms"] += e0.elapsed_time(e1)
            a["flops"] += flops
            a["bytes"] += nbytes
        return dict(agg)


def timed(name: str, key: str, flops: float, nbytes: float, fn, *args):
    """Run fn(*args); when a KernelTimer is active bracket it with events on the current stream."""
    t = _active
    if t is None:
        return fn(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn(*args)
    e1.record()
    t.records.append((name, key, flops, nbytes, e0, e1))
    return rc
