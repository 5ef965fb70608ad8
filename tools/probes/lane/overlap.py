"""Background lane: the decoder backward's deferred work on a second, CU-masked stream, under the chain of small launches.

Why: at config 2 the backward's query-side chain (40 dependent launches, 52-192 workgroups
each) takes 0.51 ms on a 256-CU chip, and 0.32 ms of weight-gradient / K-V gradient products that depend only on
per-layer intermediates used to run AFTER it.  Forked branches of one captured graph do not overlap on this runtime
(tools/probes/graph_branch_probe.py); two graphs on two streams do, when the background stream is CU-masked so that a
wide background launch cannot sit in front of every small chain launch (tools/probes/cumask_probe.py).

How: while the main step is captured (graph A), the fused backward (fused.py) hands its deferred launches to the lane
instead of issuing them -- ``lane.gate()`` publishes "everything launched so far is complete" with a one-thread kernel,
``lane.submit(fn)`` queues a closure behind the most recent gate -- and ``lane.join()`` makes graph A wait for the lane
to have executed everything submitted so far.  After graph A is captured the queued program is captured into graph B on
the lane's stream, sharing A's memory pool (the closures keep their operands alive, so the allocator cannot have reused
them inside A).  A step = replay B on the lane's stream + replay A on the main stream; the only ordering between the
two is the device-side flags (csrc/lane.hip), whose values are per-replay epochs, so nothing is reset between steps.

The reference's counterpart is DDP's "overlap what only feeds the optimizer with the rest of the backward"
(trainer/build.py:66-75); here the same idea is applied to the weight-gradient products themselves.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, List, Optional, Tuple

import torch

from . import _lib as L

_EPOCH_A, _EPOCH_B, _ERR, _FLAG0 = 0, 1, 2, 8
MAX_FLAGS = 120


def default_cus() -> int:
    """CUs of the background stream (multiple of 8: N/8 per XCD).  0 disables the lane."""
    return int(os.environ.get("PQ3D_BG_CUS", "128"))


class BackgroundLane:
    def __init__(self, device=None, cus: Optional[int] = None, timeout_us: int = 200_000):
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        cus = default_cus() if cus is None else int(cus)
        if cus <= 0 or cus % 8 or cus > 256:
            raise ValueError("BackgroundLane: cus must be a positive multiple of 8 (N/8 CUs per XCD), at most 256")
        self.cus, self.timeout_us = cus, int(timeout_us)
        words = (C.c_uint32 * 8)(*[sum((1 << b) for b in range(32) if 32 * w + b < cus) for w in range(8)])
        sp = C.c_void_p()
        with torch.cuda.device(self.device):
            L.check(L.lib().pq3d_lane_stream_create(words, 8, C.byref(sp)), "pq3d_lane_stream_create")
            self._raw = sp.value
            self.stream = torch.cuda.ExternalStream(self._raw, device=self.device)
            # [epoch A, epoch B, error count, ..., flags]: one 32-bit word each, on their own cache lines where it matters
            self.state = torch.zeros(_FLAG0 + MAX_FLAGS, dtype=torch.int32, device=self.device)
            # 100 MHz wall-clock stamps left by the lane kernels: row f = [signal time, wait begin, wait end] of flag f;
            # rows 0 / 1 column 0 = the epoch bumps of graph A / graph B (the start of each graph's replay)
            self.stamps = torch.zeros(_FLAG0 + MAX_FLAGS, 3, dtype=torch.int64, device=self.device)
        self.kinds: List[str] = []
        self.armed = False            # True while a main graph is being captured with this lane
        self.program: List[Tuple[str, object]] = []
        self._n = 0
        self._joined = True           # nothing submitted since the last join

    # ---- addresses
    def _p(self, i: int) -> C.c_void_p:
        return C.c_void_p(self.state.data_ptr() + 4 * i)

    def _ts(self, i: int) -> C.c_void_p:
        return C.c_void_p(self.stamps.data_ptr() + 24 * i)

    def _flag(self) -> int:
        if self._n >= MAX_FLAGS:
            raise RuntimeError("BackgroundLane: too many gates / joins in one step")
        self._n += 1
        return _FLAG0 + self._n - 1

    # ---- capture-time API (called on the MAIN stream while graph A is captured)
    def begin(self) -> None:
        self.program, self._n, self._joined, self.armed, self.kinds = [], 0, True, True, []
        L.check(L.lib().pq3d_lane_bump(self._p(_EPOCH_A), self._ts(_EPOCH_A), L.stream()), "pq3d_lane_bump")

    def gate(self) -> None:
        """Everything launched on the main stream so far is complete before what is submitted next starts."""
        f = self._flag()
        self.kinds.append("gate")
        L.check(L.lib().pq3d_lane_signal(self._p(f), self._p(_EPOCH_A), self._ts(f), L.stream()), "pq3d_lane_signal")
        self.program.append(("wait", f))

    def submit(self, fn: Callable[[], None]) -> None:
        self.program.append(("run", fn))
        self._joined = False

    def join(self) -> None:
        """The main stream waits until the lane has executed everything submitted so far."""
        if self._joined:
            return
        f = self._flag()
        self.kinds.append("join")
        self.program.append(("signal", f))
        L.check(L.lib().pq3d_lane_wait(self._p(f), self._p(_EPOCH_A), self.timeout_us, self._p(_ERR), self._ts(f), L.stream()),
                "pq3d_lane_wait")
        self._joined = True

    def end(self) -> None:
        self.join()
        self.armed = False

    # ---- graph B (called on the lane's stream, inside its capture)
    def run_program(self) -> None:
        L.check(L.lib().pq3d_lane_bump(self._p(_EPOCH_B), self._ts(_EPOCH_B), L.stream()), "pq3d_lane_bump")
        for op, arg in self.program:
            if op == "wait":
                L.check(L.lib().pq3d_lane_wait(self._p(arg), self._p(_EPOCH_B), self.timeout_us, self._p(_ERR), self._ts(arg),
                                               L.stream()), "pq3d_lane_wait")
            elif op == "signal":
                L.check(L.lib().pq3d_lane_signal(self._p(arg), self._p(_EPOCH_B), self._ts(arg), L.stream()), "pq3d_lane_signal")
            else:
                arg()

    def errors(self) -> int:
        """Number of pollers that timed out since creation (host sync).  Non-zero = a graph ran without its partner."""
        return int(self.state[_ERR].item())

    def timeline(self) -> List[dict]:
        """Hand-offs of the LAST replayed step in microseconds since graph A's start (host sync): for a gate, when the
        main graph published it, and when the lane began / stopped waiting for it (begin = the lane finished its previous
        chunk); for a join, when the lane published it and when the main graph began / stopped waiting."""
        st = self.stamps.cpu()
        t0 = int(st[_EPOCH_A, 0])
        us = lambda v: (int(v) - t0) / 100.0
        rows = [dict(kind="start", main_us=0.0, lane_us=us(st[_EPOCH_B, 0]))]
        for k, kind in enumerate(self.kinds):
            f = _FLAG0 + k
            rows.append(dict(kind=kind, published_us=us(st[f, 0]), wait_begin_us=us(st[f, 1]), wait_end_us=us(st[f, 2])))
        return rows

    def check(self) -> None:
        n = self.errors()
        if n:
            raise RuntimeError(f"BackgroundLane: {n} device-side waits timed out -- the two graphs of a step did not run "
                               "together (a serialising profiler? one graph replayed alone?); results are invalid")

    def __del__(self):
        try:
            if getattr(self, "_raw", None):
                torch.cuda.synchronize(self.device)
                L.lib().pq3d_lane_stream_destroy(C.c_void_p(self._raw))
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


class InlineLane:
    """Same capture-time interface, everything executed at once on the calling stream: the eager warm-up pass in front of
    a laned capture uses it so that every kernel variant the per-layer hand-overs launch has run once outside a capture."""
    armed = True

    def gate(self) -> None:
        pass

    def submit(self, fn: Callable[[], None]) -> None:
        fn()

    def join(self) -> None:
        pass


def current(owner) -> Optional[BackgroundLane]:
    """The armed lane attached to `owner` (a QueryMaskEncoder), or None."""
    lane = getattr(owner, "bg_lane", None)
    return lane if (lane is not None and lane.armed) else None


class LanedGraph:
    """``fn`` captured as graph A (main stream) + graph B (the lane's program).  ``owners``: objects whose ``bg_lane``
    attribute is set during the capture (the fused decoder looks there).  ``replay()`` launches both; a consumer on the
    calling stream is ordered behind A, and A ends with a join, so behind B as well."""

    def __init__(self, fn: Callable[[], None], lane: BackgroundLane, owners=(), main: Optional[torch.cuda.Stream] = None,
                 capture_error_mode: str = "thread_local"):
        self.lane = lane
        self.main = main if main is not None else torch.cuda.Stream(device=lane.device)
        self.gA, self.gB = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        cur = torch.cuda.current_stream(lane.device)
        self.main.wait_stream(cur)
        for o in owners:
            o.bg_lane = lane
        try:
            with torch.cuda.stream(self.main):
                self.gA.capture_begin(capture_error_mode=capture_error_mode)
                try:
                    lane.begin()
                    fn()
                    lane.end()
                finally:
                    lane.armed = False
                    self.gA.capture_end()
            lane.stream.wait_stream(self.main)
            with torch.cuda.stream(lane.stream):
                self.gB.capture_begin(pool=self.gA.pool(), capture_error_mode=capture_error_mode)
                try:
                    lane.run_program()
                finally:
                    self.gB.capture_end()
        finally:
            for o in owners:
                o.bg_lane = None
            self.n_background = sum(1 for op, _ in lane.program if op == "run")
            lane.program = []       # drop the closures (and with them the references to graph A's intermediates)
        cur.wait_stream(self.main)
        cur.wait_stream(lane.stream)

    def replay(self) -> None:
        cur = torch.cuda.current_stream(self.lane.device)
        same = cur.cuda_stream == self.main.cuda_stream
        if not same:
            self.main.wait_stream(cur)
        with torch.cuda.stream(self.lane.stream):
            self.gB.replay()
        with torch.cuda.stream(self.main):
            self.gA.replay()
        if not same:
            cur.wait_stream(self.main)

    __call__ = replay
