"""Data-parallel gradient exchange (the only collective on the path): one process per GPU,
parameters replicated, scenes sharded, all-reduce(mean) of the parameter gradients over RCCL/xGMI.

In drop-in mode under the reference trainer, torch DDP (backend 'nccl' == RCCL on ROCm) wraps the module and
does this itself (trainer/build.py:66-75); this helper is what bench.py and a standalone harness use.  It keeps
ONE flat fp32 buffer per bucket: gradients are packed with a single multi-tensor copy, all-reduced in place
(one large message per bucket: xGMI is point-to-point, a few large messages beat many small ones) and the
parameters' .grad are re-pointed at views of the reduced buffer (no unpack copy)."""
from __future__ import annotations

from typing import Iterable, List

import torch
import torch.distributed as dist


class GradSlots(dict):
    """{id(param): (flat buffer, element offset, numel)} plus ``params`` = {id(param): param}: the slot map handed to the
    fused executor (``encoder.grad_arena``).  ``params`` lets the executor offer the slots of parameters OUTSIDE the decoder
    (the input encoders) to their own backward functions for the duration of one backward pass (ops.arena_*)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.params = {}


class NativeComm:
    """An RCCL communicator behind the C-ABI (``pq3d_comm_init`` / ``pq3d_allreduce_grads[_wire]``, csrc/comm.hip) --
    the exchange a host without torch.distributed binds; FlatGradAllReducer(comm=...) routes its buckets through it.  Replaces
    DDP's gradient all-reduce (reference trainer/build.py:66-75).  One per process, made on the CURRENT device; collective over
    all ranks.  Everything is stream-ordered on the caller's current stream."""

    def __init__(self, rank: int, world: int, unique_id: bytes):
        import ctypes as C
        from . import _lib
        assert len(unique_id) == 128
        self.rank, self.world = int(rank), int(world)
        self._h = C.c_void_p()
        self._scratch = {}
        _lib.check(_lib.lib().pq3d_comm_init(self.rank, self.world, C.c_char_p(unique_id), C.byref(self._h)), "pq3d_comm_init")

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C
        from . import _lib
        buf = C.create_string_buffer(128)
        _lib.check(_lib.lib().pq3d_comm_unique_id(buf), "pq3d_comm_unique_id")
        return buf.raw

    @classmethod
    def from_process_group(cls, group=None) -> "NativeComm":
        """Rendezvous through an initialised torch.distributed group (any backend): rank 0's id is broadcast as an object."""
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(rank, world, box[0])

    def rccl_version(self) -> int:
        import ctypes as C
        from . import _lib
        v = C.c_int32()
        _lib.check(_lib.lib().pq3d_comm_info(self._h, None, None, C.byref(v)), "pq3d_comm_info")
        return v.value

    def all_reduce(self, t: torch.Tensor, mean: bool = True, wire_bf16: bool = False) -> None:
        """In place on the current stream.  wire_bf16 (fp32 buffers): bf16 on the links, fp32 accumulation (comm.hip)."""
        from . import _lib
        assert t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, torch.bfloat16)
        L = _lib.lib()
        if wire_bf16:
            assert t.dtype == torch.float32
            need = L.pq3d_allreduce_wire_scratch_bytes(self.world, t.numel())
            sc = self._scratch.get(t.data_ptr())
            if sc is None or sc.numel() < need:     # fixed address per bucket: capturable
                sc = self._scratch[t.data_ptr()] = torch.empty(need, dtype=torch.uint8, device=t.device)
            _lib.check(L.pq3d_allreduce_grads_wire(self._h, t.data_ptr(), t.numel(), sc.data_ptr(), sc.numel(), int(mean), _lib.stream()),
                       "pq3d_allreduce_grads_wire")
        else:
            _lib.check(L.pq3d_allreduce_grads(self._h, t.data_ptr(), t.numel(), _lib.dt_of(t), int(mean), _lib.stream()),
                       "pq3d_allreduce_grads")

    def close(self) -> None:
        if self._h:
            from . import _lib
            torch.cuda.synchronize()
            _lib.check(_lib.lib().pq3d_comm_destroy(self._h), "pq3d_comm_destroy")
            self._h = None


class FlatGradAllReducer:
    """``groups``: explicit buckets (lists of parameters) in the order their gradients become final -- bench.py /
    TrainStep pass [decoder (+ mask head) parameters, everything else]: the fused decoder backward finishes ALL of the
    first bucket (93 % of the bytes at config 2) before the key/value input gradients and the encoders' backward run, so
    its all-reduce can be launched early on a side stream (``launch(0)``) and overlaps that tail."""

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 64 << 20, group=None,
                 keep_order: bool = False, groups=None, wire_dtype=None, comm: "NativeComm" = None):
        """comm: a NativeComm -- the buckets go through the C-ABI exchange (pq3d_allreduce_grads / _wire) instead of torch.distributed.
        wire_dtype = torch.bfloat16: the buckets cross the links as bf16 (half the bytes of the fp32 all-reduce: xGMI rings are
        per-link bound, config 2's 36.8 MB of fp32 gradients are ~0.3 ms of wire time at 8 ranks) with FP32 ACCUMULATION -- not a
        bf16 all-reduce, whose ring sums in bf16: an all-to-all hands every rank the bf16 pieces of ITS shard from all ranks, it sums
        them in fp32 (and divides), and an all-gather returns the bf16-rounded means.  Two roundings per element in total (each
        rank's contribution once, the mean once: ~4e-3 relative, the level the 'bf16' mode's gradients already have); every rank
        ends with bit-identical buffers.  None (default): fp32 all-reduce, DDP's arithmetic (trainer/build.py:66-75)."""
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = group
        self.comm = comm
        assert wire_dtype in (None, torch.bfloat16, torch.float32)
        self.wire_dtype = None if wire_dtype == torch.float32 else wire_dtype
        self._wire = {}          # bucket index -> (send, recv, shard, gathered) staging buffers (fixed addresses: capturable)
        if groups is not None:
            self.buckets = [[p for p in g if p.requires_grad] for g in groups]
            assert sorted(id(p) for b in self.buckets for p in b) == sorted(id(p) for p in self.params), \
                "groups must partition the parameters"
        else:
            self.buckets = [[]]
            size = 0
            # default: reverse registration order ~ order gradients become ready; keep_order: the caller's layout (the
            # flat optimizer wants parameter groups back to back)
            for p in (self.params if keep_order else reversed(self.params)):
                nb = p.numel() * 4
                if size + nb > bucket_bytes and self.buckets[-1]:
                    self.buckets.append([])
                    size = 0
                self.buckets[-1].append(p)
                size += nb
        self.buckets = [b for b in self.buckets if b]
        self.flat = [torch.zeros(sum(p.numel() for p in b), dtype=torch.float32, device=b[0].device) for b in self.buckets]
        self._side = None        # side stream of the early launches
        self._pending = []       # (work handle or None, bucket index)
        self._launched = set()
        # single-rank probe (bench.py PQ3D_BENCH_FORCE_DIST=1): issue the collectives even with ONE rank, so that the RCCL
        # code path (ReduceOp.AVG, side stream, capture inside a HIP graph) executes on a one-GPU box
        self.force_collectives = False

    def slots(self):
        """{id(param): (flat buffer, element offset, numel)} -- hand this to the fused executor
        (``encoder.grad_arena``): it then writes parameter gradients straight into the flat buffer (fresh view
        objects per backward, so autograd adopts them without a copy) and pack() skips them."""
        out = GradSlots()
        for flat, bucket in zip(self.flat, self.buckets):
            off = 0
            for p in bucket:
                n = p.numel()
                out[id(p)] = (flat, off, n)
                out.params[id(p)] = p
                off += n
        return out

    def pack(self, buckets=None) -> None:
        """Copy every .grad into the flat buffers (capturable: fixed addresses, one foreach copy per bucket).
        Gradients that already ARE views of the flat buffer (written in place by the fused executor) are skipped."""
        unused = []   # slots of parameters without a gradient this step: zeroed by ONE launch at the end
        prezeroed = set()
        if self.flat and self.flat[0].is_cuda:
            from . import ops
            ops.dw_deferred_flush()   # no reader of the slots gets ahead of queued weight-gradient products (ops._DwDeferred)
            prezeroed = ops.arena_zeroed_buffers()   # buffers a whole-pass gradient arena zero-filled for this very pass
        for bi, (flat, bucket) in enumerate(zip(self.flat, self.buckets)):
            if buckets is not None and bi not in buckets:
                continue
            if bi in self._launched:
                # its all-reduce is already in flight on the side stream (early launch from inside the backward: the
                # gradients were written in place): packing now would race with / overwrite the reduced data
                continue
            views, grads, off = [], [], 0
            for p in bucket:
                n = p.numel()
                v = flat[off:off + n].view_as(p)
                if p.grad is not None:
                    if p.grad.data_ptr() != v.data_ptr():
                        views.append(v)
                        grads.append(p.grad)
                elif flat.data_ptr() not in prezeroed:
                    unused.append(v)
                off += n
            if views:
                if flat.is_cuda and all(g.dtype == torch.float32 for g in grads):
                    from . import ops           # one launch of our own (torch's multi-tensor copy: 11 us for 60 small tensors)
                    ops.copy_many(views, grads)
                else:
                    torch._foreach_copy_(views, grads)
        if unused:
            if unused[0].is_cuda:
                from . import ops           # (an unused sub-module -- the caption model's encoder stack -- is 50 slots)
                for s0 in range(0, len(unused), 64):
                    ops.zero_many(unused[s0:s0 + 64])
            else:
                for v in unused:
                    v.zero_()

    # ---- collective -------------------------------------------------------------------------------------------------
    def _world(self) -> int:
        if self.comm is not None:
            return self.comm.world
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def _active(self) -> bool:
        return self._world() > 1 or (self.force_collectives and (self.comm is not None or dist.is_initialized()))

    def _reduce_any(self, bi: int):
        """(work handle or None, caller divides?) of bucket bi's exchange issued at the current stream's position."""
        if self.comm is not None:
            return self._reduce_native(bi), False
        if self.wire_dtype is not None:
            return self._reduce_wire(bi), False
        return self._reduce(self.flat[bi])

    def _reduce_native(self, bi: int) -> None:
        """Mean over ranks of bucket bi through the C-ABI communicator, in place, on the current stream."""
        self.comm.all_reduce(self.flat[bi], mean=True, wire_bf16=self.wire_dtype is not None)

    def _reduce(self, f: torch.Tensor):
        """mean over ranks, in place.  RCCL ('nccl') averages inside the collective (ReduceOp.AVG: no separate divide
        launch); gloo (CPU tests, the one-GPU two-rank hook) has no AVG -> SUM, the caller divides."""
        if dist.get_backend(self.group) == "nccl":
            return dist.all_reduce(f, op=dist.ReduceOp.AVG, group=self.group, async_op=True), False
        return dist.all_reduce(f, op=dist.ReduceOp.SUM, group=self.group, async_op=True), True

    def _reduce_wire(self, bi: int) -> None:
        """Mean over ranks of bucket bi with a bf16 wire format and fp32 accumulation, in place, on the CURRENT stream
        (stream-ordered collectives: with RCCL the host does not block)."""
        f, W, g = self.flat[bi], self._world(), self.group
        n = f.numel()
        per = -(-n // W)
        per += (-per) % 8          # 16-byte pieces
        buf = self._wire.get(bi)
        if buf is None or buf[0].numel() != W * per:
            mk = lambda *sh, dt: torch.zeros(*sh, dtype=dt, device=f.device)
            buf = self._wire[bi] = (mk(W * per, dt=self.wire_dtype), mk(W * per, dt=self.wire_dtype), mk(per, dt=self.wire_dtype),
                                    mk(W * per, dt=self.wire_dtype))
        send, recv, shard, gathered = buf
        send[:n].copy_(f)                                        # fp32 -> bf16 (the padding stays zero)
        if W > 1:
            dist.all_to_all_single(recv, send, group=g)          # piece r of every rank -> rank r
        else:
            recv.copy_(send)
        shard.copy_(recv.view(W, per).float().sum(0).div_(float(W)))   # fp32 accumulation, mean, one rounding
        if W > 1:
            dist.all_gather_into_tensor(gathered, shard, group=g)
        else:
            gathered.copy_(shard)
        f.copy_(gathered[:n])                                    # bf16 -> fp32: identical bits on every rank

    def launch(self, bi: int, after=None, side: bool = True) -> None:
        """Start the all-reduce of bucket ``bi`` on a side stream that waits for the work queued so far on the current
        stream -- or, with ``after`` (an event recorded on the current stream when the bucket became final), only for
        that point: work queued behind it then overlaps the transfer even though it was enqueued first.  The current
        stream carries on (the rest of the backward overlaps the transfer).  finish() joins.
        side=False: no side stream -- the collective is issued from the CURRENT stream's position (the backend's own
        communication stream picks up there) and only joined in finish(); work enqueued on the current stream AFTER this
        call overlaps it.  Two stream hops per bucket instead of four (each hop is a barrier packet + signal wait on this
        runtime): what a graph-replayed step uses between its pieces."""
        if not self._active() or bi in self._launched:
            return
        if self.flat[bi].is_cuda:
            from . import ops
            ops.dw_deferred_flush()   # weight gradients still queued for this pass (ops._DwDeferred) land before the bucket leaves
        cur = torch.cuda.current_stream() if self.flat[bi].is_cuda else None
        if cur is not None and torch.cuda.is_current_stream_capturing():
            # Inside a HIP-graph capture the collective stays ON the capturing stream (a synchronous op: RCCL's internal
            # stream forks from and rejoins this one).  Forking a side stream first -- cur -> side -> RCCL's stream -> side
            # -> cur -- makes hipStreamEndCapture SEGFAULT on this stack (torch 2.10 / ROCm 7.2, found with a one-rank
            # communicator on an MI355X: tools/probes/rccl_capture_probe.py); branches of one graph do not run
            # concurrently on this runtime anyway, so nothing is lost.
            if self.comm is not None:
                self._reduce_native(bi)
            elif self.wire_dtype is not None:
                self._reduce_wire(bi)
            elif dist.get_backend(self.group) == "nccl":
                dist.all_reduce(self.flat[bi], op=dist.ReduceOp.AVG, group=self.group)
            else:
                dist.all_reduce(self.flat[bi], op=dist.ReduceOp.SUM, group=self.group)
                self.flat[bi].div_(float(self._world()))
            self._launched.add(bi)
            return
        on_side = cur is not None and side
        if on_side:
            if self._side is None:
                self._side = torch.cuda.Stream()
            if after is not None:
                self._side.wait_event(after)
            else:
                self._side.wait_stream(cur)
            with torch.cuda.stream(self._side):
                h, div = self._reduce_any(bi)
        else:
            h, div = self._reduce_any(bi)
        self._pending.append((h, bi, div, on_side))
        self._launched.add(bi)

    def finish(self, side: bool = True) -> None:
        """Launch whatever has not been launched, wait for everything, leave the mean in the flat buffers."""
        if not self._active():
            return
        for bi in range(len(self.flat)):
            self.launch(bi, side=side)
        world = float(self._world())
        any_side = False
        for h, bi, div, on_side in self._pending:
            if self.flat[bi].is_cuda and on_side:
                any_side = True
                with torch.cuda.stream(self._side):
                    if h is not None:
                        h.wait()
                    if div:
                        self.flat[bi].div_(world)
            else:
                if h is not None:
                    h.wait()
                if div:
                    self.flat[bi].div_(world)
        if self._side is not None and self.flat[0].is_cuda and self._pending and any_side:
            # (nothing is pending on the side stream when the collectives ran in-stream inside a capture: no join, which a
            # capturing stream could not take from a stream outside the capture anyway)
            torch.cuda.current_stream().wait_stream(self._side)
        self._pending, self._launched = [], set()

    def all_reduce(self) -> None:
        self.finish()

    def unpack_views(self) -> None:
        """Point every .grad at its slice of the reduced flat buffer."""
        for flat, bucket in zip(self.flat, self.buckets):
            off = 0
            for p in bucket:
                n = p.numel()
                p.grad = flat[off:off + n].view_as(p)
                off += n

    def step(self) -> None:
        self.pack()
        self.all_reduce()
        self.unpack_views()
