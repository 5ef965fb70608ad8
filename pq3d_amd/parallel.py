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


class FlatGradAllReducer:
    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 64 << 20, group=None,
                 keep_order: bool = False):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = group
        self.buckets: List[List[torch.nn.Parameter]] = [[]]
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
        self.flat = [torch.zeros(sum(p.numel() for p in b), dtype=torch.float32, device=b[0].device)
                     for b in self.buckets if b]

    def slots(self):
        """{id(param): (flat buffer, element offset, numel)} -- hand this to the fused executor
        (``encoder.grad_arena``): it then writes parameter gradients straight into the flat buffer (fresh view
        objects per backward, so autograd adopts them without a copy) and pack() skips them."""
        out = {}
        for flat, bucket in zip(self.flat, self.buckets):
            off = 0
            for p in bucket:
                n = p.numel()
                out[id(p)] = (flat, off, n)
                off += n
        return out

    def pack(self) -> None:
        """Copy every .grad into the flat buffers (capturable: fixed addresses, one foreach copy per bucket).
        Gradients that already ARE views of the flat buffer (written in place by the fused executor) are skipped."""
        for flat, bucket in zip(self.flat, self.buckets):
            views, grads, off = [], [], 0
            for p in bucket:
                n = p.numel()
                v = flat[off:off + n].view_as(p)
                if p.grad is not None:
                    if p.grad.data_ptr() != v.data_ptr():
                        views.append(v)
                        grads.append(p.grad)
                else:
                    v.zero_()
                off += n
            if views:
                torch._foreach_copy_(views, grads)

    def all_reduce(self) -> None:
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if world == 1:
            return
        handles = [dist.all_reduce(f, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for f in self.flat]
        for h in handles:
            h.wait()
        torch._foreach_div_(self.flat, float(world))

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
