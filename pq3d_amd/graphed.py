"""Drop-in speed without whole-step graph capture.

The reference trainer (trainer/query3d_trainer.py:30-45, trainer/build.py:66-75) calls ``out = model(data_dict)``,
builds the loss in Python and runs ``accelerator.backward(loss)`` under DDP -- it cannot replay one captured step.
Launched eagerly, a step of this package is ~130 dependent kernel launches whose descriptors are marshalled through
ctypes: host-bound at ~3.5x the device time.  ``GraphedQuery3D`` wraps a ``Query3DUnified`` so that its forward and its
backward are each ONE HIP-graph replay behind an ordinary autograd node (``torch.cuda.make_graphed_callables``): the
loss, the optimizer and DDP's gradient hooks stay eager and see ordinary ``.grad`` tensors, the model's ~130 launches
cost two replays plus the copies of the batch into the static input buffers.

Constraints (those of make_graphed_callables): fixed input shapes / dtypes / key set (one wrapper per shape), no
data-dependent host control flow inside the model (true on this path: nothing synchronises), train / eval mode fixed at
wrap time.  Not usable for greedy generation (host-side token loop) -- eval-mode caption decoding keeps its own graph
(pq3d_amd/t5.py)."""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch
import torch.nn as nn

OUT_TENSORS = ("query_embeds", "ground_logits", "generation_logits", "generation_label")   # the label: long, no gradient
OUT_LISTS = ("predictions_class", "predictions_mask")


class _Flat(nn.Module):
    """Positional-tensor view of Query3DUnified.forward(data_dict) -> data_dict."""

    def __init__(self, model: nn.Module, in_keys: Sequence[str], const: Dict[str, object]):
        super().__init__()
        self.model, self.in_keys, self.const = model, list(in_keys), dict(const)
        self.layout: List[tuple] = []     # filled by the first call: (key, count or None)

    def forward(self, *tensors):
        dd = dict(self.const)
        dd.update(zip(self.in_keys, tensors))
        out = self.model(dd)
        flat, layout = [], []
        for k in OUT_TENSORS:
            if k in out and torch.is_tensor(out[k]):
                flat.append(out[k]); layout.append((k, None))
        for k in OUT_LISTS:
            if k in out:
                flat.extend(out[k]); layout.append((k, len(out[k])))
        self.layout = layout
        return tuple(flat)


class _Replay(torch.autograd.Function):
    """One autograd node for the whole model: forward = replay of the captured forward graph, backward = replay of the
    captured backward graph, which leaves every parameter gradient in the owner's flat buffers."""

    @staticmethod
    def forward(ctx, owner, _anchor, *inputs):
        # `_anchor` is a 0-d leaf that requires grad: the parameters are not inputs of this node (their gradients are left
        # in the owner's buffers), so it is what makes autograd call backward() at all
        ctx.owner = owner
        owner._copy_in(inputs)
        owner.fwd_graph.replay()
        return tuple(o.detach() for o in owner.static_out)

    @staticmethod
    def backward(ctx, *gouts):
        ow = ctx.owner
        gs = [g if g is not None else z for g, z in zip(gouts, ow.zero_gout)]
        torch._foreach_copy_(ow.static_gout, [gs[i] for i in ow.gout_idx])
        ow.bwd_graph.replay()
        ow._publish_grads()
        return (None, None) + tuple(ow.static_gin)


class GraphedQuery3D(nn.Module):
    """``gm = GraphedQuery3D(model, sample_data_dict)``; then ``out = gm(data_dict)`` wherever ``model(data_dict)`` was
    called (same keys and shapes as the sample).  Parameters are the wrapped model's own (state_dict / optimizer see them
    through ``gm.model``).

    mode='direct' (default): the captured backward writes the parameter gradients straight into two persistent flat fp32
    buffers (the fused decoder in place, the rest with one multi-tensor copy) and ``.grad`` of every parameter is a view of
    them -- no per-parameter AccumulateGrad kernels (170 of them at config 2).  ``loss.backward()`` therefore OVERWRITES
    ``.grad`` (one backward per step; no gradient accumulation) and autograd hooks on the parameters do not fire.
    mode='autograd': ``torch.cuda.make_graphed_callables`` -- parameter gradients flow through AccumulateGrad as usual
    (DDP's hooks fire, accumulation works) at the price of one small copy kernel per parameter per step."""

    def __init__(self, model: nn.Module, sample: Dict[str, object], num_warmup_iters: int = 3, mode: str = "direct"):
        super().__init__()
        assert mode in ("direct", "autograd")
        self.model, self.mode = model, mode
        self.in_keys = [k for k, v in sample.items() if torch.is_tensor(v)]
        const = {k: v for k, v in sample.items() if not torch.is_tensor(v)}
        self._flat = _Flat(model, self.in_keys, const)
        args = tuple(sample[k].detach().clone().requires_grad_(sample[k].requires_grad) for k in self.in_keys)
        self._shapes = [(tuple(a.shape), a.dtype) for a in args]
        if mode == "autograd":
            self._graphed = torch.cuda.make_graphed_callables(self._flat, args, num_warmup_iters=num_warmup_iters,
                                                              allow_unused_input=True)
            return
        from .parallel import FlatGradAllReducer
        self.static_in = list(args)
        params = [p for p in model.parameters() if p.requires_grad]
        enc = getattr(model, "unified_encoder", None)
        dec_ids = {id(p) for p in enc.parameters()} if enc is not None else set()
        if hasattr(model, "mask_head"):
            dec_ids |= {id(p) for p in model.mask_head.parameters()}
        groups = [[p for p in params if id(p) in dec_ids], [p for p in params if id(p) not in dec_ids]]
        self.reducer = FlatGradAllReducer(params, groups=[g for g in groups if g])   # flat buffers + slot map (+ DP exchange)
        if enc is not None and groups[0]:
            enc.grad_arena, enc.grad_arena_buffers = self.reducer.slots(), [self.reducer.flat[0]]
        self._params = params
        gin_idx = [i for i, a in enumerate(args) if a.requires_grad]

        def run_bwd(outs):
            if enc is not None:
                enc.grad_arena_dirty = False
            req = [o for o in outs if o.requires_grad]
            grads = torch.autograd.grad(req, [args[i] for i in gin_idx] + params, grad_outputs=self.static_gout,
                                        allow_unused=True)
            gin, gp = grads[:len(gin_idx)], grads[len(gin_idx):]
            slots = self.reducer.slots()
            views, srcs = [], []
            for p, g in zip(params, gp):
                flat, off, n = slots[id(p)]
                v = flat[off:off + n].view_as(p)
                if g is None:
                    v.zero_()
                elif g.data_ptr() != v.data_ptr():
                    views.append(v); srcs.append(g)
            if views:
                torch._foreach_copy_(views, srcs)
            return gin

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(num_warmup_iters):
                outs = self._flat(*args)
                self.gout_idx = [i for i, o in enumerate(outs) if o.requires_grad]
                self.static_gout = [torch.zeros_like(outs[i]) for i in self.gout_idx]
                run_bwd(outs)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.fwd_graph, self.bwd_graph = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.fwd_graph):
            outs = self._flat(*args)
        self.static_out = list(outs)
        self.zero_gout = [torch.zeros_like(o) for o in outs]
        with torch.cuda.graph(self.bwd_graph, pool=self.fwd_graph.pool()):
            gin = run_bwd(outs)
        self.static_gin = [None] * len(args)
        for i, g in zip(gin_idx, gin):
            self.static_gin[i] = g
        self._grad_views = None
        self._anchor = torch.zeros((), device=args[0].device, requires_grad=True)

    # -- direct mode plumbing ------------------------------------------------------------------------------------------
    def _copy_in(self, inputs):
        by = {}
        for dst, src in zip(self.static_in, inputs):
            if dst.data_ptr() != src.data_ptr():
                by.setdefault(dst.dtype, ([], []))
                by[dst.dtype][0].append(dst); by[dst.dtype][1].append(src)
        for dsts, srcs in by.values():
            torch._foreach_copy_(dsts, srcs)

    def _publish_grads(self):
        if self._grad_views is None:
            slots = self.reducer.slots()
            self._grad_views = []
            for p in self._params:
                flat, off, n = slots[id(p)]
                self._grad_views.append(flat[off:off + n].view_as(p))
        for p, v in zip(self._params, self._grad_views):
            p.grad = v

    def forward(self, data_dict: Dict[str, object]) -> Dict[str, object]:
        args = []
        for k, (shape, dtype) in zip(self.in_keys, self._shapes):
            t = data_dict[k]
            if tuple(t.shape) != shape or t.dtype != dtype:
                raise ValueError(f"GraphedQuery3D was captured with {k}: {shape} {dtype}, got {tuple(t.shape)} {t.dtype}")
            args.append(t)
        flat = self._graphed(*args) if self.mode == "autograd" else _Replay.apply(self, self._anchor, *args)
        out = dict(data_dict)
        i = 0
        for k, n in self._flat.layout:
            if n is None:
                out[k] = flat[i]; i += 1
            else:
                out[k] = list(flat[i:i + n]); i += n
        if "ground_logits" in out:
            out["og3d_logits"] = out["ground_logits"]
        return out
