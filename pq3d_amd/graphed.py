"""Drop-in speed without whole-step graph capture.

The reference trainer (trainer/query3d_trainer.py:30-45, trainer/build.py:66-75) calls ``out = model(data_dict)``,
builds the loss in Python and runs ``accelerator.backward(loss)`` under DDP -- it cannot replay one captured step.
Launched eagerly, a step of this package is ~100 dependent kernel launches whose descriptors are marshalled through
ctypes: host-bound at ~4x the device time.  ``GraphedQuery3D`` wraps a ``Query3DUnified`` so that its forward and its
backward are each ONE HIP-graph replay behind an ordinary autograd node: the loss, the optimizer and DDP's gradient
hooks stay eager and see ordinary ``.grad`` tensors; the model's launches cost two replays plus the copies of the batch
into the static input buffers.

Constraints (those of any captured graph): fixed input shapes / dtypes / key set (one wrapper per shape), no
data-dependent host control flow inside the model (true on this path: nothing synchronises), train / eval mode fixed at
wrap time.  Not usable for greedy generation (host-side token loop) -- eval-mode caption decoding keeps its own graph
(pq3d_amd/t5.py)."""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch
import torch.nn as nn

from . import ops

OUT_TENSORS = ("query_embeds", "ground_logits", "generation_logits", "generation_label")   # the label: long, no gradient
OUT_LISTS = ("predictions_class", "predictions_mask")


def _is_tensor_list(v) -> bool:
    return isinstance(v, (list, tuple)) and len(v) > 0 and all(torch.is_tensor(t) for t in v)


class _Flat(nn.Module):
    """Positional-tensor view of Query3DUnified.forward(data_dict) -> data_dict.  ``specs``: (key, None) for a tensor
    entry, (key, i) for element i of a list-of-tensors entry (multi-scale voxel pyramids / voxel2segment lists)."""

    def __init__(self, model: nn.Module, specs: Sequence[tuple], const: Dict[str, object]):
        super().__init__()
        self.model, self.specs, self.const = model, list(specs), dict(const)
        self.layout: List[tuple] = []     # filled by the first call: (key, count or None)

    def forward(self, *tensors):
        dd = dict(self.const)
        for (k, i), t in zip(self.specs, tensors):
            if i is None:
                dd[k] = t
            else:
                dd.setdefault(k, [])
                assert len(dd[k]) == i
                dd[k].append(t)
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
    """One autograd node for the whole model: forward = replay of the captured forward graph, backward = replay of a
    captured backward graph, which leaves every parameter gradient in the owner's flat buffers.
    inputs = the batch tensors followed (mode 'autograd') by the parameters."""

    @staticmethod
    def forward(ctx, owner, _anchor, n_in, *inputs):
        # `_anchor` is a 0-d leaf that requires grad: in 'direct' mode the parameters are not inputs of this node (their
        # gradients are left in the owner's buffers), so it is what makes autograd call backward() at all
        ctx.owner, ctx.n_in, ctx.n_par = owner, n_in, len(inputs) - n_in
        owner._copy_in(inputs[:n_in])
        owner.fwd_graph.replay()
        owner._fwd_token = ctx.token = object()   # the saved activations of THIS forward live in the graph's static memory
        return tuple(o.detach() for o in owner.static_out)

    @staticmethod
    def backward(ctx, *gouts):
        ow = ctx.owner
        if ctx.token is not ow._fwd_token:
            raise RuntimeError("GraphedQuery3D: backward of a forward whose saved activations were overwritten by a later "
                               "forward of the same wrapper (forward and backward are static graphs: run one backward per "
                               "forward, or wrap the evaluation passes in torch.no_grad())")
        gs = [g if g is not None else z for g, z in zip(gouts, ow.zero_gout)]
        torch._foreach_copy_(ow.static_gout, [gs[i] for i in ow.gout_idx])
        acc = ow._accumulating()
        pg = (None,) * ctx.n_par
        if ow.mode == "direct":
            ow._bwd_graph(acc).replay()
            ow._publish_grads()
        elif not acc:
            ow._bwd_graph(False).replay()
            # fresh view objects of the flat buffers: AccumulateGrad adopts them without a copy (and runs its hooks: DDP);
            # parameters the captured batch never reached get None, as from an eager backward
            pg = ow._fresh_grad_views()
        else:
            # accumulating micro-batch in 'autograd' mode: this micro-batch's gradients are formed FRESH in a second set of
            # flat buffers and handed to autograd, whose AccumulateGrad adds them onto .grad (= the views of the first set)
            # and runs its hooks -- DDP's bucket hooks fire on the last micro-batch of a no_sync window
            ow._bwd_graph("delta").replay()
            pg = ow._fresh_grad_views(delta=True)
        return (None, None, None) + tuple(ow.static_gin) + pg


class GraphedQuery3D(nn.Module):
    """``gm = GraphedQuery3D(model, sample_data_dict)``; then ``out = gm(data_dict)`` wherever ``model(data_dict)`` was
    called (same keys and shapes as the sample).  Parameters are the wrapped model's own (state_dict / optimizer see them
    through ``gm.model``).

    UNREACHED PARAMETERS ARE FIXED AT CAPTURE TIME: a parameter the captured sample batch's loss does not reach keeps
    ``.grad = None`` on every replay (both modes; under DDP's hook-driven 'autograd' mode this needs
    ``find_unused_parameters=True``).  A later batch that WOULD reach such a parameter (data-dependent head routing, a prompt
    type absent from the sample) replays the captured graph all the same -- its gradient for that parameter is not computed.
    Capture on a batch that exercises every branch the run will use, or rebuild the wrapper when the routing changes;
    ``unreached_parameter_names()`` lists what the capture left out so a caller can assert on it.

    Both modes replay the same two captured graphs; the captured backward writes the parameter gradients into two
    persistent flat fp32 buffers (the fused decoder in place, the rest with one multi-tensor copy).
    mode='direct' (default): ``.grad`` of every parameter is SET to its view of those buffers after the replay -- no
    autograd bookkeeping per parameter; autograd hooks on the parameters do not fire.
    mode='autograd': the parameters are inputs of the autograd node and receive fresh views of the flat buffers as their
    gradients: AccumulateGrad adopts a view without a copy when ``.grad`` is None and runs its hooks (DDP's bucket hooks
    fire, ``.grad`` is an ordinary tensor) -- the ~170 host-side AccumulateGrad visits overlap the backward replay on the
    device (round 2 used torch.cuda.make_graphed_callables here: one copy kernel per parameter per step, 3.5 ms at c2).
    Gradient accumulation (both modes; construct with ``accumulation=True``): a backward that finds the parameters'
    ``.grad`` still aliasing the flat buffers (no ``zero_grad(set_to_none=True)`` since the last one) accumulates -- torch
    semantics, the reference trains under accelerator.accumulate (trainer/query3d_trainer.py:35).  'direct' replays a
    variant of the backward graph that ADDS into the buffers; 'autograd' replays a variant that forms the micro-batch's
    gradients in a second set of buffers and returns those, so AccumulateGrad does the addition and its hooks (DDP) run on
    every micro-batch.  Parameters the captured batch never reaches (a bypassed sub-module) keep ``.grad = None`` in both
    modes, as after an eager backward (torch.optim.AdamW skips them)."""

    def __init__(self, model: nn.Module, sample: Dict[str, object], num_warmup_iters: int = 3, mode: str = "direct",
                 accumulation: bool = False, chain_check_every: int = 64):
        super().__init__()
        assert mode in ("direct", "autograd")
        # every `chain_check_every` replays the host reads the one-launch chains' error word (ops.chain_check: one stream
        # synchronisation; a hand-off that timed out raises instead of training on stale rows); 0 = never (call .check() yourself)
        self.chain_check_every, self._since_check = int(chain_check_every), 0
        assert num_warmup_iters >= 1, "GraphedQuery3D needs at least one eager warm-up iteration before capture"
        self.model, self.mode = model, mode
        specs, sample_tensors, const = [], [], {}
        for k, v in sample.items():
            if torch.is_tensor(v):
                specs.append((k, None)); sample_tensors.append(v)
            elif _is_tensor_list(v):
                for i, t in enumerate(v):
                    specs.append((k, i)); sample_tensors.append(t)
            else:
                if isinstance(v, (list, tuple, dict)) and any(torch.is_tensor(t) for t in (v.values() if isinstance(v, dict) else v)):
                    raise ValueError(f"GraphedQuery3D: data_dict[{k!r}] mixes tensors and non-tensors; only tensors and "
                                     "lists of tensors can be static graph inputs")
                const[k] = v
        self.specs, self._const = specs, const
        self._flat = _Flat(model, specs, const)
        args = tuple(t.detach().clone().requires_grad_(t.requires_grad) for t in sample_tensors)
        self._shapes = [(tuple(a.shape), a.dtype) for a in args]
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
            enc.grad_arena, enc.grad_arena_buffers = self.reducer.slots(), list(self.reducer.flat)
        self._params = params
        self._slots = self.reducer.slots()
        self._unused = set()      # ids of parameters without a gradient in the captured backward
        self._delta = None        # second set of flat buffers + slot map ('autograd' mode with accumulation)
        if mode == "autograd" and accumulation:
            self._delta = FlatGradAllReducer(params, groups=[g for g in groups if g])
            self._slots_delta = self._delta.slots()
        self._args = args
        self._gin_idx = [i for i, a in enumerate(args) if a.requires_grad]
        self._fwd_token = None

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(num_warmup_iters):
                outs = self._flat(*args)
                self.gout_idx = [i for i, o in enumerate(outs) if o.requires_grad]
                self.static_gout = [torch.zeros_like(outs[i]) for i in self.gout_idx]
                self._run_bwd(outs, False)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.fwd_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.fwd_graph):
            outs = self._flat(*args)
        self.zero_gout = [torch.zeros_like(o) for o in outs]
        self._bwd = {False: torch.cuda.CUDAGraph(), True: None, "delta": None}
        self.static_gin = None
        with torch.cuda.graph(self._bwd[False], pool=self.fwd_graph.pool()):
            gin = self._run_bwd(outs, False, retain=accumulation, record_unused=True)
        self.static_gin = [None] * len(args)
        for i, g in zip(self._gin_idx, gin):
            self.static_gin[i] = g
        if accumulation and mode == "direct":   # the accumulating variant, from the same (still alive) autograd graph
            self._bwd[True] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._bwd[True], pool=self.fwd_graph.pool()):
                self._run_bwd(outs, True, retain=False)
        elif accumulation:                      # 'autograd': the same backward, fresh, into the second set of buffers
            self._bwd["delta"] = torch.cuda.CUDAGraph()
            if enc is not None and groups[0]:
                enc.grad_arena, enc.grad_arena_buffers = self._slots_delta, list(self._delta.flat)
            try:
                with torch.cuda.graph(self._bwd["delta"], pool=self.fwd_graph.pool()):
                    self._run_bwd(outs, False, retain=False, delta=True)
            finally:
                if enc is not None and groups[0]:
                    enc.grad_arena, enc.grad_arena_buffers = self._slots, list(self.reducer.flat)
        # Release the captured forward's autograd graph: its AccumulateGrad nodes were created on the capture stream, and as
        # long as they live every eager backward of mode 'autograd' finds them (they are per-parameter singletons), sees a
        # stream mismatch and pays one hipEventRecord + hipStreamWaitEvent pair per parameter on the capture stream, behind
        # which the step's stream then stalls (config 2: 3.3 instead of 1.6 ms per step; config 4: 5.89 instead of 3.75 ms).
        # `del` alone is not enough: with a mask head the captured graph sits in a reference cycle (found in round 4 with
        # rocprofv3 --hip-runtime-trace, tools/probes/dropin_hiptrace2.sh: 175 waits per step on
        # torch.cuda.graph.default_capture_stream) and survives until Python's cycle collector happens to run.
        self.static_out = [o.detach() for o in outs]
        del outs, gin
        for p in params:
            p.grad = None
        import gc
        gc.collect()
        self._grad_views = None
        self._anchor = torch.zeros((), device=args[0].device, requires_grad=True)

    # -- backward body (captured twice: fresh / accumulating) ------------------------------------------------------------
    def _grad_view(self, p, delta: bool = False):
        flat, off, n = (self._slots_delta if delta else self._slots)[id(p)]
        return flat[off:off + n].view_as(p)

    def _fresh_grad_views(self, delta: bool = False):
        """One NEW view object per parameter (None for the unreached ones), in parameter order: what mode 'autograd' hands to
        AccumulateGrad, which adopts a gradient without a copy only if nobody else holds the object.  One as_strided call each
        from cached (buffer, shape, strides, offset) -- the slice + view_as pair and the dictionary look-ups were 0.3-0.5 ms
        of host time per step at 160 parameters, and the step of config 2 is host-bound in this mode on a slow host."""
        key = "_view_meta_delta" if delta else "_view_meta"
        meta = getattr(self, key, None)
        if meta is None:
            slots = self._slots_delta if delta else self._slots
            meta = []
            for p in self._params:
                if id(p) in self._unused:
                    meta.append(None)
                else:
                    flat, off, n = slots[id(p)]
                    meta.append((flat, tuple(p.shape), tuple(torch.empty(p.shape, device="meta").stride()), off))
            setattr(self, key, meta)
        ast = torch.as_strided
        return tuple(None if m is None else ast(m[0], m[1], m[2], m[3]) for m in meta)

    def _run_bwd(self, outs, accumulate: bool, retain: bool = False, record_unused: bool = False, delta: bool = False):
        """delta: write (fresh) into the second set of flat buffers; the input gradients still land in the first graph's."""
        params = self._params
        slots, flats = (self._slots_delta, self._delta.flat) if delta else (self._slots, self.reducer.flat)
        # the fused decoder decides "fresh step or accumulate" from its parameters' .grad (pq3d_amd/fused.py): alias the
        # flat buffers -> add in place; None -> zero, then write
        for p in params:
            p.grad = self._grad_view(p) if accumulate else None
        req = [o for o in outs if o.requires_grad]
        with ops.grad_arena(slots, flats) as arena:   # every slot offered for the whole pass (zeroed here when fresh)
            grads = torch.autograd.grad(req, [self._args[i] for i in self._gin_idx] + params, grad_outputs=self.static_gout,
                                        allow_unused=True, retain_graph=retain)
        gin, gp = grads[:len(self._gin_idx)], grads[len(self._gin_idx):]
        if not accumulate:
            arena.verify_returned(params, gp)   # tied weights: the returned gradient must still be the slot (ops.arena_verify)
        views, srcs = [], []
        for p, g in zip(params, gp):
            v = self._grad_view(p, delta)
            if g is None:
                if not accumulate:
                    # no gradient from autograd: either written in place by the fused executor in accumulate mode (never
                    # here) or genuinely unused -> zero in a fresh step, untouched when accumulating
                    v.zero_()
                    if record_unused:
                        self._unused.add(id(p))
            elif g.data_ptr() != v.data_ptr():
                views.append(v); srcs.append(g)
        if views:
            (torch._foreach_add_ if accumulate else torch._foreach_copy_)(views, srcs)
        if (accumulate or delta) and self.static_gin is not None:   # input gradients land in the buffers of the fresh graph
            dst = [self.static_gin[i] for i in self._gin_idx]
            if dst:
                torch._foreach_copy_(dst, list(gin))
        return gin

    def unreached_parameter_names(self):
        """Names of the parameters the captured sample batch's loss did not reach: their ``.grad`` stays None on every
        replay, whatever later batches contain (see the class docstring)."""
        names = {id(p): n for n, p in self.model.named_parameters()}
        return sorted(names.get(i, "<unnamed>") for i in self._unused)

    def _accumulating(self) -> bool:
        """True when every parameter's .grad still aliases its flat-buffer view (no zero_grad since the last backward)."""
        ptrs = getattr(self, "_view_ptrs", None)
        if ptrs is None:   # (parameter, address of its slot): cached -- a view per parameter per step was 0.3 ms of host time
            ptrs = self._view_ptrs = [(p, self._grad_view(p).data_ptr()) for p in self._params if id(p) not in self._unused]
        alias = [p.grad is not None and p.grad.data_ptr() == a for p, a in ptrs]
        if all(alias) and alias:
            return True
        if any(alias):
            raise RuntimeError("GraphedQuery3D: some parameters' .grad alias the flat gradient buffers and others do not -- "
                               "zero all gradients (set_to_none=True) or none between micro-batches")
        return False

    def _bwd_graph(self, accumulate):
        if self._bwd[accumulate] is None:
            raise RuntimeError("GraphedQuery3D: a backward found the parameters' .grad still set (gradient accumulation over "
                               "micro-batches) but the wrapper was built without the accumulating backward graph: construct "
                               "it with GraphedQuery3D(model, sample, accumulation=True), or zero the gradients "
                               "(set_to_none=True) between backward passes")
        return self._bwd[accumulate]

    # -- plumbing ------------------------------------------------------------------------------------------------------
    def _copy_in(self, inputs):
        by = {}
        for dst, src in zip(self.static_in, inputs):
            if dst.data_ptr() != src.data_ptr():
                by.setdefault(dst.dtype, ([], []))
                by[dst.dtype][0].append(dst); by[dst.dtype][1].append(src)
        for dsts, srcs in by.values():
            # the static inputs are saved tensors of the captured forward's autograd graph, which stays alive (the accumulating
            # backward variant is captured from it later): refreshing their contents must not advance their version counters
            with torch.autograd._unsafe_preserve_version_counter(tuple(dsts)):
                torch._foreach_copy_(dsts, srcs)

    def _publish_grads(self):
        if self._grad_views is None:
            self._grad_views = [None if id(p) in self._unused else self._grad_view(p) for p in self._params]
        for p, v in zip(self._params, self._grad_views):
            p.grad = v

    def check(self) -> None:
        """Raise ops.ChainHandoffError if a chain launch of the replays since the last check gave up in a hand-off (synchronises)."""
        from . import ops
        self._since_check = 0
        ops.chain_check(self.static_in[0].device if self.static_in else None)

    def forward(self, data_dict: Dict[str, object]) -> Dict[str, object]:
        self._since_check += 1
        if self.chain_check_every and self._since_check >= self.chain_check_every:
            self.check()
        args = []
        for (k, i), (shape, dtype) in zip(self.specs, self._shapes):
            t = data_dict[k] if i is None else data_dict[k][i]
            if tuple(t.shape) != shape or t.dtype != dtype:
                raise ValueError(f"GraphedQuery3D was captured with {k}{'' if i is None else [i]}: {shape} {dtype}, "
                                 f"got {tuple(t.shape)} {t.dtype}")
            args.append(t)
        for k, v in self._const.items():   # non-tensor entries were frozen into the graphs as constants: they must not change
            if k not in data_dict:
                raise ValueError(f"GraphedQuery3D: data_dict lacks {k!r} (present in the captured sample)")
            w = data_dict[k]
            try:
                same = (w is v) or bool(w == v)
            except Exception:  # noqa: BLE001  (objects without a usable ==: identity only)
                same = False
            if not same:
                raise ValueError(f"GraphedQuery3D: data_dict[{k!r}] = {w!r} differs from the captured constant {v!r}; "
                                 "build one wrapper per value")
        pars = tuple(self._params) if self.mode == "autograd" else ()
        flat = _Replay.apply(self, self._anchor, len(args), *args, *pars)
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
