"""The training step around the decoder for the HIP model: what ``Query3DTrainer.train_step`` /
``backward`` do per batch (trainer/query3d_trainer.py:18-45) --

    model.train(); out = model(data_dict); loss = Loss(out); zero_grad; backward [DDP all-reduce(mean)];
    clip_grad_norm_(grad_norm); AdamW.step(); LambdaLR.step()

with the optimizer side on ONE flat fp32 parameter/gradient buffer and three kernels (pq3d_sumsq_partials,
pq3d_train_scalars, pq3d_adamw): no per-parameter launches, no host synchronisation (the step counter, learning
rate, bias corrections and clip coefficient live on the device), so the whole step is HIP-graph capturable.

Parameter groups follow ``model.get_opt_params()`` (query3d_unified.py:224-238 -> optim/utils.py:1-18): weight decay
0.01 except names containing 'bias' / 'LayerNorm.bias' / 'LayerNorm.weight' (NB the decoder's norms are called
``norm`` so their *weights* are decayed -- reference behaviour, reproduced), per-module learning rates.
The parameters stay ordinary ``nn.Parameter`` objects (state_dict / checkpoint compatible); their storage is moved
into the flat buffer (``p.data`` becomes a view), and the fused executor writes gradients straight into the matching
flat gradient buffer, which is also what the data-parallel all-reduce operates on."""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, Optional, Sequence

import torch

from . import _lib as L
from . import ops
from .parallel import FlatGradAllReducer


class TrainStep:
    def __init__(self, model: torch.nn.Module, loss_fn: Callable[[dict], torch.Tensor], *,
                 opt_groups: Optional[Sequence[dict]] = None, lr: float = 1e-4, betas=(0.9, 0.98), eps: float = 1e-8,
                 grad_norm: Optional[float] = None, sched: str = "warmup_cosine", warmup_steps: int = 0,
                 total_steps: int = 1, sched_gamma: float = 1.0, num_gpu: int = 1, group=None,
                 unused_parameters=None, chain_check_every: int = 64):
        # every `chain_check_every` eager steps the host reads the one-launch chains' error word (ops.chain_check: one stream
        # synchronisation per N steps; a timed-out hand-off raises instead of stepping on stale rows).  A step replayed from a
        # HIP graph runs no host code: call .check() next to wherever the loss is read.  0 = never.
        self.chain_check_every, self._since_check = int(chain_check_every), 0
        groups = list(opt_groups) if opt_groups is not None else model.get_opt_params()
        self.model, self.loss_fn, self.group = model, loss_fn, group
        self.hp = L.AdamWHp()
        self.hp.lr, self.hp.beta1, self.hp.beta2, self.hp.eps = lr, betas[0], betas[1], eps
        self.hp.max_grad_norm = float(grad_norm) if grad_norm else 0.0
        self.hp.sched = L.SCHED[sched]
        # optim/scheduler.py:20 scales the warm-up by num_gpu because accelerate's prepared scheduler (trainer/build.py:123)
        # advances the LambdaLR num_gpu times per optimizer step: factor of optimizer step k = lambda(k * num_gpu)
        self.hp.warmup_steps = int(warmup_steps) * int(num_gpu)
        self.hp.sched_stride = int(num_gpu)
        self.hp.total_steps, self.hp.sched_gamma = int(total_steps), float(sched_gamma)
        # torch.optim.AdamW (the reference's optimizer) skips a parameter whose .grad is None: no decay, no moment update.
        # The flat layout is DETERMINISTIC -- every parameter of the groups except the explicit never-used set
        # (``model.unused_parameters()``: the bypassed T5 encoder of the generation head, plus the caller's
        # ``unused_parameters``) -- so ranks, checkpoints and graph captures always agree on it.  A parameter that stays in
        # the buffer but has no gradient in some step (a head the batch did not reach) is SKIPPED for that step by the
        # AdamW kernel (a segment with lr_mul < 0: parameter and moments untouched), exactly like torch; see
        # optimizer_step.  Under data parallelism (world > 1) presence is a per-rank fact, so every parameter of the
        # layout is updated (zero gradient where a rank had none, averaged like DDP's find_unused_parameters buckets):
        # a parameter NO rank reaches is then still decayed -- list it in ``unused_parameters`` (documented divergence).
        # Second documented divergence: the bias corrections use the ONE global step count, torch.optim.AdamW a per-parameter
        # count that only advances when the parameter has a gradient -- a head that is skipped in some steps gets
        # 1 - beta^t with the global t instead of its own (identical as soon as every parameter is reached every step, which
        # is the case for the reference's stage-1 / stage-2 trainers; tests/test_gpu_trainer.py pins that case).
        unused = {id(p) for p in (model.unused_parameters() if hasattr(model, "unused_parameters") else ())}
        unused |= {id(p) for p in (unused_parameters or ())}
        self._names = {id(p): n for n, p in model.named_parameters()}
        self._seg_cache: Dict[frozenset, "L.OptSegments"] = {}
        self._layout([dict(g, params=[p for p in g["params"] if id(p) not in unused]) for g in groups])

    def _layout(self, groups) -> None:
        """(Re)build the flat parameter / gradient / moment buffers over ``groups``; parameters keep their values."""
        lr = self.hp.lr
        groups = [g for g in groups if len(g["params"])]
        if len(groups) > L.MAX_OPT_SEGMENTS:   # merge groups with equal (lr, weight_decay), keeping first-seen order
            merged: Dict[tuple, dict] = {}
            for g in groups:
                merged.setdefault((g["lr"], g["weight_decay"]), {"params": [], "lr": g["lr"],
                                                                  "weight_decay": g["weight_decay"]})["params"] += g["params"]
            groups = list(merged.values())
        assert len(groups) <= L.MAX_OPT_SEGMENTS, "too many distinct (lr, weight_decay) parameter groups"
        self.groups = groups
        params = [p for g in groups for p in g["params"]]
        assert len({id(p) for p in params}) == len(params), "a parameter appears in two groups"
        dev = params[0].device
        self.reducer = FlatGradAllReducer(params, bucket_bytes=1 << 62, group=self.group, keep_order=True)
        self.flat_g = self.reducer.flat[0]
        n = self.flat_g.numel()
        flat_p = torch.empty(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.segs = L.OptSegments()
        self.segs.n = len(groups)
        off = 0
        for s, g in enumerate(groups):
            for p in g["params"]:
                k = p.numel()
                flat_p[off:off + k].copy_(p.data.reshape(-1))
                p.data = flat_p[off:off + k].view(p.shape)     # parameter storage now lives in the flat buffer
                off += k
            self.segs.end[s] = off
            self.segs.lr_mul[s] = g["lr"] / lr
            self.segs.weight_decay[s] = g["weight_decay"]
        self.flat_p = flat_p
        self._seg_cache = {}
        if not hasattr(self, "step_count"):
            self.step_count = torch.zeros(1, dtype=torch.int64, device=dev)
            self.partials = torch.zeros(L.SUMSQ_PARTIALS, dtype=torch.float32, device=dev)
            self.scalars = torch.zeros(8, dtype=torch.float32, device=dev)   # lr, lr/bc1, 1/sqrt(bc2), clip, |g|, -
        # the fused decoder writes parameter gradients straight into flat_g (no pack copy)
        enc = getattr(self.model, "unified_encoder", None)
        if enc is not None:
            enc.grad_arena = self.reducer.slots()
            enc.grad_arena_buffers = self.reducer.flat

    # -- pieces (each capturable) ---------------------------------------------------------------------------------
    def forward_backward(self, data_dict: dict, accumulate: bool = False, loss_scale: float = 1.0) -> torch.Tensor:
        """One micro-batch: forward, loss, backward, pack.  accumulate=False starts a fresh optimizer step (gradients set
        to None first); accumulate=True ADDS this micro-batch's gradients to what the previous calls left (the reference
        trains under accelerator.accumulate, trainer/query3d_trainer.py:35): the fused decoder sees its parameters' .grad
        still aliasing the flat arena and accumulates in place, everything else accumulates through autograd."""
        if not accumulate:
            self.model.zero_grad(set_to_none=True)
        out = self.model(dict(data_dict))
        loss = self.loss_fn(out)
        if loss_scale != 1.0:
            loss = loss * loss_scale
        enc = getattr(self.model, "unified_encoder", None)
        if enc is not None and getattr(enc, "grad_arena", None) is not None:
            with ops.grad_arena(enc.grad_arena, enc.grad_arena_buffers, pack_follows=True):   # every slot offered for the whole pass
                loss.backward()
        else:
            loss.backward()
        self.reducer.pack()
        return loss.detach()

    def all_reduce(self) -> None:
        self.reducer.all_reduce()     # mean over ranks (DDP semantics); no-op at world size 1

    def _world(self) -> int:
        import torch.distributed as dist
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def _segments(self, missing: frozenset) -> "L.OptSegments":
        """Segment table of the flat buffer with the parameters in ``missing`` (ids; no gradient this step) marked
        skip (lr_mul < 0).  Cached per missing set; the empty set is the plain per-group table."""
        if not missing:
            return self.segs
        sg = self._seg_cache.get(missing)
        if sg is not None:
            return sg
        runs, off = [], 0     # [end, lr_mul, weight_decay]
        for g in self.groups:
            for p in g["params"]:
                off += p.numel()
                key = (-1.0, 0.0) if id(p) in missing else (g["lr"] / self.hp.lr, g["weight_decay"])
                if runs and (runs[-1][1], runs[-1][2]) == key:
                    runs[-1][0] = off
                else:
                    runs.append([off, key[0], key[1]])
        if len(runs) > L.MAX_OPT_SEGMENTS:
            names = [self._names.get(i, "?") for i in list(missing)[:8]]
            raise RuntimeError(f"TrainStep: {len(missing)} parameters without a gradient split the flat buffer into {len(runs)} "
                               f"segments (> {L.MAX_OPT_SEGMENTS}); pass the never-used ones as unused_parameters= "
                               f"(e.g. {names})")
        sg = L.OptSegments()
        sg.n = len(runs)
        for i, (end, lm, wd) in enumerate(runs):
            sg.end[i], sg.lr_mul[i], sg.weight_decay[i] = end, lm, wd
        self._seg_cache[missing] = sg
        return sg

    def optimizer_step(self) -> None:
        n, s = self.flat_g.numel(), L.stream()
        # parameters without a gradient in THIS step (a host-side fact of the autograd graph, also under graph capture):
        # skipped like torch.optim.AdamW does; their flat gradient is zero (pack), so the clip norm ignores them too
        missing = frozenset(id(p) for p in self.reducer.params if p.grad is None) if self._world() == 1 else frozenset()
        segs = self._segments(missing)
        L.check(L.lib().pq3d_sumsq_partials(L.ptr(self.flat_g), n, L.ptr(self.partials), s), "pq3d_sumsq_partials")
        L.check(L.lib().pq3d_train_scalars(C.byref(self.hp), L.ptr(self.step_count), L.ptr(self.partials),
                                           L.ptr(self.scalars), s), "pq3d_train_scalars")
        L.check(L.lib().pq3d_adamw(L.ptr(self.flat_p), L.ptr(self.flat_g), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq),
                                   n, C.byref(self.hp), C.byref(segs), L.ptr(self.scalars), s), "pq3d_adamw")

    def step(self, data_dict) -> torch.Tensor:
        """One full training step; returns the (detached) loss.  grad norm / lr of the step: ``self.scalars``.
        ``data_dict`` may be a list of micro-batches (gradient accumulation: the losses are scaled by 1/len so that the
        step equals one step on the concatenated batch for mean-reduced losses, as accelerate does)."""
        if isinstance(data_dict, (list, tuple)):
            k = len(data_dict)
            loss = None
            for i, mb in enumerate(data_dict):
                l = self.forward_backward(mb, accumulate=i > 0, loss_scale=1.0 / k)
                loss = l if loss is None else loss + l
        else:
            loss = self.forward_backward(data_dict)
        self.all_reduce()
        self.optimizer_step()
        self._since_check += 1
        if self.chain_check_every and self._since_check >= self.chain_check_every and not torch.cuda.is_current_stream_capturing():
            self.check()
        return loss

    def check(self) -> None:
        """Raise ops.ChainHandoffError if a chain launch since the last check gave up in a hand-off (synchronises the stream)."""
        self._since_check = 0
        ops.chain_check(self.flat_p.device)

    # -- introspection ---------------------------------------------------------------------------------------------
    @property
    def last_grad_norm(self) -> torch.Tensor:
        return self.scalars[4]

    @property
    def last_lr(self) -> torch.Tensor:
        return self.scalars[0]

    def layout(self):
        """[(parameter name, numel)] in flat-buffer order: part of the checkpoint, checked on load."""
        return [(self._names.get(id(p), f"<unnamed {i}>"), p.numel())
                for i, p in enumerate(p for g in self.groups for p in g["params"])]

    def state_dict(self) -> dict:
        return {"step": self.step_count.clone(), "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "layout": self.layout()}

    def load_state_dict(self, sd: dict) -> None:
        have = self.layout()
        if "layout" not in sd:
            # a checkpoint of the former (probed) layout: the buffers cannot be matched to parameters by position
            if sd["exp_avg"].numel() != self.exp_avg.numel():
                raise ValueError(f"TrainStep.load_state_dict: checkpoint without a 'layout' entry holds {sd['exp_avg'].numel()} "
                                 f"moment elements, this TrainStep {self.exp_avg.numel()}: rebuild it with the parameter set it "
                                 "was saved with (older checkpoints covered only the parameters that received gradients)")
        want = [tuple(x) for x in sd.get("layout", have)]
        if want != [tuple(x) for x in have]:
            diff = [a for a, b in zip(want, have) if a != b][:4]
            raise ValueError(f"TrainStep.load_state_dict: the checkpoint's flat layout ({len(want)} parameters, "
                             f"{sum(n for _, n in want)} elements) differs from this TrainStep's ({len(have)}, "
                             f"{sum(n for _, n in have)}); first differences {diff} -- build the TrainStep with the same "
                             f"parameter groups / unused_parameters")
        self.step_count.copy_(sd["step"]); self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])
