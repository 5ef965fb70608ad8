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
from .parallel import FlatGradAllReducer


class TrainStep:
    def __init__(self, model: torch.nn.Module, loss_fn: Callable[[dict], torch.Tensor], *,
                 opt_groups: Optional[Sequence[dict]] = None, lr: float = 1e-4, betas=(0.9, 0.98), eps: float = 1e-8,
                 grad_norm: Optional[float] = None, sched: str = "warmup_cosine", warmup_steps: int = 0,
                 total_steps: int = 1, sched_gamma: float = 1.0, num_gpu: int = 1, group=None):
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
        # torch.optim.AdamW (the reference's optimizer) skips a parameter whose .grad is None: no decay, no moment update;
        # the flat kernel updates every element of the flat buffer.  So the flat buffer holds only parameters that DO get
        # gradients: ``model.unused_parameters()`` (the bypassed T5 encoder of the generation head) are left out up front,
        # and the first step probes the rest (as DDP's find_unused_parameters does, trainer/build.py:66-75): parameters
        # without a gradient are dropped from the layout before any update (forward_backward).  A parameter used only in
        # SOME steps keeps being decayed in the steps it is unused -- the remaining, documented divergence.
        unused = {id(p) for p in (model.unused_parameters() if hasattr(model, "unused_parameters") else ())}
        self._probed = False
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
    def forward_backward(self, data_dict: dict) -> torch.Tensor:
        self.model.zero_grad(set_to_none=True)
        enc = getattr(self.model, "unified_encoder", None)
        if enc is not None:
            enc.grad_arena_dirty = False      # this step owns the arena: one backward per step (fused.py raises otherwise)
        out = self.model(dict(data_dict))
        loss = self.loss_fn(out)
        loss.backward()
        if not self._probed and not torch.cuda.is_current_stream_capturing():
            self._probed = True
            missing = {id(p) for p in self.reducer.params if p.grad is None}
            if missing:
                # first step: parameters this configuration never reaches leave the flat buffer (own storage, never updated --
                # what torch.optim.AdamW does with grad-None parameters); nothing has been updated yet, so re-laying out and
                # re-running the step is exact
                for p in self.reducer.params:
                    if id(p) in missing:
                        p.data = p.data.clone()
                self._layout([dict(g, params=[p for p in g["params"] if id(p) not in missing]) for g in self.groups])
                return self.forward_backward(data_dict)
        self.reducer.pack()
        return loss.detach()

    def all_reduce(self) -> None:
        self.reducer.all_reduce()     # mean over ranks (DDP semantics); no-op at world size 1

    def optimizer_step(self) -> None:
        n, s = self.flat_g.numel(), L.stream()
        L.check(L.lib().pq3d_sumsq_partials(L.ptr(self.flat_g), n, L.ptr(self.partials), s), "pq3d_sumsq_partials")
        L.check(L.lib().pq3d_train_scalars(C.byref(self.hp), L.ptr(self.step_count), L.ptr(self.partials),
                                           L.ptr(self.scalars), s), "pq3d_train_scalars")
        L.check(L.lib().pq3d_adamw(L.ptr(self.flat_p), L.ptr(self.flat_g), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq),
                                   n, C.byref(self.hp), C.byref(self.segs), L.ptr(self.scalars), s), "pq3d_adamw")

    def step(self, data_dict: dict) -> torch.Tensor:
        """One full training step; returns the (detached) loss.  grad norm / lr of the step: ``self.scalars``."""
        loss = self.forward_backward(data_dict)
        self.all_reduce()
        self.optimizer_step()
        return loss

    # -- introspection ---------------------------------------------------------------------------------------------
    @property
    def last_grad_norm(self) -> torch.Tensor:
        return self.scalars[4]

    @property
    def last_lr(self) -> torch.Tensor:
        return self.scalars[0]

    def state_dict(self) -> dict:
        return {"step": self.step_count.clone(), "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone()}

    def load_state_dict(self, sd: dict) -> None:
        self.step_count.copy_(sd["step"]); self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])
