"""CPU oracle of the training step around the decoder.  TEST INFRASTRUCTURE ONLY (see
pq3d_oracle.py's header for who may import it).

Restates, in plain tensor arithmetic, what the reference's trainer does after the loss is formed
(trainer/query3d_trainer.py:18-28): ``clip_grad_norm_`` (trainer/build.py:144-145), ``torch.optim.AdamW.step``
(third-party arithmetic: torch's documented algorithm, amsgrad=False) on the parameter groups of optim/utils.py:1-18,
then ``LambdaLR.step`` with the lambdas of optim/scheduler.py:5-17.  Pinned against the reference's own
optimizer/scheduler objects by the fixture tests/golden/F7_adamw_c1.npz (tests/golden/make_golden.py)."""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

Tensor = torch.Tensor
NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight")   # optim/utils.py:2 (substring match on the parameter name)


def weight_decay_of(name: str) -> float:
    """optim/utils.py:1-18.  NB: 'norm.weight' does not contain 'LayerNorm.weight' -> LayerNorm weights ARE decayed."""
    return 0.0 if any(nd in name for nd in NO_DECAY) else 0.01


def warmup_cosine(step: int, warmup_step: int, total_step: int) -> float:
    """optim/scheduler.py:5-8."""
    if step <= warmup_step and warmup_step > 0:
        return step / warmup_step
    return max(0.5 * (1 + math.cos((step - warmup_step) / (total_step - warmup_step) * math.pi)), 1e-5)


def warmup_exp(step: int, warmup_step: int, total_step: int, gamma: float) -> float:
    """optim/scheduler.py:11-14."""
    if step <= warmup_step and warmup_step > 0:
        return step / warmup_step
    return gamma ** (step * 1.0 / (total_step - warmup_step))


def lr_factor(sched: str, step: int, warmup: int, total: int, gamma: float = 1.0) -> float:
    if sched == "constant":
        return 1.0
    return warmup_cosine(step, warmup, total) if sched == "warmup_cosine" else warmup_exp(step, warmup, total, gamma)


def clip_coef(grads: List[Tensor], max_norm: Optional[float]):
    """torch.nn.utils.clip_grad_norm_: coef = clamp(max_norm / (total_norm + 1e-6), max=1)."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    if not max_norm:
        return 1.0, total
    return float(min(1.0, max_norm / (float(total) + 1e-6))), total


class AdamWState:
    def __init__(self):
        self.t = 0
        self.m: Dict[str, Tensor] = {}
        self.v: Dict[str, Tensor] = {}


def adamw_step(params: Dict[str, Tensor], grads: Dict[str, Tensor], st: AdamWState, *, lr: float, betas=(0.9, 0.98),
               eps: float = 1e-8, grad_norm: Optional[float] = None, sched: str = "warmup_cosine", warmup_steps: int = 0,
               total_steps: int = 1, gamma: float = 1.0, lr_of: Optional[Dict[str, float]] = None,
               num_gpu: int = 1):
    """One optimizer + scheduler step IN PLACE on ``params`` (fp32 tensors keyed by parameter name).
    Returns (lr used, gradient norm before clipping).  ``num_gpu`` > 1 reproduces the reference's multi-process
    schedule: get_scheduler scales the warm-up by num_gpu (optim/scheduler.py:20) and the accelerate-prepared
    scheduler (trainer/build.py:123) steps the LambdaLR num_gpu times per optimizer step, so the factor of optimizer
    step k is lambda(k * num_gpu, warmup_steps * num_gpu, total_steps)."""
    names = [n for n in params if n in grads]
    coef, total = clip_coef([grads[n] for n in names], grad_norm)
    # LambdaLR: lambda(number of scheduler steps taken so far)
    fac = lr_factor(sched, st.t * num_gpu, warmup_steps * num_gpu, total_steps, gamma)
    st.t += 1
    b1, b2 = betas
    bc1, bc2 = 1.0 - b1 ** st.t, 1.0 - b2 ** st.t
    for n in names:
        g = grads[n] * coef
        lr_n = (lr_of[n] if lr_of else lr) * fac
        p = params[n]
        p.mul_(1.0 - lr_n * weight_decay_of(n))
        m = st.m.setdefault(n, torch.zeros_like(p))
        v = st.v.setdefault(n, torch.zeros_like(p))
        m.lerp_(g, 1.0 - b1)
        v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-(lr_n / bc1))
    return lr * fac, total
