"""CPU: the train-step oracle (oracle/train_oracle.py: clip_grad_norm_ + AdamW + LambdaLR restated) against the F7
fixtures produced with the reference's own optimizer / scheduler objects (tests/golden/make_golden.py:run_train_case)."""
import numpy as np
import pytest
import torch

from oracle import pq3d_oracle as O
from oracle import train_oracle as T
from tests import util


def oracle_train(args, steps):
    _cfg, _model, sd, dd = util.model_case(args)
    params = {k: v.clone() for k, v in sd.items() if v.dtype.is_floating_point and not k.endswith("gauss_B")}
    frozen = {k: v for k, v in sd.items() if k not in params}
    lr_of = {n: (args["head_lr"] if (args.get("head_lr") and n.startswith("ground_head.")) else args["lr"])
             for n in params}
    st = T.AdamWState()
    init = {k: v.clone() for k, v in params.items()}
    hist = []
    for _ in range(steps):
        leaf = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        collect = []
        out = O.query3d_unified_forward({**leaf, **frozen}, util.oracle_cfg(args),
                                        {k: (v.clone() if torch.is_tensor(v) else v) for k, v in dd.items()},
                                        collect=collect)
        loss = util.synthetic_loss(out, args["heads"], collect[-1])
        loss.backward()
        grads = {k: v.grad for k, v in leaf.items() if v.grad is not None}
        lr, norm = T.adamw_step(params, grads, st, lr=args["lr"], grad_norm=args["grad_norm"],
                                warmup_steps=args["warmup_steps"], total_steps=args["total_steps"], lr_of=lr_of)
        hist.append((loss.item(), float(norm), lr, {k: params[k] - init[k] for k in params}))
    return hist


@pytest.mark.parametrize("name", util.fixtures("F7_"))
def test_train_oracle_matches_reference_optimizer(name):
    z, args = util.load_fixture(name)
    hist = oracle_train(args, args["steps"])
    for s, (loss, norm, lr, delta) in enumerate(hist):
        assert abs(loss - z["loss"][s]) <= 2e-5 * max(1.0, abs(z["loss"][s])), (s, loss, z["loss"][s])
        assert abs(norm - z["grad_norm"][s]) <= 1e-4 * z["grad_norm"][s]
        assert abs(lr - z["lr"][s]) <= 1e-9
        for n, dlt in delta.items():
            # updates are O(lr); Adam's m/sqrt(v) amplifies gradient rounding where |g| ~ eps -> absolute tolerance
            # 2e-3 of the step size, relative 1e-3
            util.check_against(z, f"delta/{s}/{n}", dlt, atol=2e-3 * max(args["lr"], 1e-12) * (1 if lr > 0 else 0) + 1e-9,
                               rtol=2e-3, cap=util.MAX_TRAIN, what=f"step {s} ")


def test_weight_decay_groups_follow_reference_quirk():
    assert T.weight_decay_of("unified_encoder.unified_encoder.0.ffn.norm.weight") == 0.01   # NOT exempt in the reference
    assert T.weight_decay_of("unified_encoder.unified_encoder.0.ffn.norm.bias") == 0.0
    assert T.weight_decay_of("unified_encoder.unified_encoder.0.ffn.linear1.weight") == 0.01
    assert np.isclose(T.warmup_cosine(0, 2, 10), 0.0) and np.isclose(T.warmup_cosine(2, 2, 10), 1.0)


@pytest.mark.parametrize("num_gpu,sched,gamma", [(2, "warmup_cosine", 1.0), (8, "warmup_cosine", 1.0), (4, "warmup_exp", 0.1)])
def test_multi_process_schedule_matches_accelerate_prepared_lambdalr(num_gpu, sched, gamma):
    """The reference's scheduler under num_gpu processes: get_scheduler multiplies the warm-up by num_gpu
    (optim/scheduler.py:20) and trainer/build.py:123 hands the LambdaLR to accelerate, whose AcceleratedScheduler steps
    it num_processes times per optimizer step.  Pinned two ways: (1) the installed accelerate really does that (source
    check), (2) a torch LambdaLR driven that way gives exactly train_oracle.adamw_step's learning rates."""
    import inspect
    from accelerate.scheduler import AcceleratedScheduler
    src = inspect.getsource(AcceleratedScheduler.step)
    assert "num_processes" in src and "for _ in range(num_processes)" in src
    warm, total, lr0, steps = 3, 40, 1e-3, 12
    p = torch.nn.Parameter(torch.zeros(4))
    opt = torch.optim.AdamW([p], lr=lr0)
    fn = {"warmup_cosine": lambda s: T.warmup_cosine(s, warm * num_gpu, total),
          "warmup_exp": lambda s: T.warmup_exp(s, warm * num_gpu, total, gamma)}[sched]
    lam = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=fn)
    st = T.AdamWState()
    params = {"w": torch.zeros(4)}
    for _ in range(steps):
        want = opt.param_groups[0]["lr"]          # lr the optimizer step of this iteration uses
        opt.step()
        for _ in range(num_gpu):                  # AcceleratedScheduler.step (split_batches=False)
            lam.step()
        got, _ = T.adamw_step(params, {"w": torch.ones(4)}, st, lr=lr0, sched=sched, warmup_steps=warm, total_steps=total,
                              gamma=gamma, num_gpu=num_gpu)
        assert abs(got - want) <= 1e-12, (got, want)
