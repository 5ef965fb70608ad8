"""GPU: the HIP training step (pq3d_amd/trainer.py: forward, backward, clip, flat AdamW, LR schedule -- three optimizer
kernels on one flat buffer) against the F7 fixtures made with the reference's own optimizer and scheduler objects."""
import pytest
import torch

from pq3d_amd.trainer import TrainStep
from tests import util

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build(args, compute="fp32"):
    cfg, model, sd, dd = util.model_case(args)
    cfg.solver["lr"] = args["lr"]
    if args.get("head_lr"):
        cfg.model.ground_head["lr"] = args["head_lr"]
    from pq3d_amd.modules import set_compute
    set_compute(model, compute)
    model.to(DEV).eval()     # fixtures: eval-mode forward (dropout off), full optimizer path
    ddv = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in dd.items()}
    wcache = {}

    def w(name, shape):      # loss weights resident on the device (no H2D copy inside a captured step)
        if name not in wcache:
            wcache[name] = util.loss_weight(name, shape).to(DEV)
        return wcache[name]

    def loss_fn(out):        # util.synthetic_loss with cached weights
        loss = 0.0
        if "ground" in args["heads"]:
            gl = out["ground_logits"]
            loss = loss + (torch.where(torch.isfinite(gl), gl, torch.zeros_like(gl)) * w("ground", gl.shape)).mean()
        if "mask" in args["heads"]:
            for i, (c, m) in enumerate(zip(out["predictions_class"], out["predictions_mask"])):
                cf = torch.where(torch.isfinite(c), c, torch.zeros_like(c))
                loss = loss + (cf * w(f"cls{i}", c.shape)).mean() + (m.clamp(min=-50.0) * w(f"mask{i}", m.shape)).mean()
        q = out["query_embeds"]
        return loss + (q * w("query", q.shape)).mean()

    ts = TrainStep(model, loss_fn, lr=args["lr"], grad_norm=args["grad_norm"], sched="warmup_cosine",
                   warmup_steps=args["warmup_steps"], total_steps=args["total_steps"])
    return model, ts, ddv


@pytest.mark.parametrize("name", util.fixtures("F7_"))
def test_train_step_matches_reference_optimizer(name):
    z, args = util.load_fixture(name)
    model, ts, ddv = build(args)
    init = {n: p.detach().clone() for n, p in model.named_parameters()}
    for s in range(args["steps"]):
        loss = ts.step(ddv)
        assert abs(loss.item() - z["loss"][s]) <= 3e-5 * max(1.0, abs(z["loss"][s])), (s, loss.item(), z["loss"][s])
        assert abs(ts.last_grad_norm.item() - z["grad_norm"][s]) <= 2e-4 * z["grad_norm"][s]
        assert abs(ts.last_lr.item() - z["lr"][s]) <= 1e-7
        lr = float(z["lr"][s])
        for n, p in model.named_parameters():
            rt = 1e-2 if "pairwise_loc_fc" in n else 3e-3    # ill-conditioned gradient, see test_gpu_model.py
            util.check_against(z, f"delta/{s}/{n}", p.detach() - init[n],
                               atol=3e-3 * args["lr"] * (1 if lr > 0 else 0) + 1e-9, rtol=rt, cap=util.MAX_TRAIN,
                               what=f"step {s} ")
    assert int(ts.step_count.item()) == args["steps"]


def test_parameters_stay_checkpoint_compatible_and_state_roundtrips():
    _z, args = util.load_fixture("F7_adamw_c1")
    model, ts, ddv = build(args)
    keys = set(model.state_dict().keys())
    ts.step(ddv); ts.step(ddv)
    assert set(model.state_dict().keys()) == keys
    for p in model.parameters():     # parameters are views of the flat buffer, still ordinary leaf Parameters
        assert p.is_leaf and p.data.untyped_storage().data_ptr() == ts.flat_p.untyped_storage().data_ptr()
    sd_model = {k: v.clone() for k, v in model.state_dict().items()}
    sd_opt = ts.state_dict()
    a = ts.step(ddv).item()
    model.load_state_dict(sd_model)
    ts.load_state_dict(sd_opt)
    b = ts.step(ddv).item()
    assert a == b


def test_whole_train_step_in_one_hip_graph_bf16():
    """forward + backward + clip + AdamW + schedule captured once and replayed: no host sync anywhere in the step."""
    _z, args = util.load_fixture("F7_adamw_mask")
    model, ts, ddv = build(args, "bf16")
    ref_model, ref_ts, _ = build(args, "bf16")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ts.forward_backward(ddv)          # warm-up of allocator / autograd; no optimizer step yet
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss = ts.step(ddv)
    # capture does not execute: the step counter is still 0 and the weights untouched
    assert int(ts.step_count.item()) == 0
    losses, ref_losses = [], []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        losses.append(loss.item())
        ref_losses.append(ref_ts.step(ddv).item())
    assert int(ts.step_count.item()) == 3
    assert losses[0] != losses[2]
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) <= 2e-3 * max(1.0, abs(b)), (losses, ref_losses)


def test_graph_replayed_gradients_equal_eager_gradients():
    """Every replay of a captured forward+backward must reproduce the eager gradients (regression: outputs the library
    used to zero with hipMemsetAsync were only correct on the FIRST replay of a HIP graph -- memset nodes were not
    re-executed -- so split-K / atomics results of later replays accumulated onto stale memory)."""
    _z, args = util.load_fixture("F4b_c4_slice")
    model, ts, ddv = build(dict(args, lr=1e-4, grad_norm=None, warmup_steps=0, total_steps=10), "bf16")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):          # eager reference on the side stream the capture will also use
        ts.forward_backward(ddv)
        ts.forward_backward(ddv)
        ref = ts.flat_g.clone()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ts.forward_backward(ddv)
    scale = float(ref.abs().max())
    for i in range(4):
        g.replay()
        torch.cuda.synchronize()
        assert torch.isfinite(ts.flat_g).all(), f"replay {i}"
        # same kernels, same inputs: only the order of split-K atomics may differ
        assert float((ts.flat_g - ref).abs().max()) <= 2e-3 * scale, f"replay {i}"


@pytest.mark.parametrize("num_gpu", [2, 8])
def test_multi_process_lr_schedule_matches_oracle(num_gpu):
    """num_gpu > 1: warm-up scaled by num_gpu AND num_gpu scheduler steps per optimizer step (accelerate-prepared
    LambdaLR, trainer/build.py:123): the device-side schedule must give train_oracle's learning rates (which
    tests/test_train_oracle.py pins to torch's LambdaLR driven the way accelerate drives it)."""
    from oracle import train_oracle as T
    _z, args = util.load_fixture("F7_adamw_c1")
    cfg, model, sd, dd = util.model_case(args)
    model.to(DEV).eval()
    ddv = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in dd.items()}
    ts = TrainStep(model, lambda out: out["query_embeds"].mean(), lr=1e-3, grad_norm=None, sched="warmup_cosine",
                   warmup_steps=2, total_steps=40, num_gpu=num_gpu)
    for s in range(9):
        ts.step(ddv)
        # == adamw_step(..., num_gpu=num_gpu)'s learning rate for optimizer step s (oracle/train_oracle.py)
        want = 1e-3 * T.lr_factor("warmup_cosine", s * num_gpu, 2 * num_gpu, 40)
        assert abs(float(ts.last_lr) - want) <= 1e-9 + 1e-6 * want, (s, float(ts.last_lr), want)


def test_bypassed_t5_encoder_is_not_touched_by_the_flat_optimizer():
    """torch.optim.AdamW skips grad-None parameters (no weight decay, no moments).  The generation head's T5 encoder is
    bypassed (encoder_outputs) and never gets a gradient: TrainStep keeps it out of the flat buffer, so three steps leave
    it bit-identical while the decoder side moves."""
    from pq3d_amd import synth
    from pq3d_amd.model import Query3DUnified, make_cfg
    _z, a = util.load_fixture("F8_t5_head")
    model = Query3DUnified(make_cfg(d=64, H=4, L=2, memories=["voxel", "mv"], heads=["generation"], t5=a["hf_config"]),
                           compute="fp32")
    synth.fill_module(model, 0)
    model.to(DEV).train()       # eval mode generates tokens instead of logits
    unused = model.unused_parameters()
    assert len(unused) > 0
    before = [p.detach().clone() for p in unused]
    proj_before = model.generation_head.input_proj[0].weight.detach().clone()
    dd = synth.synth_data_dict(2, 64, 10, {"voxel": 64, "mv": 64}, seed=3, memories=["voxel", "mv"])
    dd["response"] = torch.randint(2, 128, (2, 6))
    dd = {k: v.to(DEV) for k, v in dd.items()}

    def loss_fn(out):
        return torch.nn.functional.cross_entropy(out["generation_logits"].flatten(0, 1), out["generation_label"].flatten())

    ts = TrainStep(model, loss_fn, lr=1e-3, grad_norm=1.0, warmup_steps=0, total_steps=10)
    flat_ids = {id(p) for p in ts.reducer.params}
    assert not any(id(p) in flat_ids for p in unused)
    for _ in range(3):
        ts.step(dd)
    assert all(p.grad is not None for p in ts.reducer.params)      # the flat buffer holds exactly the parameters in use
    assert all(torch.equal(p, b) for p, b in zip(unused, before))
    assert not torch.equal(model.generation_head.input_proj[0].weight, proj_before)


def test_parameters_without_gradient_stay_in_the_layout_and_are_skipped_per_step():
    """A head the loss never reaches gets no gradient; torch.optim.AdamW neither decays nor moves it.  The flat layout is
    deterministic (no first-step probe: ranks / checkpoints / captures agree on it); the AdamW launch marks such parameters'
    segments 'skip' for the step, so they and their moments stay bit-identical while everything else moves."""
    from pq3d_amd import synth
    from pq3d_amd.model import Query3DUnified, make_cfg
    # a grounding head next to the mask head, and a loss that only reads the mask head's outputs: the grounding head's MLP is
    # in the layout (get_opt_params lists it, query3d_unified.py:224-238) but its parameters never see a gradient
    model = Query3DUnified(make_cfg(d=64, H=4, L=2, memories=["voxel", "mv"], heads=["mask", "ground"], use_self_mask=True,
                                    C=21, foc=(0, 2)), compute="fp32")
    synth.fill_module(model, 0)
    model.to(DEV).eval()
    dd = synth.synth_data_dict(2, 48, 10, {"voxel": 64, "mv": 64}, seed=3, memories=["voxel", "mv"])
    dd["tgt_object_id"] = torch.zeros(2, dtype=torch.long)
    dd = {k: v.to(DEV) for k, v in dd.items()}

    def loss_fn(out):
        return out["query_embeds"].square().mean() + sum(m.clamp(min=-50.0).mean() for m in out["predictions_mask"])

    ts = TrainStep(model, loss_fn, lr=1e-3, grad_norm=5.0, warmup_steps=0, total_steps=10)
    n0, lay0 = ts.flat_p.numel(), ts.layout()
    ts.forward_backward(dd)
    missing = [p for p in ts.reducer.params if p.grad is None]
    ground = {id(p) for p in model.ground_head.parameters()}
    assert missing and all(id(p) in ground for p in missing) and len(missing) == len(ground)
    before = [p.detach().clone() for p in missing]
    moved0 = [p.detach().clone() for p in ts.reducer.params if p.grad is not None][:4]
    for _ in range(3):
        ts.step(dd)
    assert ts.flat_p.numel() == n0 and ts.layout() == lay0          # layout never changes
    assert all(torch.equal(p, b) for p, b in zip(missing, before))
    off = 0
    for g in ts.groups:
        for p in g["params"]:
            k = p.numel()
            if any(p is q for q in missing):
                assert float(ts.exp_avg[off:off + k].abs().max()) == 0.0 and float(ts.exp_avg_sq[off:off + k].abs().max()) == 0.0
            off += k
    present = [p for p in ts.reducer.params if p.grad is not None][:4]
    assert any(not torch.equal(p, b) for p, b in zip(present, moved0))
    # checkpoint round trip validates the layout
    sd = ts.state_dict()
    ts.load_state_dict(sd)
    bad = dict(sd, layout=sd["layout"][1:])
    with pytest.raises(ValueError):
        ts.load_state_dict(bad)
