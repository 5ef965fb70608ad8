"""GPU: the background lane (pq3d_amd/overlap.py) -- the step as two captured graphs on two streams ordered by device-side
flags -- reproduces the single-stream step: forward bit-identical, gradients equal up to the atomics' summation order, over
many replays on fresh data; and a graph replayed WITHOUT its partner is reported (timed-out waits) instead of hanging."""
import pytest
import torch

from pq3d_amd import overlap
from pq3d_amd.modules import set_compute, set_dropout
from tests import util

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _case(name, compute):
    _z, args = util.load_fixture(name)
    _cfg, model, _sd, dd = util.model_case(args)
    set_compute(model, compute)
    model.to(DEV).train()
    set_dropout(model, 0.0)
    return args, model, {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in dd.items()}


def _loss(out, heads):
    """A capturable stand-in for util.synthetic_loss (which builds its weights on the host at every call)."""
    loss = out["query_embeds"].float().square().mean()
    if "ground" in heads:
        gl = out["ground_logits"]
        loss = loss + torch.where(torch.isfinite(gl), gl, torch.zeros_like(gl)).mean()
    if "mask" in heads:
        for m_, c_ in zip(out["predictions_mask"], out["predictions_class"]):
            loss = loss + m_.clamp(min=-50.0).mean() + torch.where(torch.isfinite(c_), c_, torch.zeros_like(c_)).mean()
    return loss


def _grads(model):
    return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("name,compute", [("F4_c2_slice", "fp32"), ("F4_c2_slice", "bf16"), ("F4b_c4_slice", "bf16"),
                                          ("F17_mixed_prompt", "fp32")])
def test_laned_step_matches_single_stream_step(name, compute):
    if name not in util.fixtures():
        pytest.skip(f"fixture {name} absent")
    args, model, dd = _case(name, compute)
    enc = model.unified_encoder
    static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in dd.items()}
    outs = {}

    def step():
        model.zero_grad(set_to_none=True)
        out = model(dict(static))
        loss = _loss(out, args["heads"])
        loss.backward()
        lane = overlap.current(enc)
        if lane is not None:
            lane.join()
        outs["q"], outs["loss"] = out["query_embeds"], loss

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
        enc.bg_lane = overlap.InlineLane()
        step()                      # the per-layer hand-overs, executed inline
        enc.bg_lane = None
        q_inline, g_inline = outs["q"].detach().clone(), _grads(model)
        step()
        q_ref, l_ref, g_ref = outs["q"].detach().clone(), outs["loss"].detach().clone(), _grads(model)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gmax = max(float(v.norm()) for v in g_ref.values())
    tol = 1e-5 if compute == "fp32" else 2e-2     # bf16 mode: split-K partial sums are rounded per launch grouping
    assert torch.equal(q_inline, q_ref)
    for n in g_ref:
        assert float((g_inline[n] - g_ref[n]).norm()) <= tol * max(float(g_ref[n].norm()), 1e-2 * gmax), ("inline", n)

    lane = overlap.BackgroundLane(DEV, cus=64)
    lg = overlap.LanedGraph(step, lane, owners=[enc])
    assert lg.n_background >= len(list(enc.unified_encoder)), "the fused backward handed nothing to the lane"
    for it in range(6):
        # gradients are views of buffers the captured graphs own: poison them so that a replay that skipped work shows
        for p in model.parameters():
            if p.grad is not None:
                p.grad.fill_(float("nan"))
        lg.replay()
        torch.cuda.synchronize()
        assert torch.equal(outs["q"], q_ref), f"replay {it}: forward differs"
        assert abs(float(outs["loss"]) - float(l_ref)) <= 1e-6 * max(1.0, abs(float(l_ref)))
        g = _grads(model)
        assert sorted(g) == sorted(g_ref)
        for n in g_ref:
            assert torch.isfinite(g[n]).all(), (it, n)
            assert float((g[n] - g_ref[n]).norm()) <= tol * max(float(g_ref[n].norm()), 1e-2 * gmax), (it, n)
    assert lane.errors() == 0
    # fresh data of the same shapes through the captured graphs
    _c2, _m2, _sd2, dd2 = util.model_case(dict(args, data_seed=args["data_seed"] + 5))
    for k, v in dd2.items():
        if torch.is_tensor(v):
            static[k].copy_(v.to(DEV))
    lg.replay()
    torch.cuda.synchronize()
    q_l, g_l = outs["q"].detach().clone(), _grads(model)
    with torch.cuda.stream(side):
        step()
    torch.cuda.synchronize()
    assert torch.equal(q_l, outs["q"])
    g_e = _grads(model)
    for n in g_e:
        assert float((g_l[n] - g_e[n]).norm()) <= tol * max(float(g_e[n].norm()), 1e-2 * gmax), ("fresh", n)
    lane.check()


def test_a_graph_replayed_without_its_partner_is_reported_not_hung():
    args, model, dd = _case("F4_c2_slice", "fp32")
    enc = model.unified_encoder

    def step():
        model.zero_grad(set_to_none=True)
        out = model(dict(dd))
        _loss(out, args["heads"]).backward()

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
        enc.bg_lane = overlap.InlineLane()
        step()
        enc.bg_lane = None
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    lane = overlap.BackgroundLane(DEV, cus=64, timeout_us=2000)
    lg = overlap.LanedGraph(step, lane, owners=[enc])
    lg.replay()
    torch.cuda.synchronize()
    assert lane.errors() == 0
    with torch.cuda.stream(lg.main):
        lg.gA.replay()              # the main graph alone: its final join can never be satisfied
    torch.cuda.synchronize()
    assert lane.errors() >= 1
    with pytest.raises(RuntimeError, match="timed out"):
        lane.check()
