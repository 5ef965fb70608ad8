"""GPU: GraphedQuery3D (forward and backward as one HIP-graph replay each behind an ordinary autograd node) reproduces the
eager model: outputs bit-identical in eval-like arithmetic (same kernels, same order), gradients equal up to the atomics'
summation order, on fresh data with the same shapes -- the drop-in path for a trainer that cannot capture whole steps."""
import pytest
import torch

from pq3d_amd.graphed import GraphedQuery3D
from pq3d_amd.modules import set_compute, set_dropout
from tests import util

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("mode", ["direct", "autograd"])
@pytest.mark.parametrize("name", ["F4_c2_slice", "F4b_c4_slice", "F5_dimloc6"])   # dimloc6: a Linear used twice (tied slot use)
def test_graphed_model_matches_eager(name, mode):
    _z, args = util.load_fixture(name)
    _cfg, model, sd, dd = util.model_case(args)
    set_compute(model, "bf16")
    model.to(DEV).train()
    set_dropout(model, 0.0)
    ddv = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in dd.items()}
    gm = GraphedQuery3D(model, ddv, mode=mode)
    # fresh data of the same shapes
    _c2, _m2, _sd2, dd2 = util.model_case(dict(args, data_seed=args["data_seed"] + 5))
    dd2 = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in dd2.items()}
    res = []
    arena = (model.unified_encoder.grad_arena, model.unified_encoder.grad_arena_buffers) if mode == "direct" else None
    for runner in (lambda d: model(dict(d)), gm):
        model.zero_grad(set_to_none=True)
        if arena is not None:    # the eager reference run keeps its gradients out of the wrapper's flat buffers
            model.unified_encoder.grad_arena = None if runner is not gm else arena[0]
        out = runner(dd2)
        loss = util.synthetic_loss(out, args["heads"], out["query_embeds"])
        loss.backward()
        res.append((out["query_embeds"].detach().clone(), loss.detach().clone(),
                    {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
    (q1, l1, g1), (q2, l2, g2) = res
    assert torch.equal(q1, q2), "graph replay must reproduce the eager forward bit for bit"
    assert abs(l1.item() - l2.item()) <= 1e-6 * max(1.0, abs(l1.item()))
    assert sorted(g1) == sorted(g2)
    gmax = max(float(v.norm()) for v in g1.values())
    for n in g1:
        assert float((g1[n] - g2[n]).norm()) <= 2e-2 * max(float(g1[n].norm()), 1e-2 * gmax), n


def _case(name="F4_c2_slice", compute="fp32"):
    _z, args = util.load_fixture(name)
    _cfg, model, _sd, dd = util.model_case(args)
    set_compute(model, compute)
    model.to(DEV).train()
    set_dropout(model, 0.0)
    ddv = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in dd.items()}
    _c2, _m2, _sd2, dd2 = util.model_case(dict(args, data_seed=args["data_seed"] + 5))
    dd2 = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in dd2.items()}
    return args, model, ddv, dd2


@pytest.mark.parametrize("mode", ["direct", "autograd"])
def test_graphed_gradient_accumulation_equals_sum_of_micro_batches(mode):
    """Two backward passes without zero_grad ACCUMULATE (torch semantics; the reference trains under
    accelerator.accumulate, trainer/query3d_trainer.py:35): the accumulating variant of the captured backward adds into the
    flat buffers.  Equals the sum of the two micro-batches' separately computed gradients to fp32 rounding."""
    args, model, dda, ddb = _case()
    gm = GraphedQuery3D(model, dda, mode=mode, accumulation=True)
    loss_of = lambda out: util.synthetic_loss(out, args["heads"], out["query_embeds"])
    singles = []
    for d_ in (dda, ddb):
        model.zero_grad(set_to_none=True)
        loss_of(gm(d_)).backward()
        singles.append({n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    model.zero_grad(set_to_none=True)
    loss_of(gm(dda)).backward()
    loss_of(gm(ddb)).backward()        # no zero_grad in between -> accumulates
    gmax = max(float(v.norm()) for v in singles[0].values())
    for n, p in model.named_parameters():
        if n not in singles[0]:
            continue
        want = singles[0][n] + singles[1][n]
        assert float((p.grad - want).norm()) <= 2e-4 * max(float(want.norm()), 1e-2 * gmax), n
    # and a fresh step after zero_grad is again a single micro-batch
    model.zero_grad(set_to_none=True)
    loss_of(gm(dda)).backward()
    for n, p in model.named_parameters():
        if n in singles[0]:
            assert float((p.grad - singles[0][n]).norm()) <= 2e-4 * max(float(singles[0][n].norm()), 1e-2 * gmax), n


def test_graphed_autograd_mode_runs_parameter_hooks_and_direct_mode_guards_stale_forwards():
    args, model, dda, ddb = _case()
    gm = GraphedQuery3D(model, dda, mode="autograd")
    fired = []
    p0 = next(p for p in model.unified_encoder.parameters() if p.requires_grad)
    h = p0.register_post_accumulate_grad_hook(lambda p: fired.append(1))
    model.zero_grad(set_to_none=True)
    out = gm(ddb)
    util.synthetic_loss(out, args["heads"], out["query_embeds"]).backward()
    h.remove()
    assert fired, "mode='autograd' must deliver gradients through AccumulateGrad (DDP's hooks live there)"
    assert p0.grad is not None and p0.grad.data_ptr() == gm._grad_view(p0).data_ptr(), "the view must be adopted without a copy"
    # a backward whose forward was overwritten by a later forward of the same wrapper raises instead of using stale state
    out1 = gm(dda)
    _out2 = gm(ddb)
    with pytest.raises(RuntimeError, match="overwritten by a later forward"):
        util.synthetic_loss(out1, args["heads"], out1["query_embeds"]).backward()
    # captured constants must not change silently; shapes are checked
    bad = dict(ddb)
    const_keys = [k for k, v in bad.items() if not torch.is_tensor(v)]
    if const_keys:
        bad[const_keys[0]] = "something else"
        with pytest.raises(ValueError, match="differs from the captured constant"):
            gm(bad)
    bad = dict(ddb)
    k0 = next(k for k, v in bad.items() if torch.is_tensor(v) and v.ndim >= 2)
    bad[k0] = bad[k0][:, :-1]
    with pytest.raises(ValueError, match="was captured with"):
        gm(bad)


def test_autograd_mode_accumulation_fires_hooks_on_every_micro_batch():
    """under no_sync-style accumulation DDP reduces on the LAST micro-batch, from its AccumulateGrad hooks -- an
    accumulating replay that hands autograd nothing would never trigger them.  The 'autograd' mode forms every micro-batch's
    gradients in a second set of buffers and returns them: hooks run on each backward, .grad keeps aliasing the first set."""
    args, model, dda, ddb = _case()
    gm = GraphedQuery3D(model, dda, mode="autograd", accumulation=True)
    fired = []
    ps = [p for p in model.parameters() if p.requires_grad and id(p) not in gm._unused]
    hs = [p.register_post_accumulate_grad_hook(lambda p_: fired.append(id(p_))) for p in ps]
    loss_of = lambda out: util.synthetic_loss(out, args["heads"], out["query_embeds"])
    model.zero_grad(set_to_none=True)
    loss_of(gm(dda)).backward()
    n1 = len(fired)
    loss_of(gm(ddb)).backward()
    for h in hs:
        h.remove()
    assert n1 == len(ps) and len(fired) == 2 * len(ps), "every parameter's hook must run on both micro-batches"
    for p in ps:
        assert p.grad.data_ptr() == gm._grad_view(p).data_ptr()


def test_parameters_the_batch_never_reaches_keep_grad_none():
    """A parameter without a gradient in the captured backward gets None (not a zero-filled view): an eager
    torch.optim.AdamW then skips it, as it does after the reference's eager backward."""
    args, model, dda, _ddb = _case()
    extra = torch.nn.Parameter(torch.ones(4, device=DEV))      # registered, never used by forward
    model.register_parameter("never_used", extra)
    for mode in ("direct", "autograd"):
        gm = GraphedQuery3D(model, dda, mode=mode)
        model.zero_grad(set_to_none=True)
        out = gm(dda)
        util.synthetic_loss(out, args["heads"], out["query_embeds"]).backward()
        assert extra.grad is None, mode
        assert all(p.grad is not None for n, p in model.named_parameters() if n != "never_used" and id(p) not in gm._unused)
        model.unified_encoder.grad_arena = None


def test_arena_refuses_a_tied_weight_that_also_gets_a_gradient_outside_the_arena():
    """a fresh arena pass lets a parameter's second arena-aware use add into the slot in place; if a function that
    does not use the arena also produces a gradient for it, autograd sums out of place and the in-place part would be lost
    silently -- the pass must end with an error instead."""
    from pq3d_amd import ops
    from pq3d_amd._lib import F32
    from pq3d_amd.parallel import FlatGradAllReducer
    w = torch.nn.Parameter(torch.randn(16, 16, device=DEV) * 0.1)
    red = FlatGradAllReducer([w])
    x = torch.randn(8, 16, device=DEV)
    # tied use through arena-aware functions only: fine, gradient = both uses
    y = ops.linear(ops.linear(x, w, None, ct=F32), w, None, ct=F32).sum()
    with ops.grad_arena(red.slots(), red.flat):
        y.backward()
    g_tied = w.grad.detach().clone()
    w.grad = None
    y = ops.linear(ops.linear(x, w, None, ct=F32), w, None, ct=F32).sum()
    y.backward()
    torch.testing.assert_close(g_tied, w.grad, rtol=1e-5, atol=1e-6)
    w.grad = None
    z = ops.linear(ops.linear(x, w, None, ct=F32), w, None, ct=F32).sum() + (w * w).sum()
    with pytest.raises(RuntimeError, match="gradient arena"):
        with ops.grad_arena(red.slots(), red.flat):
            z.backward()


def test_fused_backward_accumulates_into_the_arena_until_gradients_are_reset():
    """Eager path with a shared gradient arena (TrainStep's / the DP reducer's flat buffer): a second backward before
    zero_grad adds in place; TrainStep.step([micro-batches]) equals the step on the summed, 1/k-scaled gradients."""
    from pq3d_amd.parallel import FlatGradAllReducer
    args, model, dda, ddb = _case()
    enc = model.unified_encoder
    params = [p for p in model.parameters() if p.requires_grad]
    dec = {id(p) for p in enc.parameters()} | ({id(p) for p in model.mask_head.parameters()} if hasattr(model, "mask_head") else set())
    red = FlatGradAllReducer(params, groups=[[p for p in params if id(p) in dec], [p for p in params if id(p) not in dec]])
    enc.grad_arena, enc.grad_arena_buffers = red.slots(), [red.flat[0]]
    loss_of = lambda out: util.synthetic_loss(out, args["heads"], out["query_embeds"])
    singles = []
    for d_ in (dda, ddb):
        model.zero_grad(set_to_none=True)
        loss_of(model(dict(d_))).backward()
        singles.append({n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    model.zero_grad(set_to_none=True)
    loss_of(model(dict(dda))).backward()
    loss_of(model(dict(ddb))).backward()
    gmax = max(float(v.norm()) for v in singles[0].values())
    for n, p in model.named_parameters():
        if n in singles[0]:
            want = singles[0][n] + singles[1][n]
            assert float((p.grad - want).norm()) <= 2e-4 * max(float(want.norm()), 1e-2 * gmax), n
    enc.grad_arena = None


def test_encoder_gradients_written_into_the_offered_arena_slots():
    """With every flat buffer in enc.grad_arena_buffers the decoder backward zeroes them all in its one launch and offers the
    input encoders' slots to their backward functions for that pass (ops.arena_offer): those gradients land in the flat
    buffer (no pack copy), equal the plain path's, accumulate over micro-batches, and the offer ends with the pass."""
    from pq3d_amd import ops
    from pq3d_amd.parallel import FlatGradAllReducer
    args, model, dda, ddb = _case()
    enc = model.unified_encoder
    params = [p for p in model.parameters() if p.requires_grad]
    dec = {id(p) for p in enc.parameters()} | ({id(p) for p in model.mask_head.parameters()} if hasattr(model, "mask_head") else set())
    loss_of = lambda out: util.synthetic_loss(out, args["heads"], out["query_embeds"])
    enc.grad_arena = None
    plain = []
    for d_ in (dda, ddb):
        model.zero_grad(set_to_none=True)
        loss_of(model(dict(d_))).backward()
        plain.append({n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    red = FlatGradAllReducer(params, groups=[[p for p in params if id(p) in dec], [p for p in params if id(p) not in dec]])
    slots = red.slots()
    enc.grad_arena, enc.grad_arena_buffers = slots, list(red.flat)
    model.zero_grad(set_to_none=True)
    loss_of(model(dict(dda))).backward()
    assert ops._Arena.mode is None   # the offer ended with the backward pass
    in_place = [n for n, p in model.named_parameters() if id(p) not in dec and p.grad is not None
                and p.grad.data_ptr() == slots[id(p)][0][slots[id(p)][1]:].data_ptr()]
    assert in_place, "no encoder gradient was written in place"
    gmax = max(float(v.norm()) for v in plain[0].values())
    for n, p in model.named_parameters():
        if n in plain[0]:
            assert float((p.grad - plain[0][n]).norm()) <= 2e-4 * max(float(plain[0][n].norm()), 1e-2 * gmax), n
    red.pack()   # the remaining (foreign) gradients are copied; in-place ones are skipped
    for n, p in model.named_parameters():
        if n in plain[0]:
            fl, o, k = slots[id(p)]
            assert float((fl[o:o + k].view_as(p) - plain[0][n]).norm()) <= 2e-4 * max(float(plain[0][n].norm()), 1e-2 * gmax), n
    loss_of(model(dict(ddb))).backward()   # second micro-batch: accumulate
    for n, p in model.named_parameters():
        if n in plain[0]:
            want = plain[0][n] + plain[1][n]
            assert float((p.grad - want).norm()) <= 2e-4 * max(float(want.norm()), 1e-2 * gmax), n
    # mixed state: an encoder gradient still aliasing its slot while the decoder's were reset
    for p in params:
        if id(p) in dec:
            p.grad = None
    with pytest.raises(RuntimeError, match="alias the shared gradient arena"):
        loss_of(model(dict(dda))).backward()
    model.zero_grad(set_to_none=True)
    enc.grad_arena = None


def test_per_layer_gradient_buckets_report_readiness_in_reverse_layer_order_and_change_no_gradient():
    """Data-parallel mode of the fused backward: with enc.grad_bucket_per_layer the weight-gradient products of
    a layer (incl. the K/V rows of its in_proj weights and its spatial-bias projection) are flushed when that layer's
    backward ends and enc.grads_ready(layer) fires -- last layer first -- then 'decoder'.  Same gradients as the single
    flush at the end (same products, another launch grouping)."""
    from pq3d_amd.parallel import FlatGradAllReducer
    args, model, dda, _ddb = _case("F4b_c4_slice")
    enc = model.unified_encoder
    params = [p for p in model.parameters() if p.requires_grad]
    dec = {id(p) for p in enc.parameters()} | {id(p) for p in model.mask_head.parameters()}
    red = FlatGradAllReducer(params, groups=[[p for p in params if id(p) in dec], [p for p in params if id(p) not in dec]])
    enc.grad_arena, enc.grad_arena_buffers = red.slots(), [red.flat[0]]
    loss_of = lambda out: util.synthetic_loss(out, args["heads"], out["query_embeds"])
    res = []
    for per_layer in (False, True):
        tags = []
        enc.grads_ready, enc.grad_bucket_per_layer = tags.append, per_layer
        model.zero_grad(set_to_none=True)
        loss_of(model(dict(dda))).backward()
        res.append(({n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}, tags))
    enc.grads_ready, enc.grad_bucket_per_layer, enc.grad_arena = None, False, None
    (g0, t0), (g1, t1) = res
    L_ = len(enc.unified_encoder)
    assert t0 == ["heads", "decoder"] and t1 == ["heads"] + list(range(L_ - 1, -1, -1)) + ["decoder"], (t0, t1)   # "heads": the decoder backward starts (output heads final)
    gmax = max(float(v.norm()) for v in g0.values())
    for n in g0:
        assert float((g0[n] - g1[n]).norm()) <= 1e-4 * max(float(g0[n].norm()), 1e-2 * gmax), n


def test_whole_pass_gradient_arena_with_modular_functions():
    """ops.grad_arena around backward(): linear (+ fused bias gradient), grouped linear, RMSNorm and embedding backward write
    their parameter gradients into the offered slots -- same values as without the arena, a tied weight used twice adds in
    place once, a second backward before the gradients are reset accumulates, a sliced stacked weight is left alone."""
    from pq3d_amd import ops
    from pq3d_amd._lib import BF16, F32
    from pq3d_amd.parallel import FlatGradAllReducer
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(11)
    P = lambda *s: torch.nn.Parameter((torch.randn(*s, generator=g) * 0.1).to(dev))
    emb, w1, b1, wq, wk, wn, stacked = P(50, 64), P(128, 64), P(128), P(64, 128), P(64, 128), P(128), P(192, 64)
    params = [emb, w1, b1, wq, wk, wn, stacked]
    ids = torch.randint(0, 50, (3, 17), generator=g).to(dev)

    def loss_of(scale=1.0):
        x = ops.embedding(emb, ids)                                   # [3, 17, 64]
        h = ops.linear(x, w1, b1, ct=F32, act="relu")                 # fused bias gradient
        h = ops.rmsnorm(h, wn)
        q, k = ops.linear_group([h, h], [wq, wk], ct=F32)
        s = ops.linear(q + k, stacked[:64], None, ct=F32)             # the first rows of a stacked weight: NOT that parameter
        logits = ops.linear(s, emb, None, ct=F32)                     # tied: the embedding table again
        return (logits.square().mean() + h.mean()) * scale

    def grads():
        return [p.grad.detach().clone() for p in params]

    for p in params:
        p.grad = None
    loss_of().backward()
    ref1 = grads()
    loss_of(0.5).backward()
    ref2 = grads()
    red = FlatGradAllReducer(params)
    slots = red.slots()
    for p in params:
        p.grad = None
    with ops.grad_arena(slots, red.flat):
        loss_of().backward()
    assert ops._Arena.mode is None and ops._Arena.pending is None
    in_place = [p.grad.data_ptr() == slots[id(p)][0].data_ptr() + 4 * slots[id(p)][1] for p in params]
    assert in_place[:6] == [True] * 6 and not in_place[6], in_place
    for a, b in zip(grads(), ref1):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6)
    red.pack()
    for p, b in zip(params, ref1):
        fl, o, n = slots[id(p)]
        torch.testing.assert_close(fl[o:o + n].view_as(p), b, rtol=1e-4, atol=1e-6)
    with ops.grad_arena(slots, red.flat):   # second micro-batch: accumulate (the stacked weight through autograd)
        loss_of(0.5).backward()
    for a, b in zip(grads(), ref2):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6)
    red.pack()
    for p, b in zip(params, ref2):
        fl, o, n = slots[id(p)]
        torch.testing.assert_close(fl[o:o + n].view_as(p), b, rtol=1e-4, atol=1e-6)
