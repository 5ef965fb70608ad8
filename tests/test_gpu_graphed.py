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
@pytest.mark.parametrize("name", ["F4_c2_slice", "F4b_c4_slice"])
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
