"""GPU: the generation head (modules/heads/generation_head.py): input_proj on the HIP kernels + the HF T5 decoder body on
stock PyTorch-ROCm, against fixture F8 produced by the reference's own head class (tiny random-init architecture)."""
import numpy as np
import pytest
import torch

from pq3d_amd import synth
from pq3d_amd.modules import T5
from tests import util

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("body", ["hip", "hf"])
def test_t5_head_matches_reference_head(body):
    """body='hip': input_proj AND the teacher-forced T5 decoder on the HIP kernels; 'hf': stock HF body."""
    z, a = util.load_fixture("F8_t5_head")
    head = T5(None, variant="tiny", input_size=a["d"], use_projection=True, hf_config=a["hf_config"], body=body)
    head.compute = "fp32"
    sd = synth.fill_module(head, a["seed"])
    assert abs(synth.state_checksum(sd) - float(z["meta/weights_checksum"])) < 1e-6 * float(z["meta/weights_checksum"])
    head.to(DEV).eval()
    q = torch.from_numpy(z["q"]).to(DEV).requires_grad_(True)
    mask, labels = torch.from_numpy(z["mask"]).to(DEV), torch.from_numpy(z["labels"]).to(DEV)
    logits = head(q, mask, labels)
    util.check_against(z, "logits", logits, atol=2e-5, rtol=2e-5)
    loss = (logits * util.loss_weight("t5logits", logits.shape).to(DEV)).mean()
    loss.backward()
    assert abs(loss.item() - float(z["loss"])) < 1e-6
    util.check_against(z, "grad/q", q.grad, atol=1e-7, rtol=1e-3, cap=util.MAX_GRAD)
    for n, p in head.input_proj.named_parameters():
        util.check_against(z, "grad/input_proj." + n, p.grad, atol=1e-7, rtol=1e-3, cap=util.MAX_GRAD)
    with torch.no_grad():
        gen = head(q.detach(), mask, None)
    assert np.array_equal(gen.cpu().numpy(), z["generated"])


def test_model_with_generation_head_trains():
    from pq3d_amd.model import Query3DUnified, make_cfg
    _z, a = util.load_fixture("F8_t5_head")
    model = Query3DUnified(make_cfg(d=64, H=4, L=2, memories=["voxel", "mv"], heads=["generation"], t5=a["hf_config"]),
                           compute="bf16")
    synth.fill_module(model, 0)
    model.to(DEV).train()
    dd = synth.synth_data_dict(2, 64, 10, {"voxel": 64, "mv": 64}, seed=3, memories=["voxel", "mv"])
    dd["response"] = torch.randint(2, 128, (2, 6))
    out = model({k: v.to(DEV) for k, v in dd.items()})
    assert out["generation_logits"].shape == (2, 6, 128)
    torch.nn.functional.cross_entropy(out["generation_logits"].flatten(0, 1), out["generation_label"].flatten()).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.generation_head.input_proj.parameters())
    assert model.unified_encoder.unified_encoder[0].ffn.linear1.weight.grad is not None
    model.eval()
    with torch.no_grad():
        toks = model({k: v.to(DEV) for k, v in dd.items()})["generation_logits"]
    assert toks.dtype == torch.long and toks.shape[0] == 2


def test_t5_hip_body_train_mode_and_bf16():
    """dropout active (train mode) and bf16 operands: finite, stochastic, and close to the fp32 eval logits in bf16."""
    z, a = util.load_fixture("F8_t5_head")
    head = T5(None, variant="tiny", input_size=a["d"], use_projection=True, hf_config=a["hf_config"], body="hip")
    synth.fill_module(head, a["seed"])
    head.to(DEV)
    q = torch.from_numpy(z["q"]).to(DEV)
    mask, labels = torch.from_numpy(z["mask"]).to(DEV), torch.from_numpy(z["labels"]).to(DEV)
    head.compute = "bf16"
    head.eval()
    ref = head(q, mask, labels)
    g = torch.from_numpy(z["logits/sample"]).float()
    assert float((ref.detach().float().cpu().flatten()[:: max(1, ref.numel() // g.numel() + (ref.numel() % g.numel() > 0))][:g.numel()]
                  - g).abs().max()) < 5e-2 * float(g.abs().max())
    head.train()
    a1, a2 = head(q, mask, labels), head(q, mask, labels)
    assert torch.isfinite(a1).all() and not torch.equal(a1, a2)
    qg = q.clone().requires_grad_(True)
    head(qg, mask, labels).float().square().mean().backward()
    assert torch.isfinite(qg.grad).all() and all(p.grad is None or torch.isfinite(p.grad).all() for p in head.parameters())


def _hf_generate(head, q, mask, n_new):
    from transformers.modeling_outputs import BaseModelOutput
    enc = torch.nn.Sequential(*head.input_proj)(q) if head.use_projection else q
    out = head.model.generate(encoder_outputs=BaseModelOutput(last_hidden_state=enc), attention_mask=mask, do_sample=False,
                              max_new_tokens=n_new)
    return out[:, 1:]


@pytest.mark.parametrize("use_graph", [True, False])
def test_greedy_decoder_matches_hf_generate(use_graph):
    """KV-cache greedy decoding on the HIP kernels == HF generate (stock PyTorch-ROCm, fp32) token for token: several
    calls through the same captured graph (state reset), ragged encoder masks, and EOS / pad bookkeeping with an EOS id
    that the random-init model actually emits."""
    from pq3d_amd import t5
    _z, a = util.load_fixture("F8_t5_head")
    head = T5(None, variant="tiny", input_size=a["d"], use_projection=True, hf_config=a["hf_config"], body="hip")
    head.compute = "fp32"
    synth.fill_module(head, 5)
    head.to(DEV).eval()
    B, N, n_new = 5, 13, 23
    g = torch.Generator().manual_seed(1)
    dec = None
    for trial in range(3):
        q = torch.randn(B, N, a["d"], generator=g).to(DEV)
        mask = (torch.arange(N)[None] < torch.tensor([N, 3, 7, N, 1])[:, None]).to(DEV)
        with torch.no_grad():
            ref = _hf_generate(head, q, mask, n_new)
            if trial == 1:      # make EOS reachable: the token sequence 0 emits at its 3rd step becomes the EOS id
                head.model.generation_config.eos_token_id = int(ref[0, 2])
                ref = _hf_generate(head, q, mask, n_new)
                dec = None
            if dec is None:
                dec = t5.GreedyDecoder(head.model, B, N, head.ct, n_new, DEV, use_graph=use_graph)
            enc = torch.nn.Sequential(*head.input_proj)(q)
            got = dec(enc, mask)
        assert got.shape == ref.shape, (trial, got.shape, ref.shape)
        assert torch.equal(got, ref), (trial, got, ref)
    assert (ref == head.model.generation_config.eos_token_id).any() and ref.shape[1] <= n_new


def test_greedy_generation_through_the_head_and_budget():
    """T5.forward(labels=None) routes to the HIP decoder; the reference's ``max_new_tokens`` kwarg (config.update) sets
    the budget; tokens are self-consistent with the teacher-forced logits (argmax of position t given the prefix)."""
    from pq3d_amd import t5
    _z, a = util.load_fixture("F8_t5_head")
    head = T5(None, variant="tiny", input_size=a["d"], use_projection=True, hf_config=a["hf_config"], body="hip", max_new_tokens=9)
    head.compute = "fp32"
    synth.fill_module(head, 2)
    head.to(DEV).eval()
    assert t5.new_token_budget(head.model) == 9
    q = torch.randn(3, 6, a["d"], generator=torch.Generator().manual_seed(0)).to(DEV)
    mask = torch.ones(3, 6, dtype=torch.bool, device=DEV)
    with torch.no_grad():
        toks = head(q, mask, None)
        assert toks.shape == (3, 9) and toks.dtype == torch.long
        logits = head(q, mask, toks)
    assert torch.equal(logits.argmax(-1), toks)
    with torch.no_grad():
        assert torch.equal(head(q, mask, None), toks)     # cached decoder, replayed


@pytest.mark.parametrize("second_consumer", [False, True])
def test_dropout_gradient_handover_equals_the_fallback_bit_for_bit(second_consumer):
    """`ops.linear(..., drop=site, masked_grad=slot)` lets the NEXT rmsnorm's backward kernel write the dropout-masked
    gradient of the projection (pq3d_amd/t5.py proj_residual / norm_res) instead of a launch of its own.  The hand-over is matched
    by address: it must equal the `_dropout_apply` fallback bit for bit, and with a SECOND consumer of the projection's output
    autograd hands the projection an accumulated gradient (another address) -> the fallback runs, still correct.
    (A tensor hook that rewrites the norm's input gradient IN PLACE would not reach the pre-masked copy: unsupported, documented
    in ops.linear.)"""
    from pq3d_amd import _lib as L, ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(5)
    R, d = 96, 512
    o = torch.randn(R, d, generator=g).to(dev)
    x0 = torch.randn(R, d, generator=g).to(dev)
    w = (torch.randn(d, d, generator=g) * 0.05).to(dev)
    nw = (1 + 0.1 * torch.randn(d, generator=g)).to(dev)
    gy = torch.randn(R, d, generator=g).to(dev)
    drop = ops.make_drop(0.1, ops.drop_site(7 << 20, 0, 0), dev)

    def run(handover):
        oo, xx, ww, nn_ = (t.clone().requires_grad_(True) for t in (o, x0, w, nw))
        slot = {} if handover else None
        y = ops.linear(oo, ww, None, ct=L.BF16, residual=xx, drop=drop, masked_grad=slot)
        h, y2 = ops.rmsnorm_res(y, nn_, 1e-6, grad_drop=(drop, slot) if handover else None)
        loss = (h * gy).sum() + (y2 * gy).sum() * 0.5
        if second_consumer:
            loss = loss + (y * gy).sum() * 0.25
        loss.backward()
        return [t.grad.clone() for t in (oo, xx, ww, nn_)], slot

    ref, _ = run(False)
    got, slot = run(True)
    for a, b in zip(got[:3], ref[:3]):     # d o, d residual, d W: everything downstream of the masked gradient
        assert torch.equal(a, b)
    assert torch.allclose(got[3], ref[3], rtol=1e-5, atol=1e-5)   # the norm weight's gradient: row-block atomics, order-dependent last bits
    if second_consumer:
        assert slot and "g" in slot, "the accumulated gradient has another address: the slot must stay unconsumed (fallback ran)"
    else:
        assert not slot, "the hand-over slot must have been consumed"
