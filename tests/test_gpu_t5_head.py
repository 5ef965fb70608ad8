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
