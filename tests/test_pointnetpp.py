"""The frozen PointNet++ tokenizer (modules/layers/pointnet.py): oracle and HIP network vs fixture F12 -- the reference's
own module tree / SharedMLP / fc objects around the oracle's point-set operators (see make_golden.run_pointnetpp_case)."""
import ast

import numpy as np
import pytest
import torch

from oracle import pointnet2_oracle as po
from pq3d_amd import synth
from tests import util

SPEC = dict(sa_n_points=[32, 16, None], sa_n_samples=[32, 32, None], sa_radii=[0.2, 0.4, None])


def _net():
    from pq3d_amd.pointnetpp import POINTNETPP_TOKENIZER, PointNetPP
    return PointNetPP(**{k: [list(x) if isinstance(x, list) else x for x in v] for k, v in POINTNETPP_TOKENIZER.items()})


def test_module_tree_matches_reference_and_oracle_matches_f12():
    z, a = util.load_fixture("F12_pointnetpp")
    net = _net()
    want = ast.literal_eval(str(z["meta/keys"]))
    assert [(k, tuple(v.shape)) for k, v in net.state_dict().items()] == want      # checkpoint-compatible, same order
    sd = synth.fill_module(net, a["seed"])
    assert abs(synth.state_checksum(sd) - float(z["meta/weights_checksum"])) < 1e-6 * float(z["meta/weights_checksum"])
    trace = {}
    out = po.pointnetpp_forward({k: v.numpy() for k, v in sd.items()}, z["pc"], **SPEC, trace=trace)
    for i in range(3):
        np.testing.assert_allclose(trace[f"pooled/{i}"], z[f"pooled/{i}"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(out, z["out"], rtol=1e-5, atol=1e-5)
    # the fill rule is exercised: some radius-0.2 balls hold fewer than nsample points
    ball = z["oracle/ball/0"]
    assert (ball[:, :, -1] == ball[:, :, 0]).any() and (ball[:, :, -1] != ball[:, :, 0]).any()


@pytest.mark.gpu
@pytest.mark.parametrize("compute", ["fp32", "bf16"])
def test_hip_tokenizer_matches_f12(compute):
    from pq3d_amd import pointnet2 as P2
    z, a = util.load_fixture("F12_pointnetpp")
    net = _net()
    synth.fill_module(net, a["seed"])
    net.compute = compute
    net.cuda().eval().requires_grad_(False)
    pc = torch.from_numpy(z["pc"]).cuda()
    # the point-set kernels reproduce the oracle's indices on this input (bit-exact integer work)
    xyz = pc[..., :3].contiguous()
    fps = P2.furthest_point_sample(xyz, 32)
    assert np.array_equal(fps.cpu().numpy(), z["oracle/fps/0"])
    new_xyz = torch.gather(xyz, 1, fps.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    assert np.array_equal(P2.ball_query(0.2, 32, xyz, new_xyz).cpu().numpy(), z["oracle/ball/0"])
    out = net(pc)
    ref = z["out"]
    tol = 2e-5 if compute == "fp32" else 3e-2
    err = np.abs(out.cpu().numpy() - ref).max() / max(1.0, np.abs(ref).max())
    assert out.shape == ref.shape and err < tol, err
    # chunked execution (bounded row buffers) gives the same rows
    net.CHUNK = 3
    assert torch.equal(net(pc), out)


@pytest.mark.gpu
def test_object_encoder_with_pointnet_backbone_and_refusals():
    from pq3d_amd.modules import ObjectEncoder
    z, a = util.load_fixture("F12_pointnetpp")
    enc = ObjectEncoder(None, backbone="pointnet++", freeze_backbone=True, input_feat_size=768, hidden_size=64,
                        use_projection=True, use_cls_head=False, dropout=0.0)
    enc.compute = "fp32"
    sd = synth.fill_module(enc.backbone, a["seed"])
    enc.cuda().train()                                   # freeze_bn keeps the backbone's BatchNorm in eval (object_encoder.py:56-58)
    pc = torch.from_numpy(z["pc"]).cuda()
    emb = enc(pc.view(2, 2, *pc.shape[1:]))
    assert emb.shape == (2, 2, 64) and emb.requires_grad   # projection trains, tokenizer does not
    with torch.no_grad():
        tok = enc.backbone(pc)
    np.testing.assert_allclose(tok.cpu().numpy(), z["out"], rtol=0, atol=2e-5 * max(1.0, np.abs(z["out"]).max()))
    with pytest.raises(NotImplementedError):
        ObjectEncoder(None, backbone="pointnet++", freeze_backbone=False)
    net = _net().cuda()
    with pytest.raises(NotImplementedError):             # BatchNorm in training mode
        net.train()(pc)
    with pytest.raises(NotImplementedError):             # trainable parameters with grad enabled
        net.eval()(pc)
    with pytest.raises(RuntimeError):
        net.requires_grad_(False)(pc.cpu())


@pytest.mark.gpu
def test_group_rows_and_maxpool_edge_cases():
    """Row grouping without point features (xyz only), GroupAll (no indices, no centring), bf16 rows, and the max-pool
    over samples, against direct torch indexing."""
    from pq3d_amd import pointnetpp as PP
    g = torch.Generator().manual_seed(0)
    B, N, npnt, ns = 3, 50, 7, 5
    xyz = torch.randn(B, N, 3, generator=g).cuda()
    new_xyz = torch.randn(B, npnt, 3, generator=g).cuda()
    idx = torch.randint(0, N, (B, npnt, ns), generator=g).int().cuda()
    rows = PP.group_rows(xyz, new_xyz, None, 0, idx, npnt, ns, torch.float32)          # xyz only: Kp = 8
    assert rows.shape == (B * npnt * ns, 8)
    ref = torch.gather(xyz, 1, idx.long().view(B, -1, 1).expand(-1, -1, 3)).view(B, npnt, ns, 3) - new_xyz[:, :, None, :]
    assert torch.equal(rows[:, :3].view(B, npnt, ns, 3), ref) and float(rows[:, 3:].abs().max()) == 0.0
    feats = torch.randn(B, N, 9, generator=g).cuda()                                     # strided view: 5 of 9 columns
    rows_all = PP.group_rows(xyz, None, feats[..., 2:7], 5, None, 1, N, torch.bfloat16)  # GroupAll, bf16 rows
    assert rows_all.shape == (B * N, 8)
    want = torch.cat([xyz, feats[..., 2:7]], -1).view(B * N, 8).bfloat16()
    assert torch.equal(rows_all, want)
    pooled = PP.group_maxpool(rows_all, B, N)
    assert torch.equal(pooled, want.view(B, N, 8).max(1).values)
