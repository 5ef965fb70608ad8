"""Frozen PointNet++ tokenizer (object_encoder.py:22-28) on M object clouds of P points: the row-GEMM HIP path
(pq3d_amd/pointnetpp.py) vs the reference's channel-first layout (1x1 Conv2d + BatchNorm2d + max_pool2d on stock
PyTorch-ROCm, fed by the same HIP point-set operators).  Usage: python tools/bench_pointnetpp.py [--clouds 640] [--points 1024]"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pq3d_amd import pointnet2 as P2, synth  # noqa: E402
from pq3d_amd.pointnetpp import POINTNETPP_TOKENIZER, PointNetPP  # noqa: E402


def channel_first_forward(net, pc):
    """pointnet2_modules.py:23-70 verbatim in structure, torch layers on [B, C, npoint, nsample]."""
    xyz, feats = pc[..., :3].contiguous(), pc[..., 3:].transpose(1, 2).contiguous()
    for sa in net.encoder:
        if sa.npoint is not None:
            new_xyz = P2.gather_operation(xyz.transpose(1, 2).contiguous(), P2.furthest_point_sample(xyz, sa.npoint)).transpose(1, 2).contiguous()
            x = P2.QueryAndGroup(sa.radius, sa.nsample)(xyz, new_xyz, feats)
        else:
            new_xyz = None
            x = P2.GroupAll()(xyz, None, feats)
        for layer in sa.mlps[0]:
            bn = layer.bn.bn
            x = F.relu(F.batch_norm(F.conv2d(x, layer.conv.weight), bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps))
        feats = F.max_pool2d(x, kernel_size=[1, x.size(3)]).squeeze(-1)
        xyz = new_xyz
    return net.fc(feats.view(feats.size(0), -1))


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / iters * 1e3, 3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clouds", type=int, default=640)
    ap.add_argument("--points", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    net = PointNetPP(**{k: [list(x) if isinstance(x, list) else x for x in v] for k, v in POINTNETPP_TOKENIZER.items()})
    synth.fill_module(net, 0)
    net.cuda().eval().requires_grad_(False)
    pc = synth.pointcloud_inputs(a.clouds, a.points).cuda()
    out = {"workload": f"PointNet++ tokenizer, {a.clouds} clouds x {a.points} points x 6", "data": "synthetic"}
    with torch.no_grad():
        ref = channel_first_forward(net, pc)
        out["channel_first_torch_fp32_ms"] = timeit(lambda: channel_first_forward(net, pc), a.iters)
        for compute in ("fp32", "bf16"):
            net.compute = compute
            got = net(pc)
            out[f"hip_rows_{compute}_ms"] = timeit(lambda: net(pc), a.iters)
            out[f"hip_rows_{compute}_max_rel_err_vs_channel_first"] = float((got - ref).abs().max() / ref.abs().max())
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            net(pc); s.synchronize()
            with torch.cuda.graph(g, stream=s):
                net(pc)
        out["hip_rows_bf16_graph_ms"] = timeit(g.replay, a.iters)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
