"""Shared runners of fixture F21 (pre-norm layers, 'bias' spatial fusion, GroundHeadV1): the oracle side and the HIP side."""
import torch

from oracle import pq3d_oracle as O
from pq3d_amd import modules as M
from pq3d_amd import synth
from tests import util

CASES = (("pre_spatial_seq", True, "sequential"), ("pre_plain_par", False, "parallel"))


def layer_modules(d, H, spatial, structure, seed, compute="fp32"):
    layer = M.QueryEncoderLayer(d, H, ["voxel", "mv"], dropout=0.1, prenorm=True, spatial_selfattn=spatial, structure=structure)
    sd = synth.fill_module(layer, seed)
    M.set_compute(layer, compute)
    return layer, sd


def oracle_layer(sd, v, H, spatial, structure):
    sdo = {k: t.clone().requires_grad_(t.dtype.is_floating_point) for k, t in sd.items()}
    q = v["query"].clone().requires_grad_(True)
    input_dict = {"query": (q, v["qpad"], v["qpos"])}
    for m in ("voxel", "mv"):
        input_dict[m] = [v["feats"][m], v["pad"], v["fpos"]]
    pl = O.calc_pairwise_locs(v["centers"]) if spatial else None
    y = O.query_encoder_layer(sdo, "", q, input_dict, pl, memories=["voxel", "mv"], H=H, structure=structure,
                              spatial_selfattn=spatial, prenorm=True)
    (y * util.loss_weight("query", y.shape)).mean().backward()
    return y.detach(), q.grad, {k: t.grad for k, t in sdo.items() if t.grad is not None}


def hip_layer(layer, v, spatial, dev="cuda"):
    layer.to(dev).eval()
    q = v["query"].to(dev).requires_grad_(True)
    input_dict = {"query": (q, v["qpad"].to(dev), v["qpos"].to(dev))}
    for m in ("voxel", "mv"):
        input_dict[m] = [v["feats"][m].to(dev), v["pad"].to(dev), v["fpos"].to(dev)]
    pl = M.calc_pairwise_locs(v["centers"].to(dev)) if spatial else None
    y = layer(q, input_dict, pl)
    (y * util.loss_weight("query", y.shape).to(dev)).mean().backward()
    return y.detach(), q.grad, {n: p.grad for n, p in layer.named_parameters() if p.grad is not None}
