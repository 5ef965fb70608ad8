"""Shared helpers for the parity tests: golden-fixture access, the oracle driver, comparison metrics."""
from __future__ import annotations

import ast
import glob
import os

import numpy as np
import torch

from oracle import pq3d_oracle as O
from pq3d_amd import synth
from pq3d_amd.model import Query3DUnified, make_cfg

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MAX_FULL, MAX_GRAD, MAX_TRAIN = 8192, 1024, 256  # must match tests/golden/make_golden.py


def fixtures(prefix=""):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def model_fixtures():
    """Fixtures made by make_golden.run_model_case (whole Query3DUnified forward/backward)."""
    return [f for f in fixtures() if f.startswith(("F1_", "F2_", "F4_", "F4b_", "F5_", "F15_", "F20_"))]


def load_fixture(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    args = ast.literal_eval(str(z["meta/args"])) if "meta/args" in z else {}
    return z, args


def compress(t: torch.Tensor, cap: int = MAX_FULL):
    a = t.detach().double().cpu().numpy()
    fin = np.isfinite(a)
    af = np.where(fin, a, 0.0)
    flat = a.reshape(-1)
    stride = max(1, -(-flat.size // cap))
    return {"sample": flat[::stride].astype(np.float32), "sum": af.sum(), "l2": np.sqrt((af ** 2).sum()),
            "ninf": int((~fin).sum()), "shape": tuple(a.shape)}


def check_against(z, key, t, atol, rtol, cap=MAX_FULL, what=""):
    """Compare tensor ``t`` with the compressed golden entry ``key``; returns max abs error of the sample."""
    c = compress(t, cap)
    g = z[key + "/sample"]
    assert tuple(z[key + "/shape"]) == c["shape"], f"{what}{key}: shape {c['shape']} vs {tuple(z[key + '/shape'])}"
    assert int(z[key + "/ninf"]) == c["ninf"], f"{what}{key}: non-finite count {c['ninf']} vs {int(z[key + '/ninf'])}"
    fin = np.isfinite(g)
    assert np.array_equal(fin, np.isfinite(c["sample"])), f"{what}{key}: non-finite pattern differs"
    scale = max(1.0, float(np.abs(g[fin]).max()) if fin.any() else 1.0)
    err = float(np.abs(g[fin] - c["sample"][fin]).max()) if fin.any() else 0.0
    assert err <= atol + rtol * scale, f"{what}{key}: max abs err {err:.3e} (scale {scale:.3e})"
    l2 = float(z[key + "/l2"])
    assert abs(c["l2"] - l2) <= (atol + rtol * max(l2, 1.0)) * 10, f"{what}{key}: l2 {c['l2']} vs {l2}"
    return err


def loss_weight(name, shape, seed=99):
    return synth.synth_tensor("lossw." + name, shape, seed) * 20.0


def model_case(args, device="cpu"):
    """Rebuild (cfg, our model, synthetic state dict, data_dict) of a `run_model_case` fixture."""
    kw = {k: args[k] for k in ("use_self_mask", "num_blocks", "dim_loc", "C", "foc", "drop_test", "offline_attn",
                               "skip_pred", "activation") if k in args}
    kw.setdefault("C", 21)
    d = args["d"]
    d_in = args.get("d_in") or {m: d for m in args["memories"] if m != "prompt"}
    cfg = make_cfg(d=d, H=args["H"], L=args["L"], memories=args["memories"], heads=args["heads"], d_in=d_in,
                   spatial=args["spatial"], structure=args["structure"], ground_hidden=d // 2 * 3 // 3, **kw)
    model = Query3DUnified(cfg, compute="fp32")
    model.eval()  # the fixtures were generated in eval mode (drop_memories_test applies, dropout off)
    sd = synth.fill_module(model, args["seed"])
    dd = synth.synth_data_dict(args["B"], args["Ns"], args["Nq"], d_in,
                               seed=args["data_seed"], memories=args["memories"],
                               query_valid_min=args.get("query_valid_min"), loc_dim=args.get("dim_loc", 3))
    if args.get("prompt_loc"):   # F20: location prompts through Query3DUnified.prompt_encoder
        dd.update(synth.prompt_loc_inputs(args["B"], args["prompt_loc"], seed=args["data_seed"] + 11))
    if args.get("offline_attn"):
        r = np.random.default_rng(args["data_seed"] + 7)
        om = r.random((args["B"], args["Nq"], args["Ns"])) < 0.6
        om[:, 1, :] = True
        dd["offline_attn_mask"] = torch.from_numpy(om)
    return cfg, model, sd, dd


def oracle_cfg(args):
    return dict(memories=args["memories"], heads=args["heads"], hidden_size=args["d"], dim_loc=args.get("dim_loc", 3),
                num_heads=args["H"], num_layers=args["L"], structure=args["structure"],
                spatial_selfattn=args["spatial"], use_self_mask=args.get("use_self_mask", False),
                num_blocks=args.get("num_blocks", 1), drop_memories_test=args.get("drop_test", ()),
                use_offline_attn_mask=args.get("offline_attn", False),
                skip_query_encoder_mask_pred=args.get("skip_pred", False), filter_out_classes=list(args.get("foc", ())),
                activation=args.get("activation", "relu"), training=args.get("training", False))


def synthetic_loss(out, heads, last_query):
    """The synthetic scalar loss of make_golden.run_model_case (stands in for optim/)."""
    loss = 0.0
    if "ground" in heads:
        gl = out["ground_logits"]
        gl = torch.where(torch.isfinite(gl), gl, torch.zeros_like(gl))
        loss = loss + (gl * loss_weight("ground", gl.shape).to(gl.device)).mean()
    if "mask" in heads:
        for i, (c, m) in enumerate(zip(out["predictions_class"], out["predictions_mask"])):
            cf = torch.where(torch.isfinite(c), c, torch.zeros_like(c))
            loss = loss + (cf * loss_weight(f"cls{i}", c.shape).to(c.device)).mean() \
                + (m.clamp(min=-50.0) * loss_weight(f"mask{i}", m.shape).to(m.device)).mean()
    return loss + (last_query * loss_weight("query", last_query.shape).to(last_query.device)).mean()


def run_oracle(args, sd, dd, grads=True, emulate=None):
    """emulate=torch.bfloat16 rounds every matmul operand like the HIP bf16 path does (see oracle.operand_rounding)."""
    if emulate is not None:
        with O.operand_rounding(emulate):
            return run_oracle(args, sd, dd, grads)
    sdo = {k: v.clone().requires_grad_(grads and v.dtype.is_floating_point and not k.endswith("gauss_B"))
           for k, v in sd.items()}
    collect = []
    ddc = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in dd.items()}
    out = O.query3d_unified_forward(sdo, oracle_cfg(args), ddc, collect=collect)
    loss = synthetic_loss(out, args["heads"], collect[-1])
    g = {}
    if grads:
        loss.backward()
        g = {k: v.grad for k, v in sdo.items() if v.grad is not None}
    return out, collect, loss, g
