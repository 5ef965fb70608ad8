"""Per-sublayer parity cases at BASELINE config-2 shapes (B 8, N_q 100, N_seg 1024, d 256, H 8): each decoder sublayer
(cross-attention layer, spatial self-attention layer, FFN layer, mask-head call) is fed IDENTICAL inputs on the HIP
path and in the oracle (run in float64 on the CPU = the fp32 reference arithmetic without its own round-off), forward
and backward.  Shared by tests/test_gpu_sublayer_parity.py (asserts) and tools/sublayer_parity.py (report under
profiles/).  Test infrastructure: imports the oracle."""
from __future__ import annotations

import numpy as np
import torch

from oracle import pq3d_oracle as O
from pq3d_amd import modules as M
from pq3d_amd import synth

C2 = dict(B=8, Nq=100, Ns=1024, d=256, H=8)


def _randn(name, shape, seed=5):
    r = synth._rng(seed, "sublayer." + name)
    return torch.from_numpy(r.standard_normal(tuple(shape)).astype(np.float32))


def _kpm(B, Ns, seed=5):
    r = synth._rng(seed, "sublayer.kpm")
    vl = r.integers(Ns // 2, Ns + 1, size=B)
    vl[0] = Ns
    return torch.from_numpy(np.arange(Ns)[None, :] >= vl[:, None])   # True = padded (PyTorch convention)


def metrics(a: torch.Tensor, ref: torch.Tensor):
    """(max|err| / max|ref|, relative L2, cosine) over the finite entries of the reference."""
    a, ref = a.detach().double().cpu().flatten(), ref.detach().double().cpu().flatten()
    fin = torch.isfinite(ref) & (ref > -1e5)
    a, ref = a[fin], ref[fin]
    if ref.numel() == 0:
        return 0.0, 0.0, 1.0
    e = a - ref
    rn, an = float(ref.norm()), float(a.norm())
    return (float(e.abs().max()) / max(float(ref.abs().max()), 1e-30), float(e.norm()) / max(rn, 1e-30),
            float((a * ref).sum()) / max(rn * an, 1e-30))


class Case:
    """One sublayer: a HIP module + its oracle function + named inputs; run() returns outputs and gradients."""

    def __init__(self, name, module, oracle_fn, inputs, grad_inputs, shapes=C2):
        self.name, self.module, self.oracle_fn = name, module, oracle_fn
        self.inputs, self.grad_inputs = inputs, list(grad_inputs)
        self.sd = synth.fill_module(module, 3)
        module.eval()

    def _loss_w(self, k, shape):
        return _randn(f"{self.name}.lossw{k}", shape, seed=11)

    def run_hip(self, compute, device="cuda"):
        M.set_compute(self.module, compute)
        self.module.to(device)
        self.module.zero_grad(set_to_none=True)
        inp = {k: (v.to(device).requires_grad_(k in self.grad_inputs) if torch.is_tensor(v) and v.is_floating_point()
                   else (v.to(device) if torch.is_tensor(v) else v)) for k, v in self.inputs.items()}
        outs = self.hip_forward(inp)
        loss = sum((o.float() * self._loss_w(i, o.shape).to(device)).sum() for i, o in enumerate(outs))
        loss.backward()
        g = {k: inp[k].grad for k in self.grad_inputs}
        g.update({"param." + n: p.grad for n, p in self.module.named_parameters() if p.grad is not None})
        return [o.detach() for o in outs], g

    def run_oracle(self, emulate=None, dtype=torch.float64):
        sd = {k: v.to(dtype).requires_grad_(v.is_floating_point()) if v.is_floating_point() else v
              for k, v in self.sd.items()}
        inp = {k: (v.to(dtype).requires_grad_(k in self.grad_inputs) if torch.is_tensor(v) and v.is_floating_point()
                   else v) for k, v in self.inputs.items()}
        if emulate is not None:
            with O.operand_rounding(emulate):
                outs = self.oracle_forward(sd, inp)
        else:
            outs = self.oracle_forward(sd, inp)
        loss = sum((o * self._loss_w(i, o.shape).to(dtype)).sum() for i, o in enumerate(outs))
        loss.backward()
        g = {k: inp[k].grad for k in self.grad_inputs}
        g.update({"param." + n: v.grad for n, v in sd.items() if torch.is_tensor(v) and v.grad is not None})
        return [o.detach() for o in outs], g


class CrossAttnCase(Case):
    def __init__(self, shapes=C2, mask3d=False):
        B, Nq, Ns, d, H = (shapes[k] for k in ("B", "Nq", "Ns", "d", "H"))
        mod = M.CrossAttentionLayer(d, H, dropout=0.0, batch_first=True)
        inputs = dict(tgt=_randn("ca.tgt", (B, Nq, d)), query_pos=_randn("ca.qpos", (B, Nq, d)),
                      memory=_randn("ca.mem", (B, Ns, d)), pos=_randn("ca.pos", (B, Ns, d)), kpm=_kpm(B, Ns))
        self.H = H
        super().__init__("cross_attn", mod, None, inputs, ["tgt", "query_pos", "memory", "pos"], shapes)

    def hip_forward(self, i):
        return [self.module(i["tgt"], i["memory"], memory_key_padding_mask=i["kpm"], pos=i["pos"],
                            query_pos=i["query_pos"])]

    def oracle_forward(self, sd, i):
        return [O.cross_attention_layer(sd, "", i["tgt"], i["memory"], self.H, memory_key_padding_mask=i["kpm"],
                                        pos=i["pos"], query_pos=i["query_pos"])]


class SpatialSelfAttnCase(Case):
    def __init__(self, shapes=C2):
        B, Nq, d, H = (shapes[k] for k in ("B", "Nq", "d", "H"))
        mod = M.SpatialSelfAttentionLayer(d, H, dropout=0.0, batch_first=True)
        r = synth._rng(5, "sublayer.centers")
        centers = torch.from_numpy(r.uniform(0, 4, (B, Nq, 3)).astype(np.float32))
        inputs = dict(tgt=_randn("sa.tgt", (B, Nq, d)), query_pos=_randn("sa.qpos", (B, Nq, d)),
                      pl=O.calc_pairwise_locs(centers), qmask=torch.zeros(B, Nq, dtype=torch.bool))
        self.H = H
        super().__init__("spatial_self_attn", mod, None, inputs, ["tgt", "query_pos"], shapes)

    def hip_forward(self, i):
        return [self.module(i["tgt"], tgt_key_padding_mask=i["qmask"], query_pos=i["query_pos"], pairwise_locs=i["pl"])]

    def oracle_forward(self, sd, i):
        return [O.spatial_self_attention_layer(sd, "", i["tgt"], self.H, i["pl"].to(i["tgt"].dtype),
                                               tgt_key_padding_mask=i["qmask"], query_pos=i["query_pos"])]


class FFNCase(Case):
    def __init__(self, shapes=C2):
        B, Nq, d = (shapes[k] for k in ("B", "Nq", "d"))
        mod = M.FFNLayer(d, 2048, dropout=0.0, activation="relu")
        super().__init__("ffn", mod, None, dict(tgt=_randn("ffn.tgt", (B, Nq, d))), ["tgt"], shapes)

    def hip_forward(self, i):
        return [self.module(i["tgt"])]

    def oracle_forward(self, sd, i):
        return [O.ffn_layer(sd, "", i["tgt"], "relu")]


class MaskHeadCase(Case):
    def __init__(self, shapes=C2, n_mem=3, C=201):
        B, Nq, Ns, d = (shapes[k] for k in ("B", "Nq", "Ns", "d"))
        mod = M.MaskHeadSegLevel(None, d, C, memories_for_match=["voxel", "mv", "pc"][:n_mem], filter_out_classes=[0, 2],
                                 dropout=0.0)
        kpm = _kpm(B, Ns)
        inputs = dict(query=_randn("mh.q", (B, Nq, d)), kpm=kpm)
        for m in range(n_mem):
            f = _randn(f"mh.feat{m}", (B, Ns, d))
            f[kpm] = 0.0
            inputs[f"feat{m}"] = f
        self.n_mem = n_mem
        super().__init__("mask_head", mod, None, inputs, ["query"] + [f"feat{m}" for m in range(n_mem)], shapes)

    def hip_forward(self, i):
        sf = [(i[f"feat{m}"], i["kpm"], None) for m in range(self.n_mem)]
        cls, mlog, amask = self.module(i["query"], sf, i["kpm"])
        self.amask = amask
        return [cls, mlog]

    def oracle_forward(self, sd, i):
        sf = [(i[f"feat{m}"], i["kpm"], None) for m in range(self.n_mem)]
        cls, mlog, amask = O.mask_head_seg_level(sd, "", i["query"], sf, i["kpm"], filter_out_classes=[0, 2])
        self.amask_ref = amask
        cls = torch.where(torch.isfinite(cls), cls, torch.zeros_like(cls))   # -inf columns carry no gradient
        return [cls, mlog]

    def run_hip(self, compute, device="cuda"):
        outs, g = super().run_hip(compute, device)
        outs[0] = torch.where(torch.isfinite(outs[0]), outs[0], torch.zeros_like(outs[0]))
        return outs, g

    def _loss_w(self, k, shape):
        w = super()._loss_w(k, shape)
        if k == 1:   # mask logits: padded segments hold the -1e6 fill -> no loss weight there
            w = w * (~self.inputs["kpm"])[..., None]
        return w


CASES = {"cross_attn": CrossAttnCase, "spatial_self_attn": SpatialSelfAttnCase, "ffn": FFNCase, "mask_head": MaskHeadCase}


def compare(case: Case, compute: str, ref=None):
    """Returns {'out': [(max/scale, relL2, cos) per output], 'grad': {name: (max/scale, relL2, cos)}} of the HIP path
    against the float64 oracle (or a given (outs, grads) reference)."""
    ro, rg = ref if ref is not None else case.run_oracle()
    ho, hg = case.run_hip(compute)
    res = {"out": [metrics(a, b) for a, b in zip(ho, ro)], "grad": {}}
    for k, v in rg.items():
        if k in hg and hg[k] is not None:
            res["grad"][k] = metrics(hg[k], v)
    return res
