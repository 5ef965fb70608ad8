"""Query3DUnified: the hot-path top level (model/query3d_unified.py:30-238) on the HIP modules.

``cfg`` is any attribute/dict-style config with the reference's layout (``cfg.model.memories``,
``cfg.model.unified_encoder.args`` ...; OmegaConf DictConfig, a plain nested dict or ``Cfg`` below).
Modules are looked up by the reference's class *names* in ``REGISTRY`` (modules/build.py:24-31)."""
from __future__ import annotations

from copy import copy
from functools import partial

import torch
import torch.nn as nn

from . import modules as M
from . import ops


class Cfg(dict):
    """Nested attribute dict with .get (stands in for OmegaConf's DictConfig)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in {**(d or {}), **kw}.items():
            self[k] = Cfg(v) if isinstance(v, dict) and not isinstance(v, Cfg) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


REGISTRY = {c.__name__: c for c in (M.QueryMaskEncoder, M.QueryEncoder, M.MaskHeadSegLevel, M.GroundHead, M.GroundHeadV1,
                                    M.ObjectEncoder, M.T5, M.PCDMask3DSegLevelEncoder)}


def _to_dict(c):
    try:
        from omegaconf import OmegaConf  # type: ignore
        if OmegaConf.is_config(c):
            return OmegaConf.to_container(c, resolve=True)
    except ImportError:
        pass
    return {k: (_to_dict(v) if isinstance(v, dict) else v) for k, v in dict(c).items()}


def build_module_by_name(cfg):
    """modules/build.py:24-31."""
    if cfg.name not in REGISTRY:
        raise NotImplementedError(f"Unknown module: {cfg.name}")
    kwargs = _to_dict(cfg.args) if hasattr(cfg, "args") or "args" in cfg else {}
    return REGISTRY[cfg.name](cfg, **kwargs)


def no_decay_param_group(parameters, lr, name=""):
    """optim/utils.py:1-18."""
    no_decay = ["bias", "LayerNorm.bias", "LayerNorm.weight"]
    decay_params, no_decay_params = [], []
    for n, p in parameters:
        if not p.requires_grad:
            continue
        (no_decay_params if any(nd in n for nd in no_decay) else decay_params).append(p)
    return [{"params": decay_params, "name": name, "weight_decay": 0.01, "lr": lr},
            {"params": no_decay_params, "name": name, "weight_decay": 0.0, "lr": lr}]


class Query3DUnified(nn.Module):
    """model/query3d_unified.py:30-238.  Supported inputs: offline voxel features (``use_offline_voxel_fts``),
    mv / pc segment features, and the prompt memory either pre-encoded (``data_dict['prompt_feat']`` [B,T,d]) or through
    ``prompt_encoder`` (``prompt`` / ``prompt_type`` / ``prompt_pad_masks``: 'loc' prompts on the HIP encoders, 'txt' prompts
    through a caller-supplied ``txt_encoder`` -- the CLIP text encoder itself is out of scope).  Heads: 'mask', 'ground', 'generation' (input_proj on the HIP kernels, the HF T5
    decoder body on stock PyTorch-ROCm ops, §8f-3)."""

    def __init__(self, cfg, compute: str = "bf16"):
        super().__init__()
        self.cfg = cfg
        self.memories = list(cfg.model.memories)
        self.heads = list(cfg.model.heads)
        self.use_offline_voxel_fts = cfg.model.get("use_offline_voxel_fts", False)
        self.use_offline_attn_mask = cfg.model.get("use_offline_attn_mask", False)
        self.inputs = self.memories[:]
        self.pairwise_rel_type = cfg.model.obj_loc.pairwise_rel_type
        self.spatial_dim = cfg.model.obj_loc.spatial_dim
        self.num_heads = cfg.model.unified_encoder.args.num_attention_heads
        self.skip_query_encoder_mask_pred = cfg.model.get("skip_query_encoder_mask_pred", False)
        for inp in self.inputs:
            if inp == "prompt":
                continue  # text encoder out of scope: prompt memory arrives pre-encoded
            # voxel without use_offline_voxel_fts: the reference runs its MinkowskiEngine backbone inside the model
            # (query3d_unified.py:146-152); here the backbone is outside (out of scope) and the voxel encoder is the
            # post-backbone part (M.PCDMask3DSegLevelEncoder) fed the backbone's per-level features
            enc = build_module_by_name(cfg.model.get(inp + "_encoder"))
            if hasattr(enc, "_drop_base"):   # one dropout-site range per encoder instance
                enc._drop_base = M.DROP_BASE_OBJ_ENC + (self.inputs.index(inp) << 12)
            setattr(self, inp + "_encoder", enc)
        self.dim_loc = cfg.model.obj_loc.dim_loc
        self.hidden_size = hidden_size = cfg.model.hidden_size
        if self.dim_loc > 3:
            self.coord_encoder = nn.Sequential(nn.Linear(3, hidden_size), nn.LayerNorm(hidden_size))
            self.box_encoder = nn.Sequential(nn.Linear(3, hidden_size), nn.LayerNorm(hidden_size))
        else:
            self.coord_encoder = M.CoordinateEncoder(hidden_size)
        self.unified_encoder = build_module_by_name(cfg.model.unified_encoder)
        for head in self.heads:
            setattr(self, head + "_head", build_module_by_name(cfg.model.get(head + "_head")))
        self.compute = compute
        self._coef_cache = {}
        self._zeros_cache = {}
        M.set_compute(self, compute)

    @property
    def ct(self):
        return M.CT[self.compute]

    def _pos(self, locs, coord_min, coord_max, box_times=1):
        if self.dim_loc > 3:
            ct = self.ct
            ct = ops.small_ct(ct)
            c = ops.linear(locs[:, :, :3].contiguous(), self.coord_encoder[0].weight, self.coord_encoder[0].bias, ct=ct)
            b = ops.linear(locs[:, :, 3:6].contiguous(), self.box_encoder[0].weight, self.box_encoder[0].bias, ct=ct)
            key = (locs.shape[0], box_times, locs.device)
            if key not in self._coef_cache:  # cached: no H2D copy inside a captured HIP graph
                self._coef_cache[key] = torch.tensor([[1.0] * key[0], [float(box_times)] * key[0]], device=locs.device)
            coef = self._coef_cache[key]
            return ops.add_layernorm(None, [c, b], [self.coord_encoder[1].weight, self.box_encoder[1].weight],
                                     [self.coord_encoder[1].bias, self.box_encoder[1].bias],
                                     eps=self.coord_encoder[1].eps, coef=coef)
        return self.coord_encoder(locs[:, :, :3], input_range=[coord_min, coord_max])

    PROMPT_TYPE = {"txt": 1, "image": 2, "loc": 3}   # data/datasets/constant.py:628-631 (PromptType)

    def prompt_encoder(self, data_dict):
        """query3d_unified.py:80-108: encode the prompt of every scene by its `prompt_type`.  'loc' prompts (the first dim_loc
        entries of the prompt row are a location) run through the coordinate / box encoders on the HIP kernels and occupy token 0
        alone (`mask[:, 1:] = False`); 'txt' prompts are handed to ``self.txt_encoder`` -- the reference's CLIP text encoder is out
        of scope, so a caller that trains with text prompts assigns its own module there (called as
        ``encoder(token_ids.long(), pad_mask)`` -> [n, T, d], exactly as the reference calls it).  Like the reference, the
        scenes' `prompt_pad_masks` rows are updated IN PLACE; returns (prompt_feat [B, T, d], key-padding mask = ~pad mask)."""
        prompt, ppm, ptype = data_dict["prompt"], data_dict["prompt_pad_masks"], data_dict["prompt_type"]
        prompt_feat = torch.zeros(prompt.shape + (self.hidden_size,), device=prompt.device)
        for kind in ("txt", "loc"):
            idx = ptype == self.PROMPT_TYPE[kind]
            if int(idx.sum()) == 0:
                continue
            inp, mask = prompt[idx], ppm[idx]
            if kind == "txt":
                enc = getattr(self, "txt_encoder", None)
                if enc is None:
                    raise NotImplementedError("text prompts need a text encoder: assign model.txt_encoder (the reference's CLIP "
                                              "encoder is outside the hot path, SURVEY section 2) or pass data_dict['prompt_feat']")
                feat = enc(inp.long(), mask)
            else:
                loc = inp[:, :self.dim_loc].float()
                if self.dim_loc > 3:
                    feat = self._pos(loc[:, None, :6].contiguous(), None, None)            # coord + box embedding, [n, 1, d]
                else:
                    feat = self._pos(loc[:, None, :3].contiguous(), data_dict["coord_min"][idx], data_dict["coord_max"][idx])
                mask[:, 1:] = False
            prompt_feat[idx] = feat.to(prompt_feat.dtype)      # [n, 1, d] broadcasts over the T token slots, as in the reference
            ppm[idx] = mask
        return prompt_feat, ppm.logical_not()

    def _pos_pair(self, query_locs, seg_locs, coord_min, coord_max):
        """CoordinateEncoder on queries and segments in ONE Linear+LN pass (same weights, rows concatenated)."""
        ce = self.coord_encoder
        B, Nq, Ns, d = query_locs.shape[0], query_locs.shape[1], seg_locs.shape[1], self.hidden_size
        if not hasattr(ce, "feat_proj"):
            return ce(query_locs[:, :, :3], [coord_min, coord_max]), ce(seg_locs[:, :, :3], [coord_min, coord_max])
        buf = ops.fourier_pair(query_locs[:, :, :3], seg_locs[:, :, :3], coord_min, coord_max, ce.pos_enc.gauss_B)
        y = M.linear_ln_forward(ce.feat_proj, buf, self.ct)
        yq, ys = ops.split_rows(y, B * Nq)
        return yq.view(B, Nq, d), ys.view(B, Ns, d)

    def _encode_scene_memories(self, data_dict):
        """ObjectEncoder projections of all scene memories; same-shape encoders share grouped launches."""
        names = [m for m in self.inputs if m in ("mv", "pc", "voxel") and not (m == "voxel" and not self.use_offline_voxel_fts)]
        encs = [getattr(self, m + "_encoder") for m in names]
        xs = [data_dict[m + "_seg_fts"] for m in names]
        same = len(names) > 1 and all(isinstance(e, M.ObjectEncoder) and e.use_projection and not hasattr(e, "cls_head")
                                      for e in encs) and len({tuple(x.shape) for x in xs}) == 1
        if not same:
            out = {}
            for m, e, x in zip(names, encs, xs):
                r = e(obj_feats=x) if m != "voxel" else e(x)
                out[m] = r[0] if isinstance(r, tuple) else r
            return out
        seqs = [e.input_feat_proj for e in encs]
        ys = ops.linear_ln_group(xs, [q[0].weight for q in seqs], [q[0].bias for q in seqs],
                                 [q[1].weight for q in seqs], [q[1].bias for q in seqs], ct=ops.small_ct(self.ct),
                                 eps=seqs[0][1].eps)   # split-bf16 in 'bf16' mode: see modules.linear_ln_forward
        if self.training:   # ObjectEncoder's output dropout (object_encoder.py:72-73), same sites as the per-encoder path
            ys = [ops.dropout(y, e._drop(e._head_ctx(y.device), ops.DROP_ENC_OUT, y.device)) for y, e in zip(ys, encs)]
        return dict(zip(names, ys))

    def forward(self, data_dict):
        input_dict = {}
        if self.training:
            M.begin_dropout_step(self, data_dict["query_locs"].device)
        # every pad mask ('True = valid' in data_dict) is inverted by ONE launch (ops.mask_not); equal-shape masks come back
        # as views of one stacked buffer, which the fused executor takes as the memories-stacked key-padding mask
        online_voxel = "voxel" in self.inputs and not self.use_offline_voxel_fts
        scene = [m for m in self.inputs if m in ("mv", "pc", "voxel") and not (m == "voxel" and online_voxel)]
        keys = [m + "_seg_pad_masks" for m in scene] + (["seg_pad_masks"] if (hasattr(self, "mask_head") or online_voxel) else [])
        inv = {}
        bool_ok = all(data_dict[k].dtype == torch.bool for k in keys + ["query_pad_masks"])
        if bool_ok:
            outs = ops.mask_not([data_dict[k] for k in keys] + [data_dict["query_pad_masks"]])
            mask = outs[-1]
            inv = dict(zip(keys, outs[:-1]))
            if len(scene) > 1 and len({tuple(data_dict[m + "_seg_pad_masks"].shape) for m in scene}) == 1:
                B_, Ns_ = data_dict[scene[0] + "_seg_pad_masks"].shape
                stacked = outs[0].new_empty(0).set_(outs[0].untyped_storage(), outs[0].storage_offset(),
                                                     (len(scene), B_, Ns_), (B_ * Ns_, Ns_, 1))
                input_dict["_stacked_scene_kpm"] = (stacked, scene)
        else:
            mask = data_dict["query_pad_masks"].logical_not()
        query_locs = data_dict["query_locs"][:, :, :self.dim_loc]
        coord_min, coord_max = data_dict["coord_min"], data_dict["coord_max"]
        if self.dim_loc > 3:
            query_pos = self._pos(query_locs, coord_min, coord_max)
            # NB dim_loc > 3: the reference adds the box embedding to fts_pos twice (query3d_unified.py:128,131-132)
            fts_pos = self._pos(data_dict["seg_center"], coord_min, coord_max, box_times=2)
        else:
            query_pos, fts_pos = self._pos_pair(query_locs, data_dict["seg_center"], coord_min, coord_max)
        zkey = (tuple(query_pos.shape), query_pos.device)
        if zkey not in self._zeros_cache:      # the learnable-query content starts at zero (query3d_unified.py:121): constant
            self._zeros_cache[zkey] = torch.zeros_like(query_pos)
        input_dict["query"] = (self._zeros_cache[zkey], mask, query_pos)
        enc_out = self._encode_scene_memories(data_dict)
        for inp in self.inputs:
            if inp == "prompt":
                if "prompt_feat" in data_dict:   # pre-encoded prompt memory (the text encoder ran outside)
                    feat, mask, pos = data_dict["prompt_feat"], data_dict["prompt_pad_masks"].logical_not(), None
                else:                            # query3d_unified.py:134-136
                    feat, mask = self.prompt_encoder(data_dict)
                    pos = None
            elif inp in ("mv", "pc"):
                feat = enc_out[inp]
                k = inp + "_seg_pad_masks"
                mask, pos = inv[k] if k in inv else data_dict[k].logical_not(), fts_pos
            elif inp == "voxel" and self.use_offline_voxel_fts:
                feat = enc_out[inp]
                mask = inv["voxel_seg_pad_masks"] if "voxel_seg_pad_masks" in inv else data_dict["voxel_seg_pad_masks"].logical_not()
                pos = fts_pos
            elif inp == "voxel":
                # query3d_unified.py:146-152 with the backbone's outputs supplied: multi-scale LIST of segment features
                feat = self.voxel_encoder(data_dict["voxel_pyramid"], data_dict["voxel2segment"],
                                          max_seg=data_dict["seg_center"].shape[1])
                mask = inv["seg_pad_masks"] if "seg_pad_masks" in inv else data_dict["seg_pad_masks"].logical_not()
                pos = fts_pos
            else:
                raise NotImplementedError(f"Unknow input type: {inp}")
            input_dict[inp] = [feat, mask, pos]
        offline_attn_masks = data_dict["offline_attn_mask"] if self.use_offline_attn_mask else None
        seg_fts_for_match = []
        for inp in self.inputs:
            if inp in ("voxel", "mv", "pc"):
                feats = copy(input_dict[inp][:])
                if isinstance(feats[0], list):
                    feats[0] = feats[0][-1]
                seg_fts_for_match.append(feats)
        if hasattr(self, "mask_head"):
            seg_masks = inv["seg_pad_masks"] if "seg_pad_masks" in inv else data_dict["seg_pad_masks"].logical_not()
            mask_head_partial = partial(self.mask_head, seg_fts_for_match=seg_fts_for_match, seg_masks=seg_masks,
                                        offline_attn_masks=offline_attn_masks,
                                        skip_prediction=self.skip_query_encoder_mask_pred)
            enc = self.unified_encoder
            fusable = getattr(enc, "fused", False) and enc.unified_encoder[0].structure == "parallel"
            if not fusable:
                # modular path: k_proj(seg feats) is layer-invariant -> project once, reuse in all mask-head calls
                mask_head_partial.keywords["keys"] = self.mask_head.project_keys(seg_fts_for_match)
        else:
            mask_head_partial = None
        if self.unified_encoder.spatial_selfattn:
            pairwise_locs = M.calc_pairwise_locs(query_locs[:, :, :3], None, pairwise_rel_type=self.pairwise_rel_type,
                                                 spatial_dist_norm=True, spatial_dim=self.spatial_dim)
        else:
            pairwise_locs = None

        query, predictions_class, predictions_mask = self.unified_encoder(input_dict, pairwise_locs, mask_head_partial)

        for head in self.heads:
            if head == "ground":
                logits = self.ground_head(query, data_dict["query_pad_masks"])
                data_dict["ground_logits"] = logits
                data_dict["og3d_logits"] = logits
                data_dict["ground_label"] = data_dict.get("tgt_object_id")
            elif head == "generation":   # query3d_unified.py:201-206
                label = data_dict["response"]
                logits = self.generation_head(query, data_dict["query_pad_masks"], label if self.training else None)
                data_dict["generation_logits"] = logits
                data_dict["generation_label"] = label
            elif head == "mask":
                if self.skip_query_encoder_mask_pred:
                    predictions_class, predictions_mask = [], []
                if getattr(self.unified_encoder, "_fused_final", None) is not None:
                    pred_logits, pred_masks = self.unified_encoder._fused_final  # computed inside the fused executor
                    # hand-over only: a module attribute would keep this forward's autograd graph (and with it the
                    # parameters' AccumulateGrad nodes of the stream it ran on) alive until the next forward
                    self.unified_encoder._fused_final = None
                else:
                    pred_logits, pred_masks, _ = mask_head_partial(query=query, skip_prediction=False)
                predictions_class.append(pred_logits)
                predictions_mask.append(pred_masks)
                data_dict["predictions_class"] = predictions_class
                data_dict["predictions_mask"] = predictions_mask
            else:
                raise NotImplementedError(f"Unknow head type: {head}")
        data_dict["query_embeds"] = query  # extra key (the reference does not expose the final query)
        return data_dict

    def unused_parameters(self):
        """Parameters that can never receive a gradient: the generation head's T5 ENCODER stack, bypassed through
        ``encoder_outputs`` (modules/heads/generation_head.py: the decoder cross-attends to the query tokens).  The shared
        token embedding is used by the decoder and stays.  TrainStep keeps these out of the flat optimizer (torch.optim.AdamW
        in the reference skips grad-None parameters)."""
        head = getattr(self, "generation_head", None)
        if head is None:
            return []
        shared = {id(p) for p in head.model.shared.parameters()}
        return [p for p in head.model.encoder.parameters() if id(p) not in shared]

    def get_opt_params(self):
        """model/query3d_unified.py:224-238."""
        def get_lr(c, default_lr):
            return default_lr if c is None or c.get("lr") is None else c.get("lr")

        groups = []
        for name, module in self._modules.items():
            lr = get_lr(self.cfg.model.get(name), self.cfg.solver.lr)
            groups += no_decay_param_group(module.named_parameters(), lr, name=name)
        n = sum(len(g["params"]) for g in groups)
        assert n == len(list(self.parameters())), "Some parameters are not optimized!"
        return groups


def make_cfg(*, d, H, L, memories, heads, d_in=None, spatial=True, structure="parallel", use_self_mask=False,
             num_blocks=1, dim_loc=3, C=201, foc=(), drop_test=(), offline_attn=False, skip_pred=False,
             activation="relu", ground_hidden=None, t5=None, memory_dropout=0.0) -> Cfg:
    """Config with the reference YAML layout (configs/instseg_sceneverse.yaml:92-155) for synthetic runs."""
    d_in = d_in or {m: d for m in memories}
    model = {"name": "Query3DUnified", "memories": list(memories), "heads": list(heads), "hidden_size": d,
             "use_offline_voxel_fts": True, "use_offline_attn_mask": offline_attn,
             "skip_query_encoder_mask_pred": skip_pred,
             "obj_loc": {"spatial_dim": 5, "dim_loc": dim_loc, "pairwise_rel_type": "center"},
             "unified_encoder": {"name": "QueryMaskEncoder", "args": {
                 "hidden_size": d, "num_attention_heads": H, "num_layers": L, "spatial_selfattn": spatial,
                 "memories": list(memories), "structure": structure, "use_self_mask": use_self_mask,
                 "num_blocks": num_blocks, "drop_memories_test": list(drop_test), "activation": activation,
                 "memory_dropout": memory_dropout}},
             "mask_head": {"name": "MaskHeadSegLevel", "args": {"hidden_size": d, "num_targets": C,
                                                                 "memories_for_match": list(memories),
                                                                 "filter_out_classes": list(foc)}},
             "ground_head": {"name": "GroundHead", "args": {"input_size": d, "hidden_size": ground_hidden or d // 2,
                                                             "dropout": 0.3}}}
    if "generation" in heads:   # unified_tasks_sceneverse.yaml:174-181 (t5: dict -> explicit HF config, offline)
        model["generation_head"] = {"name": "T5", "args": {"variant": "t5-small", "input_size": d,
                                                            "use_projection": True, **({"hf_config": t5} if t5 else {})},
                                    "lr": 1e-5}
    for m in memories:
        if m != "prompt":
            model[f"{m}_encoder"] = {"name": "ObjectEncoder", "args": {
                "input_feat_size": d_in[m], "hidden_size": d, "use_projection": True, "use_cls_head": False,
                "dropout": 0.1}}
    return Cfg({"model": model, "solver": {"lr": 1e-4}})
