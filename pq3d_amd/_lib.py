"""ctypes binding of libpq3d_hip.so (include/pq3d_hip.h).  No CPU fallback: if the library is missing or
a call fails, a Pq3dError is raised -- the product path never routes through torch math or the oracle."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

F32, BF16, BF16X3 = 0, 1, 2
ACT = {None: 0, "none": 0, "relu": 1, "gelu": 2, "add": 3, "planes": 4}
MAXG = 32
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PQ3D_LIB_PATH") or os.path.join(_HERE, "libpq3d_hip.so")   # override: A/B builds of the kernels


class Pq3dError(RuntimeError):
    pass


class Dropout(C.Structure):
    """pq3d_dropout: p, site id, device pointer to the 64-bit seed word."""
    _fields_ = [("p", C.c_float), ("site", C.c_uint32), ("seed", C.c_void_p)]


class Drop:
    """Host-side handle of one dropout site: probability, site id and the device seed tensor (int64[1])."""
    __slots__ = ("p", "site", "seed")

    def __init__(self, p: float, site: int, seed: torch.Tensor):
        self.p, self.site, self.seed = float(p), int(site), seed

    def at(self, site_add: int) -> "Drop":
        return Drop(self.p, self.site + site_add, self.seed)

    def c(self) -> Dropout:
        d = Dropout()
        d.p, d.site, d.seed = self.p, self.site & 0xFFFFFFFF, ptr(self.seed)
        return d


def set_drop(field: Dropout, drop: Optional["Drop"]) -> None:
    if drop is not None and drop.p > 0.0:
        field.p, field.site, field.seed = drop.p, drop.site & 0xFFFFFFFF, ptr(drop.seed)


MAX_OPT_SEGMENTS = 16


class OptSegments(C.Structure):
    _fields_ = [("n", C.c_int32), ("end", C.c_int64 * MAX_OPT_SEGMENTS), ("lr_mul", C.c_float * MAX_OPT_SEGMENTS),
                ("weight_decay", C.c_float * MAX_OPT_SEGMENTS)]


class AdamWHp(C.Structure):
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("max_grad_norm", C.c_float), ("sched", C.c_int32),
                ("warmup_steps", C.c_int32), ("total_steps", C.c_int32), ("sched_gamma", C.c_float),
                ("sched_stride", C.c_int32)]


SCHED = {"constant": 0, "warmup_cosine": 1, "warmup_exp": 2}
SUMSQ_PARTIALS = 1024


class MaskPrepDesc(C.Structure):
    _fields_ = [("layers", C.c_int32), ("B", C.c_int32), ("Ns", C.c_int32), ("Nq", C.c_int32), ("nsplit", C.c_int32),
                ("X", C.c_void_p * MAXG), ("seg_len", C.c_void_p), ("sig", C.c_void_p), ("sp_part", C.c_void_p),
                ("sg_part", C.c_void_p)]


class MatchCostDesc(C.Structure):
    _fields_ = [("layers", C.c_int32), ("B", C.c_int32), ("Nq", C.c_int32), ("Nt", C.c_int32), ("Ns", C.c_int32),
                ("C", C.c_int32), ("nsplit", C.c_int32), ("w_class", C.c_float), ("w_mask", C.c_float),
                ("w_dice", C.c_float), ("ignore_label", C.c_int64),
                ("TXS", C.c_void_p), ("sp_part", C.c_void_p), ("sg_part", C.c_void_p), ("t_sum", C.c_void_p),
                ("seg_len", C.c_void_p), ("n_inst", C.c_void_p), ("cls_logits", C.c_void_p * MAXG), ("labels", C.c_void_p),
                ("cost", C.c_void_p)]


class MaskGradDesc(C.Structure):
    _fields_ = [("layers", C.c_int32), ("B", C.c_int32), ("Ns", C.c_int32), ("Nq", C.c_int32), ("Nt", C.c_int32),
                ("Nm", C.c_int32),
                ("sig", C.c_void_p), ("T", C.c_void_p), ("TXS", C.c_void_p), ("sig_sum", C.c_void_p), ("t_sum", C.c_void_p),
                ("seg_len", C.c_void_p), ("q_idx", C.c_void_p), ("t_idx", C.c_void_p), ("n_match", C.c_void_p),
                ("g", C.c_void_p), ("dX", C.c_void_p * MAXG)]


TT_MAX_PROBLEMS = 56


class TtProblem(C.Structure):
    _fields_ = [("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("dtA", C.c_int32), ("dtB", C.c_int32),
                ("lda", C.c_int64), ("ldb", C.c_int64), ("A", C.c_void_p), ("B", C.c_void_p), ("B2", C.c_void_p),
                ("C", C.c_void_p), ("colsum", C.c_void_p)]


class CeDesc(C.Structure):
    _fields_ = [("layers", C.c_int32), ("C", C.c_int32), ("R", C.c_int64), ("ignore_index", C.c_int64),
                ("logits", C.c_void_p * MAXG), ("target", C.c_void_p), ("row_loss", C.c_void_p), ("lse", C.c_void_p),
                ("scale", C.c_void_p), ("dlogits", C.c_void_p * MAXG), ("scale_mul", C.c_void_p)]


class GemmDesc(C.Structure):
    _fields_ = [
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("groups", C.c_int32), ("batch", C.c_int32),
        ("ct", C.c_int32),
        ("dtA", C.c_int32), ("dtA2", C.c_int32), ("dtB", C.c_int32), ("dtC", C.c_int32), ("dtC2", C.c_int32),
        ("dtAux", C.c_int32), ("dtBias", C.c_int32),
        ("transA", C.c_int32), ("transB", C.c_int32), ("act", C.c_int32), ("act_grad", C.c_int32),
        ("splitk", C.c_int32), ("kconcat", C.c_int32), ("accumulate", C.c_int32), ("dtB2", C.c_int32),
        ("alpha", C.c_float), ("row_fill", C.c_float),
        ("lda", C.c_int64), ("ldb", C.c_int64), ("ldc", C.c_int64),
        ("strideA", C.c_int64), ("strideB", C.c_int64), ("strideC", C.c_int64),
        ("A", C.c_void_p * MAXG), ("A2", C.c_void_p * MAXG), ("B", C.c_void_p * MAXG), ("B2", C.c_void_p * MAXG),
        ("bias", C.c_void_p * MAXG), ("C", C.c_void_p * MAXG), ("C2", C.c_void_p * MAXG),
        ("aux", C.c_void_p * MAXG), ("row_mask", C.c_void_p * MAXG), ("colsum", C.c_void_p * MAXG),
        ("row_scale", C.c_void_p), ("row_fill_flag", C.c_void_p), ("mask_out", C.c_void_p),
        ("drop", Dropout),
    ]


class AttnProj(C.Structure):
    _fields_ = [("mode", C.c_int32), ("dm", C.c_int32), ("x", C.c_void_p), ("x2", C.c_void_p), ("w", C.c_void_p * 3),
                ("b", C.c_void_p * 3)]


class AttnDesc(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("H", C.c_int32), ("Lq", C.c_int32), ("Lk", C.c_int32), ("dh", C.c_int32),
        ("ct", C.c_int32), ("dt", C.c_int32), ("zero_attn", C.c_int32), ("mask_bmod", C.c_int32), ("scale", C.c_float),
        ("q_sb", C.c_int64), ("q_sl", C.c_int64), ("q_sh", C.c_int64),
        ("k_sb", C.c_int64), ("k_sl", C.c_int64), ("k_sh", C.c_int64),
        ("v_sb", C.c_int64), ("v_sl", C.c_int64), ("v_sh", C.c_int64),
        ("o_sb", C.c_int64), ("o_sl", C.c_int64), ("o_sh", C.c_int64),
        ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("o", C.c_void_p), ("lse", C.c_void_p),
        ("kpm", C.c_void_p), ("mask", C.c_void_p), ("row_open", C.c_void_p), ("bias", C.c_void_p),
        ("dout", C.c_void_p), ("dq", C.c_void_p), ("dk", C.c_void_p), ("dv", C.c_void_p),
        ("delta", C.c_void_p), ("dbias", C.c_void_p), ("ksplit", C.c_int32), ("ws", C.c_void_p),
        ("drop", Dropout), ("drop_bmod", C.c_int32), ("proj", AttnProj), ("mask_bits", C.c_void_p),
        ("k_lo", C.c_void_p), ("v_lo", C.c_void_p), ("q_bf", C.c_void_p), ("o_bf", C.c_void_p),
    ]


class LnDesc(C.Structure):
    _fields_ = [
        ("R", C.c_int32), ("d", C.c_int32), ("M", C.c_int32), ("rows_per_scene", C.c_int32),
        ("dt_x", C.c_int32), ("dt_o", C.c_int32), ("dt_y", C.c_int32), ("eps", C.c_float),
        ("x", C.c_void_p), ("o", C.c_void_p * MAXG), ("gamma", C.c_void_p * MAXG), ("beta", C.c_void_p * MAXG),
        ("coef", C.c_void_p), ("y", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p),
        ("dy", C.c_void_p), ("dx", C.c_void_p), ("d_o", C.c_void_p * MAXG), ("dgamma", C.c_void_p * MAXG),
        ("dbeta", C.c_void_p * MAXG), ("accumulate", C.c_int32), ("independent", C.c_int32),
        ("ys", C.c_void_p * MAXG), ("dys", C.c_void_p * MAXG),
        ("sum_branches", C.c_int32), ("osum", C.c_void_p), ("drop", Dropout), ("dx_zeroed", C.c_int32),
        ("dy2", C.c_void_p), ("dy3", C.c_void_p),
    ]


class ChainFfnDesc(C.Structure):
    _fields_ = [("R", C.c_int32), ("d", C.c_int32), ("F", C.c_int32), ("eps1", C.c_float), ("eps2", C.c_float)] + \
               [(n, C.c_void_p) for n in ("o_s", "Wo", "bo", "x1s", "g1", "be1", "f", "x2", "mean1", "rstd1", "W1", "b1", "h", "W2",
                                          "b2", "zp", "z", "g2", "be2", "x3", "mean2", "rstd2", "flags", "err")] + \
               [("nq", C.c_int32), ("qpos", C.c_void_p), ("Wq", C.c_void_p * 3), ("bq", C.c_void_p * 3), ("qout", C.c_void_p * 3),
                ("qout_f32", C.c_int32)] + [(n, C.c_void_p) for n in ("sa_q", "sa_k", "sa_v", "sa_bias", "sa_kpm", "sa_lse")] + \
               [("sa_nq", C.c_int32), ("sa_scale", C.c_float)]


class ChainCaDesc(C.Structure):
    _fields_ = [("R", C.c_int32), ("d", C.c_int32), ("M", C.c_int32), ("rows_per_scene", C.c_int32), ("eps", C.c_float),
                ("o", C.c_void_p * 3), ("Wo", C.c_void_p * 3), ("bo", C.c_void_p * 3), ("x", C.c_void_p), ("gamma", C.c_void_p * 3),
                ("beta", C.c_void_p * 3), ("coef", C.c_void_p), ("op", C.c_void_p * 3), ("x1", C.c_void_p), ("mean", C.c_void_p),
                ("rstd", C.c_void_p), ("qpos", C.c_void_p), ("Wqkv", C.c_void_p * 3), ("bqkv", C.c_void_p * 3), ("qkv", C.c_void_p * 3),
                ("flags", C.c_void_p), ("err", C.c_void_p), ("o_f32", C.c_int32)]


class ChainMhDesc(C.Structure):
    _fields_ = [("R", C.c_int32), ("d", C.c_int32), ("Mm", C.c_int32), ("C", C.c_int32), ("eps", C.c_float), ("fill", C.c_float)] + \
               [(n, C.c_void_p) for n in ("x", "W0", "b0", "gamma", "beta", "W4", "b4", "colfill")] + \
               [("Wq", C.c_void_p * 3), ("bq", C.c_void_p * 3)] + \
               [(n, C.c_void_p) for n in ("h1", "h2", "mean", "rstd", "cls")] + \
               [("qm", C.c_void_p * 3), ("flags", C.c_void_p), ("err", C.c_void_p)]


class ChainMhBwdDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("R", "d", "Mm", "C", "dq_f32")] + \
               [(n, C.c_void_p) for n in ("dc", "colfill", "dcl", "W4", "h1", "mean", "rstd", "gamma", "dgamma", "dbeta", "dh2", "dpre",
                                          "W0", "cur")] + \
               [("dq", C.c_void_p * 3), ("Wq", C.c_void_p * 3), ("out", C.c_void_p), ("flags", C.c_void_p), ("err", C.c_void_p),
                ("lnws", C.c_void_p), ("nq", C.c_int32), ("dqc", C.c_void_p * 3), ("Wqc", C.c_void_p * 3), ("dxr", C.c_void_p),
                ("gq", C.c_void_p)]


class ChainFfnBwdDesc(C.Structure):
    _fields_ = [("R", C.c_int32), ("d", C.c_int32), ("F", C.c_int32)] + \
               [(n, C.c_void_p) for n in ("dx", "x2", "z", "g2", "mean2", "rstd2", "dg2", "db2", "dy", "W2", "h", "dhp", "W1", "part", "x1s",
                                          "f", "g1", "mean1", "rstd1", "dg1", "db1", "df", "flags", "err", "lnws")] + \
               [("nq", C.c_int32), ("dq", C.c_void_p * 3), ("Wq", C.c_void_p * 3), ("dxr", C.c_void_p), ("gq", C.c_void_p), ("dxo", C.c_void_p)]


class ChainSaBwdDesc(C.Structure):
    _fields_ = [("R", C.c_int32), ("d", C.c_int32), ("M", C.c_int32), ("rows_per_scene", C.c_int32),
                ("dqkv", C.c_void_p * 3), ("Wl", C.c_void_p * 3), ("aux2", C.c_void_p), ("g3", C.c_void_p * 3), ("x", C.c_void_p),
                ("op", C.c_void_p * 3), ("gamma", C.c_void_p * 3), ("mean", C.c_void_p), ("rstd", C.c_void_p), ("coef", C.c_void_p),
                ("dop", C.c_void_p * 3), ("dxr", C.c_void_p), ("dgamma", C.c_void_p * 3), ("dbeta", C.c_void_p * 3),
                ("Wo", C.c_void_p * 3), ("do_all", C.c_void_p * 3), ("flags", C.c_void_p), ("err", C.c_void_p), ("lnws", C.c_void_p)]


_lib = None

_SIGS = {
    "pq3d_gemm": [C.POINTER(GemmDesc), C.c_void_p],
    "pq3d_gemm_set_wk": [C.c_int, C.c_int],
    "pq3d_attn_fwd": [C.POINTER(AttnDesc), C.c_void_p],
    "pq3d_attn_bwd": [C.POINTER(AttnDesc), C.c_void_p],
    "pq3d_gemm_tt_multi": [C.POINTER(TtProblem), C.c_int32, C.c_void_p],
    "pq3d_gemm_tt_multi_wide": [C.c_int32],
    "pq3d_chain_ffn_fwd": [C.POINTER(ChainFfnDesc), C.c_void_p],
    "pq3d_chain_ca_fwd": [C.POINTER(ChainCaDesc), C.c_void_p],
    "pq3d_chain_mh_fwd": [C.POINTER(ChainMhDesc), C.c_void_p],
    "pq3d_chain_mh_bwd": [C.POINTER(ChainMhBwdDesc), C.c_void_p],
    "pq3d_chain_ffn_bwd": [C.POINTER(ChainFfnBwdDesc), C.c_void_p],
    "pq3d_chain_sa_bwd": [C.POINTER(ChainSaBwdDesc), C.c_void_p],
    "pq3d_mask_pack": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p],
    "pq3d_mask_row_all": [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p],
    "pq3d_attn_resident": [C.c_int],
    "pq3d_mask_not": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p],
    "pq3d_zero_many": [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p],
    "pq3d_copy_many": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p],
    "pq3d_sum_n": [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p],
    "pq3d_sum_pair": [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p],
    "pq3d_mean_all": [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p],
    "pq3d_fill_scaled": [C.c_void_p, C.c_int64, C.c_void_p, C.c_float, C.c_void_p],
    "pq3d_mean_many": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p],
    "pq3d_mean_many_bwd": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p],
    "pq3d_cast_transpose": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p],
    "pq3d_add_ln_fwd": [C.POINTER(LnDesc), C.c_void_p],
    "pq3d_add_ln_bwd": [C.POINTER(LnDesc), C.c_void_p],
    "pq3d_colsum": [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p],
    "pq3d_colsum_grouped": [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_int64, C.c_int64,
                            C.c_int64, C.c_int32, C.c_void_p],
    "pq3d_scale_rows": [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                        C.c_void_p, C.c_void_p],
    "pq3d_scale_rows_grouped": [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p],
    "pq3d_add_cast": [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32, C.c_int32,
                      C.c_int64, C.c_void_p],
    "pq3d_chain_device_ok": [C.c_int32, C.c_void_p],
    "pq3d_test_occupy_cus": [C.c_int32, C.c_int32, C.c_int64, C.c_void_p],
    "pq3d_split_planes": [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                          C.POINTER(C.c_int64), C.c_int32, C.c_void_p],
    "pq3d_bias_add_rows": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p],
    "pq3d_act_bwd": [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int64,
                     C.c_void_p],
    "pq3d_fill_cols": [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int32, C.c_float, C.c_void_p],
    "pq3d_mask_inv_den": [C.POINTER(C.c_void_p), C.c_int32, C.c_int64, C.c_void_p, C.c_void_p],
    "pq3d_pairwise_locs": [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p],
    "pq3d_fourier": [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                     C.c_int32, C.c_void_p],
    "pq3d_fourier_pair": [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                          C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p],
    "pq3d_spatial_bias_fwd": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                              C.c_void_p],
    "pq3d_spatial_bias_bwd": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                              C.c_int32, C.c_int32, C.c_void_p],
    "pq3d_spatial_bias_bwd_acc": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                  C.c_int32, C.c_int32, C.c_void_p],
    "pq3d_spatial_bias_fwd_grouped": [C.c_void_p] + [C.POINTER(C.c_void_p)] * 3 + [C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                                                                   C.c_void_p],
    "pq3d_spatial_bias_bwd_grouped": [C.c_void_p] + [C.POINTER(C.c_void_p)] * 5 + [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p],
    "pq3d_gate_mix_fwd": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p],
    "pq3d_gate_mix_bwd": [C.c_void_p] * 7 + [C.c_int64, C.c_void_p],
    "pq3d_segment_plan_bytes": [C.c_int64, C.c_int64],
    "pq3d_segment_ws_bytes": [C.c_int64, C.c_int64, C.c_int64],
    "pq3d_segment_plan": [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p],
    "pq3d_segment_reduce": [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                            C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p],
    "pq3d_segment_gather": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p],
    "pq3d_sumsq_partials": [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p],
    "pq3d_train_scalars": [C.POINTER(AdamWHp), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "pq3d_adamw": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(AdamWHp), C.POINTER(OptSegments),
                   C.c_void_p, C.c_void_p],
    "pq3d_mask_cost_prep": [C.POINTER(MaskPrepDesc), C.c_void_p],
    "pq3d_mask_cost_nsplit": [C.c_int32],
    "pq3d_match_cost": [C.POINTER(MatchCostDesc), C.c_void_p],
    "pq3d_matched_mask_grad": [C.POINTER(MaskGradDesc), C.c_void_p],
    "pq3d_cross_entropy_fwd": [C.POINTER(CeDesc), C.c_void_p],
    "pq3d_cross_entropy_bwd": [C.POINTER(CeDesc), C.c_void_p],
    "pq3d_cross_entropy_mean": [C.POINTER(CeDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "pq3d_padded_mask_sums": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p],
    "pq3d_padded_mask_grad": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                              C.c_int32, C.c_int32, C.c_void_p],
    "pq3d_pad_sequence": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int32,
                          C.c_void_p, C.c_void_p],
    "pq3d_pad_sequence_2d": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64,
                             C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p],
    "pq3d_furthest_point_sampling": [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p],
    "pq3d_ball_query": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p],
    "pq3d_gather_points": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_void_p],
    "pq3d_gather_points_grad": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_void_p],
    "pq3d_three_nn": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p],
    "pq3d_three_interpolate": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                               C.c_void_p],
    "pq3d_three_interpolate_grad": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_void_p],
    "pq3d_group_rows": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32,
                        C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p],
    "pq3d_group_maxpool": [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_void_p],
    "pq3d_rmsnorm_fwd": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_void_p],
    "pq3d_rmsnorm_bwd_res": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                             C.c_int32, C.c_int32, C.c_void_p],
    "pq3d_rmsnorm_bwd": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                         C.c_int32, C.c_void_p],
    "pq3d_embedding_fwd": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p],
    "pq3d_embedding_bwd_acc": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p],
    "pq3d_dropout_mask": [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(Dropout), C.c_void_p],
    "pq3d_dropout_apply": [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.POINTER(Dropout),
                           C.c_void_p],
    "pq3d_dropout_apply_scaled": [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.POINTER(Dropout),
                                  C.c_float, C.c_void_p],
    "pq3d_rmsnorm_bwd_res_drop": [C.c_void_p] * 7 + [C.c_int64, C.c_int32, C.c_int32, C.POINTER(Dropout), C.c_void_p, C.c_void_p],
    "pq3d_embedding_drop_fwd": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.POINTER(Dropout), C.c_void_p],
    "pq3d_embedding_drop_bwd_acc": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.POINTER(Dropout), C.c_void_p],
    "pq3d_t5_prep": [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                     C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_void_p],
    "pq3d_t5_bias_bwd": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p],
    # data-parallel gradient exchange over RCCL (csrc/comm.hip)
    "pq3d_comm_unique_id": [C.c_void_p],
    "pq3d_comm_init": [C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)],
    "pq3d_comm_destroy": [C.c_void_p],
    "pq3d_comm_info": [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)],
    "pq3d_allreduce_grads": [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p],
    "pq3d_allreduce_wire_scratch_bytes": [C.c_int32, C.c_int64],
    "pq3d_allreduce_grads_wire": [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p],
    "pq3d_test_wire_reduce": [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p],
}
_RET64 = {"pq3d_segment_plan_bytes", "pq3d_segment_ws_bytes", "pq3d_allreduce_wire_scratch_bytes"}   # size queries: bytes (or -1), not a status code
EXPORTS = sorted(list(_SIGS) + ["pq3d_last_error", "pq3d_version"])


def lib() -> C.CDLL:
    """Load (once) the in-tree shared library; raise loudly if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Pq3dError(f"{LIB_PATH} not found: build it with `python -m pq3d_amd.build` "
                            "(there is no CPU fallback for the pq3d_amd product path)")
        L = C.CDLL(LIB_PATH)
        for name, args in _SIGS.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = C.c_int64 if name in _RET64 else C.c_int
        L.pq3d_last_error.restype = C.c_char_p
        L.pq3d_version.restype = C.c_int
        _lib = L
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().pq3d_last_error().decode(errors="replace")
        raise Pq3dError(f"{what} failed (rc={rc}): {msg}")


def stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dt_of(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise Pq3dError(f"unsupported dtype {t.dtype}")


def tdtype(dt: int) -> torch.dtype:
    return torch.float32 if dt == F32 else torch.bfloat16


def ptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise Pq3dError("pq3d_amd kernels need device tensors (no CPU fallback)")
    return t.data_ptr()


def _fill(arr, tensors: Optional[Sequence[Optional[torch.Tensor]]]):
    if tensors is None:
        return
    for i, t in enumerate(tensors):
        arr[i] = ptr(t)


def gemm(*, M, N, K, A, B, Cs, ct, lda, ldb, ldc, A2=None, B2=None, bias=None, C2=None, aux=None, row_mask=None,
         transA=False, transB=False, batch=1, strideA=0, strideB=0, strideC=0, act=None, act_grad=None, splitk=1,
         kconcat=0, accumulate=False, alpha=1.0, row_scale=None, row_fill_flag=None, row_fill=0.0,
         mask_out=None, colsum=None, drop=None) -> None:
    """kconcat: number of consecutive groups concatenated along K per output (True = all groups)."""
    if kconcat is True:
        kconcat = len(A)
    kconcat = int(kconcat or 0)
    d = GemmDesc()
    d.M, d.N, d.K, d.groups, d.batch, d.ct = M, N, K, len(A), batch, ct
    d.dtA, d.dtB, d.dtC = dt_of(A[0]), dt_of(B[0]), dt_of(Cs[0])
    first = lambda ts: next((t for t in (ts or []) if t is not None), None)
    d.dtA2 = dt_of(first(A2)) if first(A2) is not None else 0
    d.dtB2 = dt_of(first(B2)) if first(B2) is not None else 0
    d.dtC2 = dt_of(first(C2)) if first(C2) is not None else 0
    d.dtAux = dt_of(first(aux)) if first(aux) is not None else 0
    d.dtBias = dt_of(first(bias)) if first(bias) is not None else 0
    d.transA, d.transB = int(transA), int(transB)
    d.act, d.act_grad, d.splitk, d.kconcat, d.accumulate = ACT[act], ACT[act_grad], splitk, kconcat, int(accumulate)
    d.alpha, d.row_fill = alpha, row_fill
    d.lda, d.ldb, d.ldc = lda, ldb, ldc
    d.strideA, d.strideB, d.strideC = strideA, strideB, strideC
    _fill(d.A, A); _fill(d.A2, A2); _fill(d.B, B); _fill(d.B2, B2); _fill(d.bias, bias); _fill(d.C, Cs)
    _fill(d.C2, C2); _fill(d.aux, aux); _fill(d.row_mask, row_mask); _fill(d.colsum, colsum)
    d.row_scale, d.row_fill_flag, d.mask_out = ptr(row_scale), ptr(row_fill_flag), ptr(mask_out)
    set_drop(d.drop, drop)
    from .profiler import timed
    nb = (M * K * (2 if d.dtA else 4) + N * K * (2 if d.dtB else 4)) * len(A) * batch + \
        M * N * (2 if d.dtC else 4) * (len(A) // max(kconcat, 1)) * batch
    key = f"M{M}N{N}K{K}g{len(A)}b{batch}{'T' if transA else 'N'}{'T' if transB else 'N'}" \
          f"{'k%d' % kconcat if kconcat else ''}{'s%d' % splitk if splitk > 1 else ''}ct{ct}"
    check(timed("pq3d_gemm", key, 2.0 * M * N * K * len(A) * batch, nb, lib().pq3d_gemm, C.byref(d), stream()),
          "pq3d_gemm")
