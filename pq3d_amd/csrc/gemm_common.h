// Pieces shared by the GEMM kernel families (gemm.hip: 64x64 tiles with a k-tile pipeline; gemm_wk.hip: whole-K tiles
// for the small-M, latency-bound launches): raw 16-byte source chunks, the transposing LDS fragment read, vector
// load/store helpers and the fused epilogue applied to NV contiguous columns of one output row.
#pragma once
#include "common.h"

// ---- raw chunk of EPL source elements -------------------------------------------------------------------
template <typename TS, int EPL> struct Raw;
template <> struct Raw<float, 8> {
  float4 a, b;
  PQ_DEV void load(const float* p) { a = *(const float4*)p; b = *(const float4*)(p + 4); }
  PQ_DEV void to_float(float* v) const { v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w; }
};
template <> struct Raw<float, 4> {
  float4 a;
  PQ_DEV void load(const float* p) { a = *(const float4*)p; }
  PQ_DEV void to_float(float* v) const { v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; }
};
template <> struct Raw<bf16_t, 8> {
  u32x4 a;
  PQ_DEV void load(const bf16_t* p) { a = *(const u32x4*)p; }
  PQ_DEV void to_float(float* v) const {
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(a[j] << 16); v[2 * j + 1] = __uint_as_float(a[j] & 0xffff0000u); }
  }
};
template <> struct Raw<bf16_t, 4> {
  u32x2 a;
  PQ_DEV void load(const bf16_t* p) { a = *(const u32x2*)p; }
  PQ_DEV void to_float(float* v) const {
#pragma unroll
    for (int j = 0; j < 2; ++j) { v[2 * j] = __uint_as_float(a[j] << 16); v[2 * j + 1] = __uint_as_float(a[j] & 0xffff0000u); }
  }
};

// bf16 fragment (row `r0 + li` of the operand, k-slots 8*lg..8*lg+7 of k-step ks) from a [k][m]-oriented LDS tile via
// ds_read_b64_tr_b16 (lane (li,lg) pointing at row base + li/4, columns 4*(li%4).. receives tile[base + j][li]).
typedef short v4i16_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16_t lds_v4i16_t;
PQ_DEV u32x4 km_frag(const bf16_t* tile, int ldk, int r0, int ks, int li, int lg) {
  const bf16_t* p0 = tile + (ks * 32 + 8 * lg + (li >> 2)) * ldk + r0 + 4 * (li & 3);
  const v4i16_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_t*)p0);
  const v4i16_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_t*)(p0 + 4 * ldk));
  const u32x2 lo = __builtin_bit_cast(u32x2, a), hi = __builtin_bit_cast(u32x2, b);
  return (u32x4){lo.x, lo.y, hi.x, hi.y};
}

// ---- NV contiguous elements <-> floats (NV in {4, 8, 16}); vec_ok: aligned full segment ------------------------
template <int NV> PQ_DEV void load_vec(const void* p, int dt, long idx, bool vec_ok, int nvalid, float* v) {
  if (dt == PQ3D_F32) {
    const float* q = (const float*)p + idx;
    if (vec_ok) {
#pragma unroll
      for (int j = 0; j < NV; j += 4) { const float4 t = *(const float4*)(q + j); v[j] = t.x; v[j + 1] = t.y; v[j + 2] = t.z; v[j + 3] = t.w; }
    } else {
#pragma unroll
      for (int j = 0; j < NV; ++j) v[j] = j < nvalid ? q[j] : 0.f;
    }
  } else {
    const bf16_t* q = (const bf16_t*)p + idx;
    if (vec_ok) {
      if constexpr (NV >= 8) {
#pragma unroll
        for (int j = 0; j < NV; j += 8) {
          const u32x4 t = *(const u32x4*)(q + j);
#pragma unroll
          for (int k = 0; k < 4; ++k) { v[j + 2 * k] = __uint_as_float(t[k] << 16); v[j + 2 * k + 1] = __uint_as_float(t[k] & 0xffff0000u); }
        }
      } else {
        const u32x2 t = *(const u32x2*)q;
#pragma unroll
        for (int k = 0; k < 2; ++k) { v[2 * k] = __uint_as_float(t[k] << 16); v[2 * k + 1] = __uint_as_float(t[k] & 0xffff0000u); }
      }
    } else {
#pragma unroll
      for (int j = 0; j < NV; ++j) v[j] = j < nvalid ? bf2f(q[j]) : 0.f;
    }
  }
}
template <int NV> PQ_DEV void store_vec(void* p, int dt, long idx, bool vec_ok, int nvalid, const float* v) {
  if (dt == PQ3D_F32) {
    float* q = (float*)p + idx;
    if (vec_ok) {
#pragma unroll
      for (int j = 0; j < NV; j += 4) *(float4*)(q + j) = make_float4(v[j], v[j + 1], v[j + 2], v[j + 3]);
    } else {
#pragma unroll
      for (int j = 0; j < NV; ++j) if (j < nvalid) q[j] = v[j];
    }
  } else {
    bf16_t* q = (bf16_t*)p + idx;
    if (vec_ok) {
      if constexpr (NV >= 8) {
#pragma unroll
        for (int j = 0; j < NV; j += 8)
          *(u32x4*)(q + j) = (u32x4){pack_bf2(v[j], v[j + 1]), pack_bf2(v[j + 2], v[j + 3]), pack_bf2(v[j + 4], v[j + 5]), pack_bf2(v[j + 6], v[j + 7])};
      } else {
        *(u32x2*)q = (u32x2){pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
      }
    } else {
#pragma unroll
      for (int j = 0; j < NV; ++j) if (j < nvalid) q[j] = f2bf(v[j]);
    }
  }
}

// The per-group pointers of a block's (first) group.  load(d, g) reads them from the descriptor's arrays; the fast kernel
// requests them SPECULATIVELY for group blockIdx.z together with the header scalars (one scalar-cache round trip instead
// of two dependent ones: the true group index needs splitk / batch / kconcat from the header) and reloads only when the
// guess was wrong (split-K, batched or K-concatenated launches).
struct GPtrs {
  const void *A, *A2, *B, *B2, *bias, *aux;
  void *C, *C2;
  const uint8_t* row_mask;
  PQ_DEV void load(const pq3d_kdesc& d, int g) {
    const pq3d_kgroup& q = d.gp[g];
    A = q.A; A2 = q.A2; B = q.B; B2 = q.B2; bias = q.bias; aux = q.aux; C = q.C; C2 = q.C2; row_mask = q.row_mask;
  }
};

// Fused epilogue of NV contiguous columns [col, col + NV) of output row `row` (row < M, col < N) of group g, batch entry z:
// v holds alpha * accumulator (+ bias if bias_done).  Order: bias, C2 (pre-activation), activation, dropout,
// activation-gradient / "+ aux", row mask / scale, row fill, C store, mask bytes.
template <int NV>
PQ_DEV void epi_row(const pq3d_kdesc& d, const GPtrs& gp, float (&v)[NV], int g, int z, int row, int col, bool bias_done) {
  constexpr int AL = NV >= 8 ? 8 : 4;   // element alignment that makes every participant's segment a whole vector access
  const int nvalid = min(NV, d.N - col);
  void* C = gp.C;
  const long ci = (long)z * d.strideC + (long)row * d.ldc + col;
  // vector path: full segment, aligned for the widest participant
  const uintptr_t pbits = (uintptr_t)C | (uintptr_t)gp.C2 | (uintptr_t)gp.aux;
  const bool vec_ok = nvalid == NV && (d.ldc % AL == 0) && (d.strideC % AL == 0) && (col % AL == 0) && (pbits & 15) == 0;
  if (gp.bias && !bias_done) {
    float bv[NV];
    load_vec<NV>(gp.bias, d.dtBias, col, nvalid == NV && (col % AL == 0) && (((uintptr_t)gp.bias) & 15) == 0, nvalid, bv);
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] += bv[j];
  }
  if (gp.C2) store_vec<NV>(gp.C2, d.dtC2, ci, vec_ok, nvalid, v);
  if (d.act == PQ3D_ACT_RELU) {
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = fmaxf(v[j], 0.f);
  } else if (d.act == PQ3D_ACT_GELU) {
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = gelu_f(v[j]);
  }
  if (drop_on(d.drop)) {   // dropout of the activated output; col is even, so pairs never straddle threads
    const DropState ds = drop_init(d.drop, g, d.N);
    const uint32_t drow = (uint32_t)((long)z * d.M + row);
#pragma unroll
    for (int j = 0; j < NV; j += 2) {
      const uint32_t w = drop_word(ds, drow, (uint32_t)(col + j) >> 1);
      v[j] = drop_keep_lo(ds, w) ? v[j] * ds.scale : 0.f;
      v[j + 1] = drop_keep_hi(ds, w) ? v[j + 1] * ds.scale : 0.f;
    }
  }
  if (d.act_grad && (gp.aux || d.act_grad != PQ3D_ACT_ADD)) {   // "+ aux" without an aux pointer in this group: no addend
    float av[NV];
    load_vec<NV>(gp.aux, d.dtAux, ci, vec_ok, nvalid, av);
    if (d.act_grad == PQ3D_ACT_RELU) {
#pragma unroll
      for (int j = 0; j < NV; ++j) v[j] = av[j] > 0.f ? v[j] : 0.f;
    } else if (d.act_grad == PQ3D_ACT_GELU) {
#pragma unroll
      for (int j = 0; j < NV; ++j) v[j] *= gelu_grad_f(av[j]);
    } else {
#pragma unroll
      for (int j = 0; j < NV; ++j) v[j] += av[j];
    }
  }
  const long ri = (long)z * d.M + row;
  float rsc = 1.f;
  if (gp.row_mask) rsc = gp.row_mask[ri] ? 1.f : 0.f;
  if (d.row_scale) rsc *= d.row_scale[ri];
  if (gp.row_mask || d.row_scale) {
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] *= rsc;
  }
  if (d.row_fill_flag && d.row_fill_flag[ri]) {
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = d.row_fill;
  }
  store_vec<NV>(C, d.dtC, ci, vec_ok, nvalid, v);
  if (d.mask_out) {
#pragma unroll
    for (int j = 0; j < NV; ++j)
      if (j < nvalid) d.mask_out[((long)z * d.N + col + j) * d.M + row] = (1.f / (1.f + __expf(-v[j])) < 0.5f) ? 1 : 0;
  }
}
