// Shared device helpers for the pq3d gfx950 kernels (CDNA4 only: wave64, MFMA 16x16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pq3d_hip.h"

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define PQ_DEV __device__ __forceinline__

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

PQ_DEV float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
// fp32 -> bf16, round-to-nearest-even with NaN handling in hardware: native __bf16 conversions lower to ONE
// v_cvt_pk_bf16_f32 per pair on gfx950 (a hand-rolled bit trick with a NaN branch costs a divergent exec-mask
// sequence per element and dominated the GEMM staging loop).
PQ_DEV bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
PQ_DEV unsigned pack_bf2(float lo, float hi) {
  const f32x2 f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
}

template <typename T> struct Cvt;
template <> struct Cvt<float> {
  static PQ_DEV float from(float f) { return f; }
  static PQ_DEV float to(float v) { return v; }
};
template <> struct Cvt<bf16_t> {
  static PQ_DEV bf16_t from(float f) { return f2bf(f); }
  static PQ_DEV float to(bf16_t v) { return bf2f(v); }
};

// ---------------------------------------------------------------------------------------------
// MFMA abstraction.  One "step" consumes a 16-byte fragment per lane for A and for B:
//   lane l: i = l & 15 (row of A / column of B), g = l >> 4 (k-group);
//   the fragment holds EPL consecutive k elements starting at k = step*KSTEP + g*EPL.
//   bf16: EPL = 8, KSTEP = 32 (one v_mfma_f32_16x16x32_bf16)
//   f32 : EPL = 4, KSTEP = 16 (four v_mfma_f32_16x16x4_f32; MFMA j pairs element j of A and B, so the
//         k order inside the step is a permutation applied identically to A and B -> same dot product).
// C/D layout (both): lane holds C[row = 4*g + r][col = i], r = 0..3.
// ---------------------------------------------------------------------------------------------
template <typename CT> struct Mma;
template <> struct Mma<bf16_t> {
  static constexpr int EPL = 8, KSTEP = 32;
  typedef u32x4 Frag;
  static PQ_DEV void mma(f32x4& c, const Frag& a, const Frag& b) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static constexpr int EPL = 4, KSTEP = 16;
  typedef u32x4 Frag;
  static PQ_DEV void mma(f32x4& c, const Frag& a, const Frag& b) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[j]), __uint_as_float(b[j]), c, 0, 0, 0);
  }
};

// Load `n` (<= 8) consecutive elements of dtype `dt` (PQ3D_F32 / PQ3D_BF16) starting at element index
// `idx` of `base` into out[0..n) as floats; elements at position >= valid are returned as 0.
// Fast 16-byte paths when fully valid and aligned.
template <int N>
PQ_DEV void load_elems(const void* __restrict__ base, int dt, long idx, int valid, float (&out)[N]) {
  if (dt == PQ3D_F32) {
    const float* p = (const float*)base + idx;
    if (valid >= N && ((((uintptr_t)p) & 15) == 0)) {
#pragma unroll
      for (int j = 0; j < N; j += 4) {
        float4 v = *(const float4*)(p + j);
        out[j] = v.x; out[j + 1] = v.y; out[j + 2] = v.z; out[j + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < N; ++j) out[j] = (j < valid) ? p[j] : 0.f;
    }
  } else {
    const bf16_t* p = (const bf16_t*)base + idx;
    if (valid >= N && ((((uintptr_t)p) & (2 * N - 1)) == 0)) {
      if (N == 8) {
        u32x4 v = *(const u32x4*)p;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          out[2 * j] = __uint_as_float(v[j] << 16);
          out[2 * j + 1] = __uint_as_float(v[j] & 0xffff0000u);
        }
      } else {
        u32x2 v = *(const u32x2*)p;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          out[2 * j] = __uint_as_float(v[j] << 16);
          out[2 * j + 1] = __uint_as_float(v[j] & 0xffff0000u);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < N; ++j) out[j] = (j < valid) ? bf2f(p[j]) : 0.f;
    }
  }
}

// Pack EPL floats into one 16-byte fragment of compute type CT.
template <typename CT> PQ_DEV u32x4 pack_frag(const float* v);
template <> PQ_DEV u32x4 pack_frag<bf16_t>(const float* v) {
  u32x4 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) r[j] = pack_bf2(v[2 * j], v[2 * j + 1]);
  return r;
}
template <> PQ_DEV u32x4 pack_frag<float>(const float* v) {
  u32x4 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) r[j] = __float_as_uint(v[j]);
  return r;
}

PQ_DEV void store_elem(void* base, int dt, long idx, float v) {
  if (dt == PQ3D_F32) ((float*)base)[idx] = v;
  else ((bf16_t*)base)[idx] = f2bf(v);
}
PQ_DEV float load_elem(const void* base, int dt, long idx) {
  return dt == PQ3D_F32 ? ((const float*)base)[idx] : bf2f(((const bf16_t*)base)[idx]);
}

PQ_DEV float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
PQ_DEV float gelu_grad_f(float x) {
  return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// Kernel-argument prefetch (see gemm_fast_kernel in gemm.hip): one empty asm statement that "uses" every descriptor
// scalar pins them at the top of the kernel -- one batch of scalar loads instead of one scalar-cache miss per first use.
// ---------------------------------------------------------------------------------------------
// Kernel-side form of pq3d_gemm_desc.  The public descriptor keeps one 256-byte ARRAY per kind of pointer (A[32],
// A2[32], ...): the pointers of one group sit in 10 different cache lines, and with the header and the trailing scalars a
// GEMM workgroup touched 13 lines of cold kernel-argument memory before its first operand load -- ~200 cycles each,
// 2.6k cycles measured (tools/probes/gemm_x3_timeline.py).  The kernels take this struct instead: every scalar in the
// first 3 lines, then one 80-byte record per group (2 lines).  Same scalar member names as the public struct, so kernel
// code reads `d.M`, `d.alpha` unchanged and `d.gp[g].A` for the pointers.  Built on the host by make_kdesc().
// ---------------------------------------------------------------------------------------------
struct pq3d_kgroup {
  const void *A, *A2, *B, *B2, *bias, *aux;
  void *C, *C2;
  const uint8_t* row_mask;
  float* colsum;
};
struct pq3d_kdesc {
  int32_t M, N, K, groups, batch, ct, dtA, dtA2, dtB, dtC, dtC2, dtAux, dtBias, transA, transB, act, act_grad, splitk, kconcat,
      accumulate, dtB2, xcd_order;
  float alpha, row_fill;
  int64_t lda, ldb, ldc, strideA, strideB, strideC;
  const float* row_scale;
  const uint8_t* row_fill_flag;
  uint8_t* mask_out;
  pq3d_dropout drop;
  pq3d_kgroup gp[PQ3D_MAX_GROUPS];
};
inline pq3d_kdesc make_kdesc(const pq3d_gemm_desc& d) {
  pq3d_kdesc k;
  k.M = d.M; k.N = d.N; k.K = d.K; k.groups = d.groups; k.batch = d.batch; k.ct = d.ct; k.dtA = d.dtA; k.dtA2 = d.dtA2;
  k.dtB = d.dtB; k.dtC = d.dtC; k.dtC2 = d.dtC2; k.dtAux = d.dtAux; k.dtBias = d.dtBias; k.transA = d.transA;
  k.transB = d.transB; k.act = d.act; k.act_grad = d.act_grad; k.splitk = d.splitk; k.kconcat = d.kconcat;
  k.accumulate = d.accumulate; k.dtB2 = d.dtB2; k.alpha = d.alpha; k.row_fill = d.row_fill; k.lda = d.lda; k.ldb = d.ldb;
  k.ldc = d.ldc; k.strideA = d.strideA; k.strideB = d.strideB; k.strideC = d.strideC; k.row_scale = d.row_scale;
  k.row_fill_flag = d.row_fill_flag; k.mask_out = d.mask_out; k.drop = d.drop; k.xcd_order = 0;
  for (int g = 0; g < PQ3D_MAX_GROUPS; ++g) {
    const bool on = g < d.groups;
    k.gp[g].A = on ? d.A[g] : nullptr; k.gp[g].A2 = on ? d.A2[g] : nullptr; k.gp[g].B = on ? d.B[g] : nullptr;
    k.gp[g].B2 = on ? d.B2[g] : nullptr; k.gp[g].bias = on ? d.bias[g] : nullptr; k.gp[g].aux = on ? d.aux[g] : nullptr;
    k.gp[g].C = on ? d.C[g] : nullptr; k.gp[g].C2 = on ? d.C2[g] : nullptr; k.gp[g].row_mask = on ? d.row_mask[g] : nullptr;
    k.gp[g].colsum = on ? d.colsum[g] : nullptr;
  }
  return k;
}

// ---- XCD-aware tile order (big launches) ---------------------------------------------------------------------------
// The dispatcher deals the workgroups of a grid to the 8 XCDs round-robin in linear-id order (x fastest), and every XCD has
// its own 4 MB L2.  In hardware order the workgroups that share an operand tile (the n-tiles of one 64-row slab of A; the
// 36 output tiles of one k-split of a weight gradient) land on 8 different L2s or run far apart in time, so a launch with
// more workgroups than the chip holds at once re-fetches the shared operand once per sharer: measured (rocprofv3 FETCH_SIZE,
// shipped stage-2 shape) 6-12 x the algorithmic bytes, 6 TB/s of fabric traffic for a 320 TFLOP/s weight-gradient GEMM.
// With xcd_order the workgroups ONE XCD receives (linear ids = xcd mod 8, in dispatch order) walk a contiguous range of
// logical tiles, ordered: blocks of `zrun` z-planes slowest, then SUPER-ROWS of `sr` x-tiles; inside a super-row x
// fastest, then the virtual column index (y, z inside the block).
//   sr    1 when operand B of one z-plane fits an L2 comfortably (<= 2.5 MB: a 768 x 768 fp32 weight): plain
//         column-fastest order, A and B are both fetched once per XCD (measured at M = 10240, N = K = 768, 3 planes:
//         1123 MB in hardware order, 491 MB with sr = 8, 274 MB with sr = 1).  8 otherwise: the ~150 workgroups an XCD
//         has in flight cover 8 row tiles x ~19 column tiles that stream k in step -- each A slab is fetched once, each
//         B slab once per super-row (FFN up-projection, B = 6.3 MB: 442 MB hardware order, 854 MB with sr = 1, 162 MB
//         with sr = 8).
//   zrun  consecutive z-planes (groups) that read the SAME row operand (the hoisted K/V projections: one memory, 8
//         weight matrices; their weight gradients) are walked as extra columns of one plane, so the shared slab is
//         fetched once per run instead of once per group.
// Host side: xcd_order_for() turns it on for launches of >= PQ3D_XCD_MIN workgroups (smaller launches are co-resident
// as a whole; the index arithmetic would only add latency there).
#ifndef PQ3D_XCD_MIN
#define PQ3D_XCD_MIN 1024   // measured: 512 / 1024 / 2048 within noise of each other at c2 and c5, 1024 best at c4 (-2.2 %), s1, s2
#endif
struct TileIdx { int x, y, z; };
#ifdef __HIPCC__
// xcd_order: 0 = hardware order, else sr | zrun << 8
PQ_DEV TileIdx tile_index(const int xcd_order) {
  TileIdx t = {(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
  if (!xcd_order) return t;   // uniform
  const int sr = xcd_order & 255, zr = xcd_order >> 8;
  if (zr < 1 || (int)gridDim.z % zr) return t;   // a run length that does not divide the z extent would map tiles out of range
  const int gx = gridDim.x, gy = gridDim.y, gyv = gy * zr, total = gx * gy * (int)gridDim.z;
  const int id = t.x + gx * (t.y + gy * t.z), xcd = id & 7, per = total >> 3, rem = total & 7;
  int l = xcd * per + min(xcd, rem) + (id >> 3);   // XCD j owns logical tiles [j * per + min(j, rem), ...): rem XCDs hold one more
  const int zb = l / (gx * gyv);
  l -= zb * gx * gyv;
  const int s = l / (sr * gyv), in = l - s * sr * gyv, rows = min(sr, gx - s * sr);   // the last super-row may be short
  const int yv = in / rows;
  t.x = s * sr + (in - yv * rows);
  const int zi = yv / gy;
  t.y = yv - zi * gy;
  t.z = zb * zr + zi;
  return t;
}
// Per-plane variant for the SMALL launches (gemm_wk, gemm_wktt: <= a few hundred workgroups per z-plane, usually all resident at once): z stays blockIdx.z (the speculative per-group pointer load of those kernels stays right), and inside the plane
// the workgroups one XCD receives (same linear id mod 8) take a contiguous range of the plane's tiles -- y fastest
// (order 1: an XCD owns whole row tiles, the A slab is fetched by one L2) or x fastest (order 255: whole column tiles, the
// B slab).  In hardware order the m-tiles of one weight slab sit on 8 different L2s and every L2 fetches every slab:
// measured 42 MB per FFN down-projection launch at config 2 for 8.5 MB of operands (profiles/pmc_traffic_r04_c2.json).
PQ_DEV TileIdx tile_index_plane(const int order) {
  TileIdx t = {(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
  if (!order) return t;   // uniform
  const int gx = gridDim.x, gy = gridDim.y, ps = gx * gy;
  const int id = t.x + gx * t.y, r = id & 7, per = ps >> 3, rem = ps & 7;
  const int l = r * per + min(r, rem) + (id >> 3);
  if ((order & 255) == 1) { t.x = l / gy; t.y = l - t.x * gy; }
  else { t.y = l / gx; t.x = l - t.y * gx; }
  return t;
}
#endif
// Order of tile_index_plane() that gives one XCD the smaller operand footprint; a_tile / b_tile: bytes of the operand slab
// of one row tile / one column tile.  0 (hardware order) for planes of fewer than 16 tiles.
inline int plane_xcd_order(int gx, int gy, long a_tile, long b_tile) {
#ifdef PQ3D_XCD_PLANE_OFF   // A/B measurement builds (tools/build_variant.py)
  return 0;
#else
  const int ps = gx * gy;
  if (ps < 16) return 0;
  const int T = (ps + 7) / 8;
  const long rows_y = (T + gy - 1) / gy + (T % gy ? 1 : 0), cols_y = T < gy ? T : gy;
  const long fy = (rows_y < gx ? rows_y : gx) * a_tile + cols_y * b_tile;
  const long cols_x = (T + gx - 1) / gx + (T % gx ? 1 : 0), rows_x = T < gx ? T : gx;
  const long fx = rows_x * a_tile + (cols_x < gy ? cols_x : gy) * b_tile;
  return (fy <= fx ? 1 : 255) | (1 << 8);
#endif
}
// workgroups: of the launch; b_plane_bytes: operand B of one z-plane; zrun: z-planes per run sharing operand A (must divide the z extent)
inline int xcd_order_for(long workgroups, long b_plane_bytes, int zrun) {
#ifdef PQ3D_XCD_OFF   // A/B measurement builds (tools/build_variant.py)
  return 0;
#else
  if (workgroups < PQ3D_XCD_MIN) return 0;
  if (zrun > 0xFFFFFF) zrun = 1;   // packed as sr | zrun << 8 (the kernel also refuses a run that does not divide gridDim.z)
  const int sr = b_plane_bytes * (zrun > 1 ? zrun : 1) <= (5L << 19) ? 1 : 8;
  return sr | ((zrun > 1 ? zrun : 1) << 8);
#endif
}
// Planes per run that share the row operand of a plain grouped launch (one z-plane per group): A and its addend A2.
inline int uniform_run(const void* const* p, int n);
inline int shared_a_run(const pq3d_gemm_desc& d) {
  if (d.kconcat > 1 || d.batch > 1 || d.splitk > 1 || d.groups <= 1) return 1;
  const int r = uniform_run(d.A, d.groups);
  return (r > 1 && uniform_run(d.A2, d.groups) % r == 0) ? r : 1;
}
// Longest run length r (dividing n) such that every block of r consecutive pointers is one pointer.
inline int uniform_run(const void* const* p, int n) {
  if (n <= 1) return 1;
  int r = 1;
  while (r < n && p[r] == p[0]) ++r;
  if (n % r) return 1;
  for (int b = 0; b < n; b += r)
    for (int i = 1; i < r; ++i)
      if (p[b + i] != p[b]) return 1;
  return r;
}

#ifdef PQ3D_NO_KARG_PIN   // A/B measurement builds
#define ATTN_KARG_PIN(d)
#define ATTN_KARG_PIN_BWD(d)
#else
#define ATTN_KARG_PIN(d)                                                                                                     \
  asm volatile("" ::"s"((d).B), "s"((d).H), "s"((d).Lq), "s"((d).Lk), "s"((d).zero_attn), "s"((d).mask_bmod), "s"((d).scale),    \
               "s"((d).q_sb), "s"((d).q_sl), "s"((d).q_sh), "s"((d).k_sb), "s"((d).k_sl), "s"((d).k_sh), "s"((d).v_sb),       \
               "s"((d).v_sl), "s"((d).v_sh), "s"((d).o_sb), "s"((d).o_sl), "s"((d).o_sh), "s"((d).q), "s"((d).k), "s"((d).v),  \
               "s"((d).o), "s"((d).lse), "s"((d).kpm), "s"((d).mask), "s"((d).row_open), "s"((d).bias), "s"((d).ksplit),        \
               "s"((d).ws))
#define ATTN_KARG_PIN_BWD(d)                                                                                                 \
  asm volatile("" ::"s"((d).dout), "s"((d).dq), "s"((d).dk), "s"((d).dv), "s"((d).delta), "s"((d).dbias), "s"((d).drop.p),      \
               "s"((d).drop.seed), "s"((d).drop_bmod))
#endif

// all-lanes sum / max of a wave with DPP quad / row permutes and the lane-half swaps: no LDS-crossbar shuffles
// (ds_bpermute: ~6 dependent LDS round trips per reduction, which dominated the one-row-per-wave kernels)
PQ_DEV float dpp_xor_f(float v, int which) {
  const int x = __float_as_int(v);
  int y;
  if (which == 0) y = __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true);         // quad_perm [1,0,3,2]
  else if (which == 1) y = __builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
  else if (which == 2) y = __builtin_amdgcn_mov_dpp(x, 0x141, 0xF, 0xF, true);   // row_half_mirror
  else y = __builtin_amdgcn_mov_dpp(x, 0x140, 0xF, 0xF, true);                   // row_mirror
  return __int_as_float(y);
}
typedef unsigned u32pair_s __attribute__((ext_vector_type(2)));
PQ_DEV float wave_max(float v) {
#pragma unroll
  for (int k = 0; k < 4; ++k) v = fmaxf(v, dpp_xor_f(v, k));
  u32pair_s a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  u32pair_s b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
PQ_DEV float wave_sum(float v) {
#pragma unroll
  for (int k = 0; k < 4; ++k) v += dpp_xor_f(v, k);
  u32pair_s a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  u32pair_s b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// ---------------------------------------------------------------------------------------------
// Counter-based dropout (definition in include/pq3d_hip.h).  One 32-bit hash word decides TWO neighbouring
// columns (16 bits each), so a kernel that owns consecutive columns pays ~5 integer ops per element.
// ---------------------------------------------------------------------------------------------
PQ_DEV uint32_t drop_fin(uint32_t h) {
  h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
  return h;
}
struct DropState {
  uint32_t k0, k1, thresh, half;
  float scale;
};
PQ_DEV DropState drop_init(const pq3d_dropout& dr, uint32_t site_add, long cols) {
  DropState s;
  const uint64_t seed = *dr.seed;
  s.k0 = drop_fin((uint32_t)seed ^ ((dr.site + site_add) * 0x9E3779B1u));
  s.k1 = drop_fin(s.k0 + (uint32_t)(seed >> 32) + 0x85ebca6bu);
  s.thresh = (uint32_t)(dr.p * 65536.f + 0.5f);
  s.half = (uint32_t)((cols + 1) >> 1);
  s.scale = 1.f / (1.f - dr.p);
  return s;
}
PQ_DEV uint32_t drop_word(const DropState& s, uint32_t row, uint32_t colpair) {
  uint32_t h = (row * s.half + colpair) ^ s.k0;
  h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h += s.k1; h *= 0x846ca68bu; h ^= h >> 16;
  return h;
}
PQ_DEV bool drop_keep_lo(const DropState& s, uint32_t w) { return (w & 0xffffu) >= s.thresh; }
PQ_DEV bool drop_keep_hi(const DropState& s, uint32_t w) { return (w >> 16) >= s.thresh; }
PQ_DEV bool drop_keep(const DropState& s, uint32_t row, uint32_t col) {
  const uint32_t w = drop_word(s, row, col >> 1);
  return (col & 1) ? drop_keep_hi(s, w) : drop_keep_lo(s, w);
}
PQ_DEV bool drop_on(const pq3d_dropout& dr) { return dr.p > 0.f && dr.seed != nullptr; }

// Zero-fill of up to PQ_ZERO_MAX fp32 buffers in ONE kernel launch (misc.hip).  Used instead of hipMemsetAsync: memset
// nodes of a captured HIP graph were observed to take effect only on the first replay on this stack (ROCm 7.x; see
// tools/probes/graph_memset_probe.py), which silently corrupted split-K / atomics outputs of later replays.
#define PQ_ZERO_MAX 64
struct ZeroList {
  int n = 0;
  float* ptr[PQ_ZERO_MAX];
  long count[PQ_ZERO_MAX];
  void add(void* p, long cnt) { if (p && cnt > 0 && n < PQ_ZERO_MAX) { ptr[n] = (float*)p; count[n] = cnt; ++n; } }
  bool full() const { return n >= PQ_ZERO_MAX; }
};
int pq3d_zero_launch(const ZeroList& z, hipStream_t s);   // returns 0 or a hipError_t (error text set)

// ---------------------------------------------------------------------------------------------
// Device of a call ("take the device from its inputs").  Every entry point that launches work takes the stream
// it launches on; the stream's device is made current for the duration of the call and the caller's device restored
// afterwards (hipSetDevice is per host thread: nothing process-wide changes, PyTorch's autograd thread and the caller's
// thread do not see each other's setting).  With the NULL stream the device comes from the first device pointer of the
// call instead.  Single-device processes (the normal one-process-per-GPU deployment with one visible device) skip all of it.
// ---------------------------------------------------------------------------------------------
int pq3d_visible_devices();   // api.cpp: cached hipGetDeviceCount
struct PqDeviceGuard {
  int prev = -1;
  PqDeviceGuard(void* stream, const void* ptr) {
    if (pq3d_visible_devices() <= 1) return;
    int want = -1, cur = 0;
    if (stream) {
      hipDevice_t dv;
      if (hipStreamGetDevice((hipStream_t)stream, &dv) == hipSuccess) want = (int)dv; else (void)hipGetLastError();
    } else if (ptr) {
      hipPointerAttribute_t a;
      if (hipPointerGetAttributes(&a, ptr) == hipSuccess && a.type == hipMemoryTypeDevice) want = a.device; else (void)hipGetLastError();
    }
    if (want < 0 || hipGetDevice(&cur) != hipSuccess || cur == want) return;
    if (hipSetDevice(want) == hipSuccess) prev = cur;
  }
  ~PqDeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
  PqDeviceGuard(const PqDeviceGuard&) = delete;
};
#define PQ_DEVICE_GUARD(stream, ptr) PqDeviceGuard pq_device_guard_((void*)(stream), (const void*)(ptr))

// > 64 KB of dynamic LDS is a per-DEVICE function attribute: set it once per (kernel, device), race-free.
#include <atomic>
template <typename K> inline int pq3d_enable_big_lds(K kern, int bytes, std::atomic<unsigned>& done) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned bit = 1u << (dev & 31);
  if (!(done.load(std::memory_order_acquire) & bit)) {
    const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    done.fetch_or(bit, std::memory_order_release);
  }
  return 0;
}

// host-side error plumbing (api.cpp)
extern "C" void pq3d_set_error(const char* msg);
#define PQ_CHECK_ARG(cond, msg)      \
  do {                               \
    if (!(cond)) {                   \
      pq3d_set_error(msg);           \
      return PQ3D_ERR_ARG;           \
    }                                \
  } while (0)
#define PQ_CHECK_DROP(dr, rows, cols, what)                                                                   \
  PQ_CHECK_ARG(!((dr).p > 0.f && (dr).seed) || ((dr).p < 1.f && (double)(rows) * (double)(((cols) + 1) / 2) < 4294967296.0), \
               what ": dropout needs p < 1 and rows * ceil(cols/2) < 2^32")
#define PQ_LAUNCH_CHECK()                                  \
  do {                                                     \
    hipError_t e_ = hipGetLastError();                     \
    if (e_ != hipSuccess) {                                \
      pq3d_set_error(hipGetErrorString(e_));               \
      return (int)e_;                                      \
    }                                                      \
  } while (0)
