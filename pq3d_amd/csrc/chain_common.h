// Shared pieces of the one-launch row-local chains (chain_ffn.hip, chain_ca.hip): a group of 8 workgroups on ONE XCD owns a
// row tile through several dependent steps and hands rows over through that XCD's L2 (stores -> s_waitcnt vmcnt(0) -> one flag
// word per member; consumers poll the flags and read the rows with L1-bypassing sc1 loads).  See chain_ffn.hip for the design
// notes and the measurements.  The GEMM pieces are gemm_wk.hip's 32 x 64 x 256 plan, instruction for instruction.
#pragma once
#include <cstdlib>

#include "common.h"

namespace {

constexpr int CT = 512;                     // threads per workgroup (8 waves)
constexpr int TM = 32, TN = 64, KC = 256;   // tile rows / columns, k elements staged at once (= gemm_wk's 32 x 64 x 256 plan)
constexpr int LDR = KC + 8, CLD = TN + 4;
constexpr int G = 8;                        // workgroups per row tile
constexpr int D = 256;                      // model width (LayerNorm rows: 4 values per lane)
constexpr int SPIN_LIMIT = 1 << 20;

typedef __attribute__((address_space(3))) unsigned char lds_b_t;

struct Ctx {
  bf16_t *Ah, *Al, *Bh, *Bl;
  float* Ct;
  int tid, lane, wave, li, lg, wm, wn;
};

PQ_DEV void split_hi_lo(const float* v, u32x4& hi, u32x4& lo) {
  hi = pack_frag<bf16_t>(v);
  float w[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    w[2 * j] = v[2 * j] - __uint_as_float(hi[j] << 16);
    w[2 * j + 1] = v[2 * j + 1] - __uint_as_float(hi[j] & 0xffff0000u);
  }
  lo = pack_frag<bf16_t>(w);
}

// 8 consecutive floats; SC1: L1-bypassing (data written by another CU of this XCD during this launch)
template <bool SC1> PQ_DEV void load8(const float* base, long off, float (&v)[8]) {
  if constexpr (SC1) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7ffffff0, 0x00020000);
    const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off * 4), 0, 16);
    const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off * 4) + 16, 0, 16);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = __uint_as_float(a[j]); v[4 + j] = __uint_as_float(b[j]); }
  } else {
    const float4 a = *(const float4*)(base + off), b = *(const float4*)(base + off + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
}
template <bool SC1> PQ_DEV void load4(const float* base, long off, float (&v)[4]) {
  if constexpr (SC1) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7ffffff0, 0x00020000);
    const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off * 4), 0, 16);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = __uint_as_float(a[j]);
  } else {
    const float4 a = *(const float4*)(base + off);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  }
}

// ---- operand staging: [rows][256 k] fp32 -> hi / lo bf16 planes in LDS (gemm_wk's put())
struct RawA { float v[2][8]; };   // 32 rows x 32 chunks = 1024 chunks / 512 threads
struct RawB { float v[4][8]; };   // 64 rows x 32 chunks = 2048 chunks
template <bool SC1> PQ_DEV void issue_a(const Ctx& c, RawA& r, const float* A, int lda, int m0, int R, int k0) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ch = c.tid + i * CT, row = ch >> 5, k = k0 + (ch & 31) * 8;
    load8<SC1>(A, (long)min(m0 + row, R - 1) * lda + k, r.v[i]);
  }
}
PQ_DEV void put_a(const Ctx& c, const RawA& r) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ch = c.tid + i * CT, o = (ch >> 5) * LDR + (ch & 31) * 8;
    u32x4 hi, lo;
    split_hi_lo(r.v[i], hi, lo);
    *(u32x4*)&c.Ah[o] = hi;
    *(u32x4*)&c.Al[o] = lo;
  }
}
PQ_DEV void issue_b(const Ctx& c, RawB& r, const float* W, int ldb, int n0, int k0) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ch = c.tid + i * CT, row = ch >> 5, k = k0 + (ch & 31) * 8;
    load8<false>(W, (long)(n0 + row) * ldb + k, r.v[i]);
  }
}
PQ_DEV void put_b(const Ctx& c, const RawB& r) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ch = c.tid + i * CT, o = (ch >> 5) * LDR + (ch & 31) * 8;
    u32x4 hi, lo;
    split_hi_lo(r.v[i], hi, lo);
    *(u32x4*)&c.Bh[o] = hi;
    *(u32x4*)&c.Bl[o] = lo;
  }
}
// 8 k-steps of one staged chunk into this wave's 16 x 16 accumulator (gemm_wk's loop at NJ = 1)
PQ_DEV void mma_chunk(const Ctx& c, f32x4& acc) {
#pragma unroll
  for (int ks = 0; ks < KC / 32; ++ks) {
    const int oa = (c.wm + c.li) * LDR + ks * 32 + c.lg * 8, ob = (c.wn + c.li) * LDR + ks * 32 + c.lg * 8;
    const u32x4 ah = *(const u32x4*)&c.Ah[oa], al = *(const u32x4*)&c.Al[oa];
    const u32x4 bh = *(const u32x4*)&c.Bh[ob], bl = *(const u32x4*)&c.Bl[ob];
    Mma<bf16_t>::mma(acc, al, bh);
    Mma<bf16_t>::mma(acc, ah, bl);
    Mma<bf16_t>::mma(acc, ah, bh);
  }
}
// (acc + bias) [relu] -> C, rows leave in 16-byte pieces through the transposed LDS tile (gemm_wk's epilogue at alpha = 1)
PQ_DEV void store_tile(const Ctx& c, const f32x4& acc, const float* bias, int n0, bool relu, float* C, int ldc, int m0, int R) {
  const float bcol = bias ? bias[n0 + c.wn + c.li] : 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) c.Ct[(c.wm + c.lg * 4 + r) * CLD + c.wn + c.li] = (acc[r] + bcol) * 1.f;
  __syncthreads();
  const int lrow = c.tid >> 4, lcol = (c.tid & 15) * 4, row = m0 + lrow;
  float4 t = *(const float4*)&c.Ct[lrow * CLD + lcol];
  if (relu) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
  if (row < R) *(float4*)(C + (long)row * ldc + n0 + lcol) = t;
  __syncthreads();   // the next tile reuses Ct (and the B planes)
}

// ---- hand-off inside the group: publish `target`, wait until every member has
PQ_DEV void handoff(const Ctx& c, unsigned* mine, unsigned* group, unsigned target, int* err) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // write-through L1: acknowledged stores are in the XCD's L2
  __syncthreads();
  if (c.wave == 0) {
    if (c.lane == 0) __hip_atomic_store(mine, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool ok = true;
    int spins = 0;
    do {
      const unsigned v = c.lane < G ? __hip_atomic_load(group + c.lane * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target;
      ok = (int)(v - target) >= 0;
    } while (!__all((int)ok) && ++spins < SPIN_LIMIT);
    if (!__all((int)ok) && c.lane == 0 && err) *err = 1;
  }
  __syncthreads();
}

// ---- LayerNorm rows (norm.hip's add_ln_fwd at d = 256: lane owns columns 4 lane .. 4 lane + 3)
struct RowStats { float mean, rstd; };
PQ_DEV RowStats row_stats4(const float (&v)[4], float eps) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) s += v[j];
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) { const float t = v[j] - mean; q += t * t; }
  const float var = wave_sum(q) / (float)D;
  return {mean, 1.f / sqrtf(var + eps)};
}
// y = LN(x + (o_0 + ... + o_{nsum-1})); the sum is kept in osum when given
template <bool SC1X>
PQ_DEV void ln_row(const Ctx& c, long row, const float* x, const float* const* o, int nsum, long ostride, const float* gamma,
                   const float* beta, float eps, float* osum, float* y, float* mean, float* rstd) {
  const long base = row * D + c.lane * 4;
  float xr[4], ov[4], v[4], gm[4], bt[4], out[4];
  load4<SC1X>(x, base, xr);
  load4<true>(o[0], base, ov);
  for (int p = 1; p < nsum; ++p) {
    float t[4];
    load4<true>(o[0] + p * ostride, base, t);
#pragma unroll
    for (int j = 0; j < 4; ++j) ov[j] += t[j];
  }
  if (osum) *(float4*)(osum + base) = make_float4(ov[0], ov[1], ov[2], ov[3]);
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = xr[j] + ov[j];
  const RowStats st = row_stats4(v, eps);
  load4<false>(gamma, c.lane * 4, gm);
  load4<false>(beta, c.lane * 4, bt);
#pragma unroll
  for (int j = 0; j < 4; ++j) { out[j] = 0.f; out[j] += 1.f * ((v[j] - st.mean) * st.rstd * gm[j] + bt[j]); }
  if (c.lane == 0) { mean[row] = st.mean; rstd[row] = st.rstd; }
  *(float4*)(y + base) = make_float4(out[0], out[1], out[2], out[3]);
}

// ---- a [rows of the group] x [3 x 256] projection on six members (steps of chain_ca / chain_ffn): member j < 6 owns group
// g = j / 2 and columns [128 (j & 1), + 128) of it; wave = 16 rows x 32 columns per row tile (2 accumulators), the weight in two
// k slabs of 128.  X3: split-bf16 with fp32 operands, optional per-group addend A2 (gemm_wk's HA2 staging: a + s2 a2, s2 = 0 with
// the operand itself as the addend where a group has none); !X3: single bf16, A stored as bf16.
constexpr int PN = 128, PK = 128, PLD = PK + 8, PCL = PN + 4;
template <int NRT> constexpr size_t proj_lds(bool x3) {
  const size_t b = (size_t)(x3 ? 2 : 1) * PN * PLD * 2, ct = (size_t)TM * PCL * 4;
  return (size_t)NRT * (x3 ? 2 : 1) * TM * LDR * 2 + (b > ct ? b : ct);
}
// out: fp32 or bf16 rows of leading dimension D; sc1a: the A rows were written by other members during this launch
// the member's two weight slabs (128 rows x 16 chunks of 8 floats = 2048 chunks, 4 per thread, each): may be requested long
// before the step runs (weights do not depend on the chain)
PQ_DEV void proj_issue_w(const Ctx& c, int j, int ng, const float* const* W, RawB (&wb)[2]) {
  if (j >= 2 * ng) return;
  const int g = j >> 1, n0 = (j & 1) * PN;
#pragma unroll
  for (int l = 0; l < 2; ++l)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ch = c.tid + i * CT;
      load8<false>(W[g], (long)(n0 + (ch >> 4)) * D + l * PK + (ch & 15) * 8, wb[l].v[i]);
    }
}
template <int NRT, bool X3, bool SC1A, typename TO>
PQ_DEV void proj_3x256(const Ctx& c, unsigned char* smem, int j, int ng, int m0, int R, const void* const* A, const float* const* A2,
                       const float* const* W, const float* const* bias, TO* const* out, RawB (&wb)[2], bool preloaded,
                       int relu_mask = 0) {   // relu_mask: bit g set -> max(., 0) on group g's outputs
  if (j >= 2 * ng) return;   // (uniform: the whole workgroup)
  constexpr int PL = X3 ? 2 : 1;
  const int g = j >> 1, n0 = (j & 1) * PN;
  bf16_t* const Ap = (bf16_t*)smem;                       // [NRT][PL planes][32][LDR]: A, whole K
  bf16_t* const Bp = Ap + NRT * PL * TM * LDR;            // [PL planes][128][PLD]: one k slab of W
  float* const Ct = (float*)Bp;                           // [32][PCL], over the slab once it is dead
  const int wr = (c.wave >> 2) * 16, wc = (c.wave & 3) * 32;
  if (!preloaded) proj_issue_w(c, j, ng, W, wb);
  // A planes
#pragma unroll
  for (int t = 0; t < NRT; ++t) {
    if constexpr (X3) {
      RawA ra, ra2;
      const float* a = (const float*)A[g];
      issue_a<SC1A>(c, ra, a, D, m0 + t * TM, R, 0);
      const bool has2 = A2 && A2[g];
      if (has2) issue_a<false>(c, ra2, A2[g], D, m0 + t * TM, R, 0);
      const float s2 = has2 ? 1.f : 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ch = c.tid + i * CT, o = (ch >> 5) * LDR + (ch & 31) * 8;
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = ra.v[i][k] + s2 * (has2 ? ra2.v[i][k] : ra.v[i][k]);
        u32x4 hi, lo;
        split_hi_lo(v, hi, lo);
        *(u32x4*)&Ap[(t * 2) * TM * LDR + o] = hi;
        *(u32x4*)&Ap[(t * 2 + 1) * TM * LDR + o] = lo;
      }
    } else {
      const bf16_t* a = (const bf16_t*)A[g];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ch = c.tid + i * CT;
        *(u32x4*)&Ap[t * TM * LDR + (ch >> 5) * LDR + (ch & 31) * 8] =
            *(const u32x4*)(a + (long)min(m0 + t * TM + (ch >> 5), R - 1) * D + (ch & 31) * 8);
      }
    }
  }
  f32x4 acc[NRT][2];
#pragma unroll
  for (int t = 0; t < NRT; ++t) { acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int l = 0; l < 2; ++l) {
    if (l > 0) __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ch = c.tid + i * CT, o = (ch >> 4) * PLD + (ch & 15) * 8;
      if constexpr (X3) {
        u32x4 hi, lo;
        split_hi_lo(wb[l].v[i], hi, lo);
        *(u32x4*)&Bp[o] = hi;
        *(u32x4*)&Bp[PN * PLD + o] = lo;
      } else *(u32x4*)&Bp[o] = pack_frag<bf16_t>(wb[l].v[i]);
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < PK / 32; ++ks) {
      u32x4 bh[2], bl[2];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const int ob = (wc + nb * 16 + c.li) * PLD + ks * 32 + c.lg * 8;
        bh[nb] = *(const u32x4*)&Bp[ob];
        if constexpr (X3) bl[nb] = *(const u32x4*)&Bp[PN * PLD + ob];
      }
#pragma unroll
      for (int t = 0; t < NRT; ++t) {
        const int oa = (wr + c.li) * LDR + l * PK + ks * 32 + c.lg * 8;
        const u32x4 ah = *(const u32x4*)&Ap[(t * PL) * TM * LDR + oa];
        if constexpr (X3) {
          const u32x4 al = *(const u32x4*)&Ap[(t * 2 + 1) * TM * LDR + oa];
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            Mma<bf16_t>::mma(acc[t][nb], al, bh[nb]);
            Mma<bf16_t>::mma(acc[t][nb], ah, bl[nb]);
            Mma<bf16_t>::mma(acc[t][nb], ah, bh[nb]);
          }
        } else {
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) Mma<bf16_t>::mma(acc[t][nb], ah, bh[nb]);
        }
      }
    }
  }
  float bcol[2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) bcol[nb] = bias[g] ? bias[g][n0 + wc + nb * 16 + c.li] : 0.f;
#pragma unroll
  for (int t = 0; t < NRT; ++t) {
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) Ct[(wr + c.lg * 4 + r) * PCL + wc + nb * 16 + c.li] = (acc[t][nb][r] + bcol[nb]) * 1.f;
    __syncthreads();
    const int orow = c.tid >> 4, row = m0 + t * TM + orow;
#pragma unroll
    for (int qd = 0; qd < 2; ++qd) {
      const int col = qd * 64 + (c.tid & 15) * 4;
      float4 v = *(const float4*)&Ct[orow * PCL + col];
      if ((relu_mask >> g) & 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      if (row < R) {
        if constexpr (sizeof(TO) == 4) *(float4*)((float*)out[g] + (long)row * D + n0 + col) = v;
        else *(u32x2*)((bf16_t*)out[g] + (long)row * D + n0 + col) = (u32x2){pack_bf2(v.x, v.y), pack_bf2(v.z, v.w)};
      }
    }
  }
}

// ---- the backward's analogue of proj_3x256: out_g = A_g W_g (+ aux_g) with W_g [256 k][256 n] row-major (input gradients dX = dY W;
// pq3d_gemm's transB layout), single-bf16 operands.  Member j < 2 ng owns group j / 2 and columns [128 (j & 1), + 128); the weight
// in two [128 k][128 n] slabs read through the transposing LDS load; wave = 16 rows x 32 columns per row tile.
constexpr int TPK = 128, TPN = 128, TPLD = TPN + 8, TPCL = TPN + 4;
template <int NRT> constexpr size_t tproj_lds() {
  const size_t b = (size_t)TPK * TPLD * 2, ct = (size_t)TM * TPCL * 4;
  return (size_t)NRT * TM * LDR * 2 + (b > ct ? b : ct);
}
PQ_DEV void tproj_issue_w(const Ctx& c, int j, int ng, const float* const* W, RawB (&wb)[2]) {
  if (j >= 2 * ng) return;
  const int g = j >> 1, n0 = (j & 1) * TPN;
#pragma unroll
  for (int l = 0; l < 2; ++l)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ch = c.tid + i * CT;   // [128 k][16 chunks of 8 n]
      load8<false>(W[g], (long)(l * TPK + (ch >> 4)) * D + n0 + (ch & 15) * 8, wb[l].v[i]);
    }
}
PQ_DEV u32x4 tp_km_frag(const bf16_t* tile, int ldk, int r0, int ks, int li, int lg) {   // gemm_common.h's km_frag
  typedef short v4s_t __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) v4s_t lds_v4s_t;
  const bf16_t* p0 = tile + (ks * 32 + 8 * lg + (li >> 2)) * ldk + r0 + 4 * (li & 3);
  const v4s_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)p0);
  const v4s_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_t*)(p0 + 4 * ldk));
  const u32x2 lo = __builtin_bit_cast(u32x2, a), hi = __builtin_bit_cast(u32x2, b);
  return (u32x4){lo.x, lo.y, hi.x, hi.y};
}
template <int NRT, bool SC1A, typename TO>
PQ_DEV void tproj_3x256(const Ctx& c, unsigned char* smem, int j, int ng, int m0, int R, const float* const* A, const float* const* aux,
                        TO* const* out, RawB (&wb)[2]) {
  if (j >= 2 * ng) return;   // (uniform: the whole workgroup)
  const int g = j >> 1, n0 = (j & 1) * TPN;
  bf16_t* const Ap = (bf16_t*)smem;                       // [NRT][32][LDR]: A as bf16, whole K
  bf16_t* const Bp = Ap + NRT * TM * LDR;                 // [128 k][TPLD]: one k slab of W
  float* const Ct = (float*)Bp;
  const int wr = (c.wave >> 2) * 16, wc = (c.wave & 3) * 32;
#pragma unroll
  for (int t = 0; t < NRT; ++t) {
    RawA ra;
    issue_a<SC1A>(c, ra, A[g], D, m0 + t * TM, R, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ch = c.tid + i * CT;
      *(u32x4*)&Ap[t * TM * LDR + (ch >> 5) * LDR + (ch & 31) * 8] = pack_frag<bf16_t>(ra.v[i]);
    }
  }
  f32x4 acc[NRT][2];
#pragma unroll
  for (int t = 0; t < NRT; ++t) { acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int l = 0; l < 2; ++l) {
    if (l > 0) __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ch = c.tid + i * CT;
      *(u32x4*)&Bp[(ch >> 4) * TPLD + (ch & 15) * 8] = pack_frag<bf16_t>(wb[l].v[i]);
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < TPK / 32; ++ks) {
      u32x4 bh[2];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) bh[nb] = tp_km_frag(Bp, TPLD, wc + nb * 16, ks, c.li, c.lg);
#pragma unroll
      for (int t = 0; t < NRT; ++t) {
        const u32x4 ah = *(const u32x4*)&Ap[t * TM * LDR + (wr + c.li) * LDR + l * TPK + ks * 32 + c.lg * 8];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) Mma<bf16_t>::mma(acc[t][nb], ah, bh[nb]);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < NRT; ++t) {
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) Ct[(wr + c.lg * 4 + r) * TPCL + wc + nb * 16 + c.li] = (acc[t][nb][r] + 0.f) * 1.f;
    __syncthreads();
    const int orow = c.tid >> 4, row = m0 + t * TM + orow;
#pragma unroll
    for (int qd = 0; qd < 2; ++qd) {
      const int col = qd * 64 + (c.tid & 15) * 4;
      float4 v = *(const float4*)&Ct[orow * TPCL + col];
      if (row < R) {
        if (aux && aux[g]) {
          const float4 a = *(const float4*)(aux[g] + (long)row * D + n0 + col);
          v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        if constexpr (sizeof(TO) == 4) *(float4*)((float*)out[g] + (long)row * D + n0 + col) = v;
        else *(u32x2*)((bf16_t*)out[g] + (long)row * D + n0 + col) = (u32x2){pack_bf2(v.x, v.y), pack_bf2(v.z, v.w)};
      }
    }
  }
}

// ---- LayerNorm parameter gradients of a group: every member leaves its column sums ([2][256] floats per LayerNorm) in the
// group's scratch; behind the next hand-off member j adds up columns [32 j, 32 j + 32) of all 8 members and issues the atomics --
// 512 per group and LayerNorm instead of 512 per WORKGROUP (all of them on the same 512 addresses: measured 5.3 us of a 25.5 us
// backward chain at config 2)
PQ_DEV void ln_partials_store(const Ctx& c, float* red, const float (&dg)[4], const float (&db)[4], float* slot) {   // slot: [512] of this member
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    red[c.wave * D + c.lane * 4 + k] = dg[k];
    red[8 * D + c.wave * D + c.lane * 4 + k] = db[k];
  }
  __syncthreads();
  {
    const int col = c.tid & (D - 1), kind = c.tid >> 8;   // 512 threads: gamma columns, then beta columns
    float sv = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) sv += red[kind * 8 * D + w * D + col];
    slot[kind * D + col] = sv;
  }
  __syncthreads();
}
// group: [8 members][stride floats]; off: this LayerNorm's [512] inside a member's record
PQ_DEV void ln_partials_reduce(const Ctx& c, int j, const float* group, int stride, int off, float* dgamma, float* dbeta) {
  if (c.tid < 64) {
    const int col = 32 * j + (c.tid & 31), kind = c.tid >> 5;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)group, 0, 0x7ffffff0, 0x00020000);
    float sv = 0.f;
#pragma unroll
    for (int m = 0; m < G; ++m)
      sv += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (m * stride + off + kind * D + col) * 4, 0, 16));
    unsafeAtomicAdd(kind ? &dbeta[col] : &dgamma[col], sv);
  }
}

// Row tiles per group: 1 while all groups are resident at once (8 workgroups per group, at most one workgroup per CU), else 2.
// The limit leaves 32 of the 256 CUs free: a collective of another stream (RCCL: one workgroup per channel) may hold CUs for the whole
// launch, and a group whose member cannot become resident would make the others spin to the limit.
// PQ3D_CHAIN_NRT=2 forces two (a tuning switch for measurements: half as many groups, each weight slab converted once for 64 rows).
inline int chain_nrt(int row_tiles) {
  static const int forced = [] { const char* e = getenv("PQ3D_CHAIN_NRT"); return e ? atoi(e) : 0; }();
  if (forced == 2) return 2;
  return row_tiles * G <= 224 ? 1 : 2;
}

// in-kernel timeline (probe builds only, tools/probes/chain_timeline.py): thread 0 of workgroup 0 stamps the 100 MHz clock
#ifdef PQ3D_CHAIN_TL
#define CH_TL(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) ((long long*)d.err)[i] = wall_clock64(); } while (0)
#else
#define CH_TL(i) do { } while (0)
#endif

}  // namespace
