// Small-sequence fp32 attention for the decoder's SELF-attention (N_q x N_q scores per scene, N_q <= 128): one workgroup
// per (scene, head) holds the whole problem -- Q, K, V, the score matrix -- in LDS and registers and does everything with
// fp32 FMAs on the vector ALU.  At 100 x 100 x 32 the whole head is 0.64 MFLOP forward / 1.6 MFLOP backward: the general
// streaming kernels of attention.hip (tiles staged through LDS, online softmax, separate dQ and dK/dV recompute kernels,
// exact-f32 MFMA at 1/16 of the bf16 rate) spend 19 us forward / 40 us backward on it, almost all of it fixed pipeline
// latency; here it is one launch each way with no recompute and no staging pipeline.
//   forward : S = scale Q K^T + bias (key padding -> -inf), row softmax, O = P V, lse
//   backward: P from the saved lse, dP = dO V^T, dS = P (dP - rowsum(dO O)); dbias = dS; dQ = scale dS K,
//             dK = scale dS^T Q, dV = P^T dO
// fp32 storage and arithmetic (used by both compute modes: SURVEY 8a rows 8 / 8b); additive bias [B,H,Lq,Lk] and key
// padding supported; 3-D masks, the zero key and attention dropout stay on the general kernels.
#include "common.h"

namespace {

// threads per workgroup: the head is latency-bound (LDS reads feeding short FMA chains), so it wants many waves per SIMD:
// 1024 threads (4 waves / SIMD) at d_h 32; 512 where the per-thread register tiles are larger (d_h 64) or the row blocks
// would drop below the 4 keys a float4 covers (d_h 16).  (256 threads: 44 us forward at config 2; see DESIGN.md.)
template <int DH> struct SN { static constexpr int T = DH == 32 ? 1024 : 512; };

// dst[row * LS + col] = src[row * Lk + col] (or 0) for a [Lq, Lk] fp32 tile, as batches of independent loads: a plain
// "load, store" loop is compiled to one outstanding load at a time (40 dependent global round trips = 40+ us here)
template <int SNT> PQ_DEV void load_tile(float* dst, const float* src, int Lq, int Lk, int LS, int tid) {
  constexpr int U = 5;
  if (src && (Lk & 3) == 0 && ((((uintptr_t)src) & 15) == 0)) {
    const int n4 = Lq * Lk / 4;
    for (int base = tid; base < n4; base += SNT * U) {
      float4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = base + u * SNT;
        v[u] = ((const float4*)src)[min(idx, n4 - 1)];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = base + u * SNT, e = idx * 4;
        if (idx < n4) *(float4*)&dst[(e / Lk) * LS + e % Lk] = v[u];
      }
    }
  } else {
    const int n = Lq * Lk;
    for (int base = tid; base < n; base += SNT * U) {
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = src ? src[min(base + u * SNT, n - 1)] : 0.f;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = base + u * SNT;
        if (e < n) dst[(e / Lk) * LS + e % Lk] = v[u];
      }
    }
  }
}

// all-lanes max / sum of a wave with DPP quad / row permutes and the lane-half swaps (no LDS-crossbar shuffles)
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
PQ_DEV float wave_max_dpp(float v) {
#pragma unroll
  for (int k = 0; k < 4; ++k) v = fmaxf(v, dpp_xor_f(v, k));
  u32pair_s a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  u32pair_s b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
PQ_DEV float wave_sum_dpp(float v) {
#pragma unroll
  for (int k = 0; k < 4; ++k) v += dpp_xor_f(v, k);
  u32pair_s a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  u32pair_s b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

template <int DH>
__global__ __launch_bounds__(SN<DH>::T) void attn_small_fwd_kernel(const pq3d_attn_desc d) {
  constexpr int SNT = SN<DH>::T;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Lq = d.Lq, Lk = d.Lk, LS = ((Lk + 3) & ~3) + 4;   // float4-readable score rows, bank-shifted
  float* Qs = sm;                    // [Lq][DH]
  float* Vs = Qs + Lq * DH;          // [Lk][DH]
  float* S = Vs + Lk * DH;           // [Lq][LS]
  float* Li = S + Lq * LS;           // [Lq] 1 / rowsum
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y, h = blockIdx.x;
  const float* q = (const float*)d.q + (long)b * d.q_sb + (long)h * d.q_sh;
  const float* k = (const float*)d.k + (long)b * d.k_sb + (long)h * d.k_sh;
  const float* v = (const float*)d.v + (long)b * d.v_sb + (long)h * d.v_sh;
  for (int c = tid; c < Lq * (DH / 4); c += SNT) {
    const int r = c / (DH / 4), x = (c % (DH / 4)) * 4;
    *(float4*)&Qs[r * DH + x] = *(const float4*)(q + (long)r * d.q_sl + x);
  }
  for (int c = tid; c < Lk * (DH / 4); c += SNT) {
    const int r = c / (DH / 4), x = (c % (DH / 4)) * 4;
    *(float4*)&Vs[r * DH + x] = *(const float4*)(v + (long)r * d.v_sl + x);
  }
  // ---- scores: thread owns key column j (its K row in registers) and every (tid / 128)-th query row
  const int j = tid & 127, ih = tid >> 7;
  float kr[DH];
  const bool jv = j < Lk;
  {
    const float* kp = k + (long)min(j, Lk - 1) * d.k_sl;
#pragma unroll
    for (int x = 0; x < DH; x += 4) { const float4 t = *(const float4*)(kp + x); kr[x] = t.x; kr[x + 1] = t.y; kr[x + 2] = t.z; kr[x + 3] = t.w; }
  }
  const bool jm = jv ? (d.kpm ? d.kpm[(long)b * Lk + j] != 0 : false) : true;
  const float* bias = d.bias ? d.bias + ((long)b * d.H + h) * Lq * (long)Lk : nullptr;
  // the additive bias tile goes into the score buffer FIRST, as one coalesced sweep with every load in flight at once
  // (read per (row, column) inside the score loop it is a chain of dependent global loads: 57 us instead of ~8)
  load_tile<SNT>(S, bias, Lq, Lk, LS, tid);
  __syncthreads();
  for (int i = ih; i < Lq; i += SNT / 128) {
    float s = 0.f;
#pragma unroll
    for (int x = 0; x < DH; x += 4) {
      const float4 t = *(const float4*)&Qs[i * DH + x];
      s = fmaf(t.x, kr[x], s); s = fmaf(t.y, kr[x + 1], s); s = fmaf(t.z, kr[x + 2], s); s = fmaf(t.w, kr[x + 3], s);
    }
    if (jv) s = s * d.scale + S[i * LS + j];
    if (jv) S[i * LS + j] = jm ? -INFINITY : s;
    else if (j < LS) S[i * LS + j] = 0.f;       // padding columns: read (times 0) by the float4 loops below
  }
  __syncthreads();
  // ---- row softmax: one wave per row
  for (int i = wave; i < Lq; i += SNT / 64) {
    const float a0 = lane < Lk ? S[i * LS + lane] : -INFINITY, a1 = lane + 64 < Lk ? S[i * LS + lane + 64] : -INFINITY;
    const float m = wave_max_dpp(fmaxf(a0, a1));
    const float e0 = lane < Lk ? __expf(a0 - m) : 0.f, e1 = lane + 64 < Lk ? __expf(a1 - m) : 0.f;
    const float l = wave_sum_dpp(e0 + e1);
    if (lane < Lk) S[i * LS + lane] = e0;
    if (lane + 64 < Lk) S[i * LS + lane + 64] = e1;
    if (lane == 0) { Li[i] = 1.f / l; d.lse[((long)b * d.H + h) * Lq + i] = m + logf(l); }
  }
  __syncthreads();
  // ---- O = P V: thread owns channel c and RPT CONSECUTIVE rows: one V read feeds RPT FMAs, P rows are read as float4
  constexpr int RPT = 128 * DH / SNT;     // rows per thread (8 / 16 / 32): 128 rows covered by SNT / DH row blocks
  const int c = tid % DH, i0 = (tid / DH) * RPT;
  float acc[RPT];
#pragma unroll
  for (int r = 0; r < RPT; ++r) acc[r] = 0.f;
  for (int jj = 0; jj < Lk; jj += 4) {
    float vv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) vv[u] = jj + u < Lk ? Vs[(jj + u) * DH + c] : 0.f;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const float4 p = *(const float4*)&S[min(i0 + r, Lq - 1) * LS + jj];   // columns >= Lk hold finite garbage * 0
      acc[r] = fmaf(p.x, vv[0], acc[r]); acc[r] = fmaf(p.y, vv[1], acc[r]);
      acc[r] = fmaf(p.z, vv[2], acc[r]); acc[r] = fmaf(p.w, vv[3], acc[r]);
    }
  }
  float* o = (float*)d.o + (long)b * d.o_sb + (long)h * d.o_sh;
#pragma unroll
  for (int r = 0; r < RPT; ++r)
    if (i0 + r < Lq) o[(long)(i0 + r) * d.o_sl + c] = acc[r] * Li[i0 + r];
}

template <int DH>
__global__ __launch_bounds__(SN<DH>::T) void attn_small_bwd_kernel(const pq3d_attn_desc d) {
  constexpr int SNT = SN<DH>::T;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Lq = d.Lq, Lk = d.Lk, LS = ((Lk + 3) & ~3) + 4;   // float4-readable score rows, bank-shifted
  float* Qs = sm;                    // [Lq][DH]
  float* Gs = Qs + Lq * DH;          // [Lq][DH]   dO
  float* Ks = Gs + Lq * DH;          // [Lk][DH]
  float* P = Ks + Lk * DH;           // [Lq][LS]
  float* dS = P + Lq * LS;           // [Lq][LS]
  float* Dl = dS + Lq * LS;          // [Lq] delta
  float* Ls = Dl + Lq;               // [Lq] lse
  const int tid = threadIdx.x;
  const int b = blockIdx.y, h = blockIdx.x;
  const float* q = (const float*)d.q + (long)b * d.q_sb + (long)h * d.q_sh;
  const float* k = (const float*)d.k + (long)b * d.k_sb + (long)h * d.k_sh;
  const float* v = (const float*)d.v + (long)b * d.v_sb + (long)h * d.v_sh;
  const float* o = (const float*)d.o + (long)b * d.o_sb + (long)h * d.o_sh;
  const float* g = (const float*)d.dout + (long)b * d.o_sb + (long)h * d.o_sh;
  const long sbase = ((long)b * d.H + h) * Lq;
  for (int c = tid; c < Lq * (DH / 4); c += SNT) {
    const int r = c / (DH / 4), x = (c % (DH / 4)) * 4;
    *(float4*)&Qs[r * DH + x] = *(const float4*)(q + (long)r * d.q_sl + x);
    *(float4*)&Gs[r * DH + x] = *(const float4*)(g + (long)r * d.o_sl + x);
  }
  for (int c = tid; c < Lk * (DH / 4); c += SNT) {
    const int r = c / (DH / 4), x = (c % (DH / 4)) * 4;
    *(float4*)&Ks[r * DH + x] = *(const float4*)(k + (long)r * d.k_sl + x);
  }
  for (int i = tid; i < Lq; i += SNT) {   // delta = rowsum(dO * O)
    float s = 0.f;
    for (int x = 0; x < DH; x += 4) {
      const float4 a = *(const float4*)(g + (long)i * d.o_sl + x), t = *(const float4*)(o + (long)i * d.o_sl + x);
      s += a.x * t.x + a.y * t.y + a.z * t.z + a.w * t.w;
    }
    Dl[i] = s;
    Ls[i] = d.lse[sbase + i];
    d.delta[sbase + i] = s;
  }
  // ---- P and dS: thread owns key column j (K and V rows in registers) and every other query row
  const int j = tid & 127, ih = tid >> 7;
  const bool jv = j < Lk;
  float kr[DH], vr[DH];
  {
    const float* kp = k + (long)min(j, Lk - 1) * d.k_sl;
    const float* vp = v + (long)min(j, Lk - 1) * d.v_sl;
#pragma unroll
    for (int x = 0; x < DH; x += 4) {
      const float4 t = *(const float4*)(kp + x), u = *(const float4*)(vp + x);
      kr[x] = t.x; kr[x + 1] = t.y; kr[x + 2] = t.z; kr[x + 3] = t.w;
      vr[x] = u.x; vr[x + 1] = u.y; vr[x + 2] = u.z; vr[x + 3] = u.w;
    }
  }
  const bool jm = jv ? (d.kpm ? d.kpm[(long)b * Lk + j] != 0 : false) : true;
  const float* bias = d.bias ? d.bias + ((long)b * d.H + h) * Lq * (long)Lk : nullptr;
  float* dbias = d.dbias ? d.dbias + ((long)b * d.H + h) * Lq * (long)Lk : nullptr;
  load_tile<SNT>(P, bias, Lq, Lk, LS, tid);   // as the forward
  __syncthreads();
  for (int i = ih; i < Lq; i += SNT / 128) {
    float s = 0.f, dp = 0.f;
#pragma unroll
    for (int x = 0; x < DH; x += 4) {
      const float4 t = *(const float4*)&Qs[i * DH + x], u = *(const float4*)&Gs[i * DH + x];
      s = fmaf(t.x, kr[x], s); s = fmaf(t.y, kr[x + 1], s); s = fmaf(t.z, kr[x + 2], s); s = fmaf(t.w, kr[x + 3], s);
      dp = fmaf(u.x, vr[x], dp); dp = fmaf(u.y, vr[x + 1], dp); dp = fmaf(u.z, vr[x + 2], dp); dp = fmaf(u.w, vr[x + 3], dp);
    }
    if (jv) s = s * d.scale + P[i * LS + j];
    const float p = jm ? 0.f : __expf(s - Ls[i]);
    const float ds = p * (dp - Dl[i]);
    if (jv) {
      P[i * LS + j] = p;
      dS[i * LS + j] = ds;
    } else if (j < LS) {
      P[i * LS + j] = 0.f;
      dS[i * LS + j] = 0.f;
    }
  }
  __syncthreads();
  if (dbias)
    for (int e = tid; e < Lq * Lk; e += SNT) dbias[e] = dS[(e / Lk) * LS + e % Lk];
  // ---- dQ = scale dS K: thread owns channel c and RPT consecutive query rows (as the forward's P V)
  constexpr int RPT = 128 * DH / SNT;
  const int c = tid % DH, i0 = (tid / DH) * RPT;
  float* dq = (float*)d.dq + (long)b * d.q_sb + (long)h * d.q_sh;
  float* dk = (float*)d.dk + (long)b * d.k_sb + (long)h * d.k_sh;
  float* dv = (float*)d.dv + (long)b * d.v_sb + (long)h * d.v_sh;
  {
    float acc[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) acc[r] = 0.f;
    for (int jj = 0; jj < Lk; jj += 4) {
      float kk[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) kk[u] = jj + u < Lk ? Ks[(jj + u) * DH + c] : 0.f;
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        const float4 t = *(const float4*)&dS[min(i0 + r, Lq - 1) * LS + jj];
        acc[r] = fmaf(t.x, kk[0], acc[r]); acc[r] = fmaf(t.y, kk[1], acc[r]);
        acc[r] = fmaf(t.z, kk[2], acc[r]); acc[r] = fmaf(t.w, kk[3], acc[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r)
      if (i0 + r < Lq) dq[(long)(i0 + r) * d.q_sl + c] = acc[r] * d.scale;
  }
  // ---- dK = scale dS^T Q, dV = P^T dO: thread owns channel c and RPT consecutive KEYS; per query row one Q / dO read
  // feeds RPT FMAs each, the dS / P rows are read as float4 over the thread's keys
  {
    const int j0 = i0;
    float ak[RPT], av[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) { ak[r] = 0.f; av[r] = 0.f; }
    if (j0 < LS) {
      for (int i = 0; i < Lq; ++i) {
        const float qv = Qs[i * DH + c], gv = Gs[i * DH + c];
#pragma unroll
        for (int r = 0; r < RPT; r += 4) {
          if (j0 + r < LS) {
            const float4 t = *(const float4*)&dS[i * LS + j0 + r], u = *(const float4*)&P[i * LS + j0 + r];
            ak[r] = fmaf(t.x, qv, ak[r]); ak[r + 1] = fmaf(t.y, qv, ak[r + 1]);
            ak[r + 2] = fmaf(t.z, qv, ak[r + 2]); ak[r + 3] = fmaf(t.w, qv, ak[r + 3]);
            av[r] = fmaf(u.x, gv, av[r]); av[r + 1] = fmaf(u.y, gv, av[r + 1]);
            av[r + 2] = fmaf(u.z, gv, av[r + 2]); av[r + 3] = fmaf(u.w, gv, av[r + 3]);
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r)
      if (j0 + r < Lk) {
        dk[(long)(j0 + r) * d.k_sl + c] = ak[r] * d.scale;
        dv[(long)(j0 + r) * d.v_sl + c] = av[r];
      }
  }
}

bool small_ok(const pq3d_attn_desc& d) {
  return d.ct == PQ3D_F32 && d.dt == PQ3D_F32 && d.Lq >= 1 && d.Lq <= 128 && d.Lk >= 1 && d.Lk <= 128 &&
         (d.dh == 16 || d.dh == 32 || d.dh == 64) && !d.mask && !d.zero_attn && !(d.drop.p > 0.f && d.drop.seed) &&
         d.ksplit <= 1;
}

template <int DH> void launch_small(const pq3d_attn_desc& d, hipStream_t s, bool bwd) {
  const size_t ls = ((d.Lk + 3) & ~3) + 4;
  const size_t fl = bwd ? (size_t)(2 * d.Lq * DH + d.Lk * DH + 2 * d.Lq * ls + 2 * d.Lq)
                        : (size_t)(d.Lq * DH + d.Lk * DH + d.Lq * ls + d.Lq);
  const size_t lds = fl * 4 + 16;
  if (bwd) {
    auto kern = attn_small_bwd_kernel<DH>;
    static bool done = false;
    if (!done) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); done = true; }
    hipLaunchKernelGGL(kern, dim3(d.H, d.B), dim3(SN<DH>::T), lds, s, d);
  } else {
    auto kern = attn_small_fwd_kernel<DH>;
    static bool done = false;
    if (!done) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); done = true; }
    hipLaunchKernelGGL(kern, dim3(d.H, d.B), dim3(SN<DH>::T), lds, s, d);
  }
}

}  // namespace

// fp32 attention with at most 128 queries and keys (the decoder's self-attention): returns false when the call is not of
// that shape (attention.hip's general kernels run instead).
bool pq3d_attn_small_try(const pq3d_attn_desc& d, hipStream_t s, bool bwd) {
  if (!small_ok(d)) return false;
  {   // the whole head must fit the 160 KB of LDS (backward at 128 x 128 x 64 does not)
    const size_t ls = ((d.Lk + 3) & ~3) + 4;
    const size_t fl = bwd ? (size_t)(2 * d.Lq * d.dh + d.Lk * d.dh + 2 * d.Lq * ls + 2 * d.Lq)
                          : (size_t)(d.Lq * d.dh + d.Lk * d.dh + d.Lq * ls + d.Lq);
    if (fl * 4 + 16 > 160 * 1024) return false;
  }
  if (bwd && !(d.dout && d.dq && d.dk && d.dv && d.delta)) return false;
  if (d.dh == 16) launch_small<16>(d, s, bwd);
  else if (d.dh == 32) launch_small<32>(d, s, bwd);
  else launch_small<64>(d, s, bwd);
  return true;
}
