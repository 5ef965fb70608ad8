// Small-sequence fp32 attention for the decoder's SELF-attention (N_q x N_q scores per scene, N_q <= 128), on the vector
// ALU with everything a workgroup needs resident in LDS / registers: no staging pipeline, no online softmax, no recompute
// kernels.  The general streaming kernels of attention.hip spend 19 us forward / 40 us backward on this 0.64 / 1.6 MFLOP
// per head problem (config 2: 64 heads), nearly all of it fixed pipeline latency on 64 workgroups.
//   forward : S = scale Q K^T + bias (key padding -> -inf), row softmax, O = P V, lse
//   backward: P from the saved lse, dP = dO V^T, dS = P (dP - rowsum(dO O)); dbias = dS; dQ = scale dS K,
//             dK = scale dS^T Q, dV = P^T dO
// fp32 storage and arithmetic (both compute modes use it: SURVEY 8a rows 8 / 8b); additive bias [B,H,Lq,Lk] and key
// padding supported; 3-D masks, the zero key and attention dropout stay on the general kernels.
// Lessons built in (measured, tools/probes): per-element global reads inside a loop (bias) become chains of dependent
// round trips -- tiles are swept into LDS / registers with all loads in flight first; one accumulator per thread makes the
// FMA chain latency-bound -- every thread carries 2-8 independent rows; wave reductions use DPP / lane-swap moves, not
// LDS-crossbar shuffles (softmax phase 37.7k -> ~3k cycles per workgroup).
#include "common.h"

namespace {

// Work split: one workgroup per (scene, head, 32-row slice): the forward splits by query rows (independent); the backward
// gives slice r the query rows [32 r, 32 r + 32) for dQ / dbias and the KEYS [32 r, 32 r + 32) for dK / dV, so every
// output is written by exactly one workgroup (no atomics) at the price of forming the overlap of the two score blocks
// twice.  Config 2 (8 scenes x 8 heads x 4 slices) puts one workgroup on each of the 256 CUs, so the latency hiding has to
// come from the workgroup's own waves: 1024 threads (4 waves per SIMD) at d_h 32 -- with 256 threads every dependent LDS
// access and FMA chain was exposed (forward 30k cycles per workgroup: scores 520 / softmax 800 cycles per ROW) -- 512 for
// d_h 16 (a thread needs >= 1 output) and d_h 64 (two 64-float rows in registers).
constexpr int SR = 32;
template <int DH> struct SN { static constexpr int T = DH == 32 ? 1024 : 512; };
// the backward holds three full [L, d_h] operands plus three score tiles (81 KB at config 2) and two K / V rows per thread
// in registers: 512 threads measured best once the dot products used packed FMAs (26 us; 256 threads 28 us; 1024 threads
// 42 us: the 128-register cap forces the operand rows out)
template <int DH> struct SNB { static constexpr int T = 512; };

// dst[row * LS + col] = src[(r0 + row) * Lk + col] (or 0) for `rows` rows of a [*, Lk] fp32 matrix, as batches of
// independent loads: a plain "load, store" loop is compiled to one outstanding load at a time (dependent global round trips)
template <int SNT> PQ_DEV void load_rows(float* dst, const float* src, int r0, int rows, int Lk, int LS, int tid) {
  constexpr int U = 4;
  const int n = rows * Lk;
  if (src && (Lk & 3) == 0 && ((((uintptr_t)src) & 15) == 0)) {
    const int n4 = n / 4;
    const float4* s4 = (const float4*)(src + (long)r0 * Lk);
    for (int base = tid; base < n4; base += SNT * U) {
      float4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = s4[min(base + u * SNT, n4 - 1)];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = base + u * SNT, e = idx * 4;
        if (idx < n4) *(float4*)&dst[(e / Lk) * LS + e % Lk] = v[u];
      }
    }
  } else {
    for (int base = tid; base < n; base += SNT * U) {
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = src ? src[(long)r0 * Lk + min(base + u * SNT, n - 1)] : 0.f;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = base + u * SNT;
        if (e < n) dst[(e / Lk) * LS + e % Lk] = v[u];
      }
    }
  }
}

// rows [r0, r0 + rows) x DH of a strided [*, DH] matrix into a dense LDS tile (zero past Lmax)
template <int DH, int SNT> PQ_DEV void load_mat(float* dst, const float* src, long sl, int r0, int rows, int Lmax, int tid) {
  for (int c = tid; c < rows * (DH / 4); c += SNT) {
    const int r = c / (DH / 4), x = (c % (DH / 4)) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < Lmax) v = *(const float4*)(src + (long)(r0 + r) * sl + x);
    *(float4*)&dst[r * DH + x] = v;
  }
}

template <int DH> PQ_DEV void load_row_regs(float (&r)[DH], const float* p) {
#pragma unroll
  for (int x = 0; x < DH; x += 4) { const float4 t = *(const float4*)(p + x); r[x] = t.x; r[x + 1] = t.y; r[x + 2] = t.z; r[x + 3] = t.w; }
}
// packed fp32 FMAs (v_pk_fma_f32: two lanes of a 64-bit register pair per instruction): 4 independent accumulator pairs,
// half the VALU instructions of the scalar chain
typedef float f32x2 __attribute__((ext_vector_type(2)));
PQ_DEV f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
template <int DH> PQ_DEV float dot_lds(const float* row, const float (&r)[DH]) {   // row: LDS, same address for the whole wave
  f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f}, a2 = {0.f, 0.f}, a3 = {0.f, 0.f};
#pragma unroll
  for (int x = 0; x < DH; x += 8) {
    const float4 t = *(const float4*)&row[x], u = *(const float4*)&row[x + 4];
    a0 = pk_fma((f32x2){t.x, t.y}, (f32x2){r[x], r[x + 1]}, a0);
    a1 = pk_fma((f32x2){t.z, t.w}, (f32x2){r[x + 2], r[x + 3]}, a1);
    a2 = pk_fma((f32x2){u.x, u.y}, (f32x2){r[x + 4], r[x + 5]}, a2);
    a3 = pk_fma((f32x2){u.z, u.w}, (f32x2){r[x + 6], r[x + 7]}, a3);
  }
  const f32x2 s = (a0 + a1) + (a2 + a3);
  return s.x + s.y;
}

template <int DH>
__global__ __launch_bounds__(SN<DH>::T) void attn_small_fwd_kernel(const pq3d_attn_desc d) {
  ATTN_KARG_PIN(d);
  constexpr int SNT = SN<DH>::T;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Lq = d.Lq, Lk = d.Lk, LS = ((Lk + 3) & ~3) + 4;   // float4-readable score rows, bank-shifted
  float* Qs = sm;                    // [SR][DH]   this slice's query rows
  float* Vs = Qs + SR * DH;          // [Lk][DH]
  float* S = Vs + Lk * DH;           // [SR][LS]
  float* Li = S + SR * LS;           // [SR] 1 / rowsum
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y, h = blockIdx.x, r0 = blockIdx.z * SR;
  const int nr = min(SR, Lq - r0);   // rows of this slice
  const float* q = (const float*)d.q + (long)b * d.q_sb + (long)h * d.q_sh;
  const float* k = (const float*)d.k + (long)b * d.k_sb + (long)h * d.k_sh;
  const float* v = (const float*)d.v + (long)b * d.v_sb + (long)h * d.v_sh;
  load_mat<DH, SNT>(Qs, q, d.q_sl, r0, SR, Lq, tid);
  load_mat<DH, SNT>(Vs, v, d.v_sl, 0, Lk, Lk, tid);
  // ---- scores: thread owns key column j (its K row in registers) and every other row of the slice
  const int j = tid & 127, ih = tid >> 7;
  const bool jv = j < Lk;
  float kr[DH];
  load_row_regs<DH>(kr, k + (long)min(j, Lk - 1) * d.k_sl);
  const bool jm = jv ? (d.kpm ? d.kpm[(long)b * Lk + j] != 0 : false) : true;
  const float* bias = d.bias ? d.bias + ((long)b * d.H + h) * Lq * (long)Lk : nullptr;
  // the additive bias rows go into the score buffer first, as one coalesced sweep with every load in flight at once
  load_rows<SNT>(S, bias, r0, nr, Lk, LS, tid);
  __syncthreads();
  for (int i = ih; i < nr; i += SNT / 128) {
    const float s = dot_lds<DH>(&Qs[i * DH], kr);
    if (jv) S[i * LS + j] = jm ? -INFINITY : s * d.scale + S[i * LS + j];
    else if (j < LS) S[i * LS + j] = 0.f;       // padding columns: read (times 0) by the float4 loop below
  }
  __syncthreads();
  // ---- row softmax: one wave per row
  for (int i = wave; i < nr; i += SNT / 64) {
    const float a0 = lane < Lk ? S[i * LS + lane] : -INFINITY, a1 = lane + 64 < Lk ? S[i * LS + lane + 64] : -INFINITY;
    const float m = wave_max(fmaxf(a0, a1));
    const float e0 = lane < Lk ? __expf(a0 - m) : 0.f, e1 = lane + 64 < Lk ? __expf(a1 - m) : 0.f;
    const float l = wave_sum(e0 + e1);
    if (lane < Lk) S[i * LS + lane] = e0;
    if (lane + 64 < Lk) S[i * LS + lane + 64] = e1;
    if (lane == 0) { Li[i] = 1.f / l; d.lse[((long)b * d.H + h) * Lq + r0 + i] = m + logf(l); }
  }
  __syncthreads();
  // ---- O = P V: thread owns channel c and RPT consecutive rows: one V read feeds RPT FMAs, P rows are read as float4
  constexpr int RPT = SR * DH / SNT;     // 1 / 1 / 4
  const int c = tid % DH, i0 = (tid / DH) * RPT;
  f32x2 ac0[RPT], ac1[RPT];   // keys (jj, jj + 1) and (jj + 2, jj + 3): packed FMAs
#pragma unroll
  for (int r = 0; r < RPT; ++r) { ac0[r] = (f32x2){0.f, 0.f}; ac1[r] = (f32x2){0.f, 0.f}; }
  for (int jj = 0; jj < Lk; jj += 4) {
    float vv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) vv[u] = jj + u < Lk ? Vs[(jj + u) * DH + c] : 0.f;
    float4 p[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) p[r] = *(const float4*)&S[min(i0 + r, nr - 1) * LS + jj];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      ac0[r] = pk_fma((f32x2){p[r].x, p[r].y}, (f32x2){vv[0], vv[1]}, ac0[r]);
      ac1[r] = pk_fma((f32x2){p[r].z, p[r].w}, (f32x2){vv[2], vv[3]}, ac1[r]);
    }
  }
  float acc[RPT];
#pragma unroll
  for (int r = 0; r < RPT; ++r) { const f32x2 t = ac0[r] + ac1[r]; acc[r] = t.x + t.y; }
  float* o = (float*)d.o + (long)b * d.o_sb + (long)h * d.o_sh;
#pragma unroll
  for (int r = 0; r < RPT; ++r)
    if (i0 + r < nr) o[(long)(r0 + i0 + r) * d.o_sl + c] = acc[r] * Li[i0 + r];
}

template <int DH>
__global__ __launch_bounds__(SNB<DH>::T) void attn_small_bwd_kernel(const pq3d_attn_desc d) {
  ATTN_KARG_PIN(d);
  ATTN_KARG_PIN_BWD(d);
  constexpr int SNT = SNB<DH>::T;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Lq = d.Lq, Lk = d.Lk, LS = ((Lk + 3) & ~3) + 4;
  constexpr int LC = SR + 4;         // row stride of the key-block tiles
  float* Qs = sm;                    // [Lq][DH]
  float* Gs = Qs + Lq * DH;          // [Lq][DH]   dO
  float* Ks = Gs + Lq * DH;          // [Lk][DH]
  float* dSr = Ks + Lk * DH;         // [SR][LS]   dS of this slice's query rows (all keys)   -> dQ, dbias
  float* Pc = dSr + SR * LS;         // [Lq][LC]   P  of this slice's keys (all query rows)   -> dV
  float* dSc = Pc + Lq * LC;         // [Lq][LC]   dS of this slice's keys                    -> dK
  float* Dl = dSc + Lq * LC;         // [Lq] delta
  float* Ls = Dl + Lq;               // [Lq] lse
  const int tid = threadIdx.x;
  const int b = blockIdx.y, h = blockIdx.x, r0 = blockIdx.z * SR, k0 = blockIdx.z * SR;
  const int nr = max(0, min(SR, Lq - r0)), nk = max(0, min(SR, Lk - k0));
  const float* q = (const float*)d.q + (long)b * d.q_sb + (long)h * d.q_sh;
  const float* k = (const float*)d.k + (long)b * d.k_sb + (long)h * d.k_sh;
  const float* v = (const float*)d.v + (long)b * d.v_sb + (long)h * d.v_sh;
  const float* o = (const float*)d.o + (long)b * d.o_sb + (long)h * d.o_sh;
  const float* g = (const float*)d.dout + (long)b * d.o_sb + (long)h * d.o_sh;
  const long sbase = ((long)b * d.H + h) * Lq;
  load_mat<DH, SNT>(Qs, q, d.q_sl, 0, Lq, Lq, tid);
  load_mat<DH, SNT>(Gs, g, d.o_sl, 0, Lq, Lq, tid);
  load_mat<DH, SNT>(Ks, k, d.k_sl, 0, Lk, Lk, tid);
  for (int i = tid; i < Lq; i += SNT) {   // delta = rowsum(dO * O)
    float s = 0.f;
    for (int x = 0; x < DH; x += 4) {
      const float4 a = *(const float4*)(g + (long)i * d.o_sl + x), t = *(const float4*)(o + (long)i * d.o_sl + x);
      s += a.x * t.x + a.y * t.y + a.z * t.z + a.w * t.w;
    }
    Dl[i] = s;
    Ls[i] = d.lse[sbase + i];
    if (blockIdx.z == 0) d.delta[sbase + i] = s;
  }
  const float* bias = d.bias ? d.bias + ((long)b * d.H + h) * Lq * (long)Lk : nullptr;
  float* dbias = d.dbias ? d.dbias + ((long)b * d.H + h) * Lq * (long)Lk : nullptr;
  if (nr > 0) load_rows<SNT>(dSr, bias, r0, nr, Lk, LS, tid);   // bias rows of the query slice, parked in the dS buffer
  // ---- (a) row block: thread owns key column j (K, V rows in registers) and every other row of the slice
  {
    const int j = tid & 127, ih = tid >> 7;
    const bool jv = j < Lk;
    float kr[DH], vr[DH];
    load_row_regs<DH>(kr, k + (long)min(j, Lk - 1) * d.k_sl);
    load_row_regs<DH>(vr, v + (long)min(j, Lk - 1) * d.v_sl);
    const bool jm = jv ? (d.kpm ? d.kpm[(long)b * Lk + j] != 0 : false) : true;
    __syncthreads();
    for (int i = ih; i < nr; i += SNT / 128) {
      const float s = dot_lds<DH>(&Qs[(r0 + i) * DH], kr), dp = dot_lds<DH>(&Gs[(r0 + i) * DH], vr);
      if (jv) {
        const float p = jm ? 0.f : __expf(s * d.scale + dSr[i * LS + j] - Ls[r0 + i]);
        dSr[i * LS + j] = p * (dp - Dl[r0 + i]);
      } else if (j < LS) {
        dSr[i * LS + j] = 0.f;
      }
    }
  }
  // ---- (b) key block: thread owns key k0 + (tid % 32) and every 8th query row
  if (nk > 0) {
    const int jc = tid & (SR - 1), rg = tid / SR;
    const int key = k0 + jc;
    const bool jv = key < Lk;
    float kr[DH], vr[DH];
    load_row_regs<DH>(kr, k + (long)min(key, Lk - 1) * d.k_sl);
    load_row_regs<DH>(vr, v + (long)min(key, Lk - 1) * d.v_sl);
    const bool jm = jv ? (d.kpm ? d.kpm[(long)b * Lk + key] != 0 : false) : true;
    constexpr int NI = 128 / (SNT / SR);   // rows per thread (Lq <= 128)
    float bv[NI];                          // this thread's bias column entries, all requested before the loop
#pragma unroll
    for (int t = 0; t < NI; ++t) {
      const int i = rg + t * (SNT / SR);
      bv[t] = (bias && jv && i < Lq) ? bias[(long)i * Lk + key] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < NI; ++t) {
      const int i = rg + t * (SNT / SR);
      if (i < Lq) {
        const float s = dot_lds<DH>(&Qs[i * DH], kr), dp = dot_lds<DH>(&Gs[i * DH], vr);
        const float p = jm ? 0.f : __expf(s * d.scale + bv[t] - Ls[i]);
        Pc[i * LC + jc] = p;
        dSc[i * LC + jc] = p * (dp - Dl[i]);
      }
    }
  }
  __syncthreads();
  if (dbias && nr > 0)
    for (int e = tid; e < nr * Lk; e += SNT) dbias[(long)(r0 + e / Lk) * Lk + e % Lk] = dSr[(e / Lk) * LS + e % Lk];
  constexpr int RPT = SR * DH / SNT;     // rows (keys) per thread: 1 / 1 / 4
  const int c = tid % DH, i0 = (tid / DH) * RPT;
  // ---- dQ = scale dS K for the slice's rows: thread owns channel c and RPT consecutive rows
  if (nr > 0) {
    f32x2 ac0[RPT], ac1[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) { ac0[r] = (f32x2){0.f, 0.f}; ac1[r] = (f32x2){0.f, 0.f}; }
    for (int jj = 0; jj < Lk; jj += 4) {
      float kk[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) kk[u] = jj + u < Lk ? Ks[(jj + u) * DH + c] : 0.f;
      float4 t[RPT];
#pragma unroll
      for (int r = 0; r < RPT; ++r) t[r] = *(const float4*)&dSr[min(i0 + r, nr - 1) * LS + jj];
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        ac0[r] = pk_fma((f32x2){t[r].x, t[r].y}, (f32x2){kk[0], kk[1]}, ac0[r]);
        ac1[r] = pk_fma((f32x2){t[r].z, t[r].w}, (f32x2){kk[2], kk[3]}, ac1[r]);
      }
    }
    float acc[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) { const f32x2 t2 = ac0[r] + ac1[r]; acc[r] = t2.x + t2.y; }
    float* dq = (float*)d.dq + (long)b * d.q_sb + (long)h * d.q_sh;
#pragma unroll
    for (int r = 0; r < RPT; ++r)
      if (i0 + r < nr) dq[(long)(r0 + i0 + r) * d.q_sl + c] = acc[r] * d.scale;
  }
  // ---- dK = scale dS^T Q, dV = P^T dO for the slice's keys: thread owns channel c and RPT consecutive keys
  if (nk > 0) {
    f32x2 akv[RPT];   // (dK, dV) accumulator pairs: one packed FMA per (query row, key)
#pragma unroll
    for (int r = 0; r < RPT; ++r) akv[r] = (f32x2){0.f, 0.f};
    for (int i = 0; i < Lq; ++i) {
      const f32x2 qg = {Qs[i * DH + c], Gs[i * DH + c]};
#pragma unroll
      for (int r = 0; r < RPT; ++r) akv[r] = pk_fma((f32x2){dSc[i * LC + i0 + r], Pc[i * LC + i0 + r]}, qg, akv[r]);
    }
    float ak[RPT], av[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) { ak[r] = akv[r].x; av[r] = akv[r].y; }
    float* dk = (float*)d.dk + (long)b * d.k_sb + (long)h * d.k_sh;
    float* dv = (float*)d.dv + (long)b * d.v_sb + (long)h * d.v_sh;
#pragma unroll
    for (int r = 0; r < RPT; ++r)
      if (i0 + r < nk) {
        dk[(long)(k0 + i0 + r) * d.k_sl + c] = ak[r] * d.scale;
        dv[(long)(k0 + i0 + r) * d.v_sl + c] = av[r];
      }
  }
}

size_t small_lds(const pq3d_attn_desc& d, bool bwd) {
  const size_t ls = ((d.Lk + 3) & ~3) + 4, dh = d.dh;
  const size_t fl = bwd ? (2 * d.Lq * dh + d.Lk * dh + SR * ls + 2 * (size_t)d.Lq * (SR + 4) + 2 * d.Lq)
                        : (SR * dh + d.Lk * dh + SR * ls + SR);
  return fl * 4 + 16;
}

bool small_ok(const pq3d_attn_desc& d) {
  return d.ct == PQ3D_F32 && d.dt == PQ3D_F32 && d.Lq >= 1 && d.Lq <= 128 && d.Lk >= 1 && d.Lk <= 128 &&
         (d.dh == 16 || d.dh == 32 || d.dh == 64) && !d.mask && !d.zero_attn && !(d.drop.p > 0.f && d.drop.seed) &&
         d.ksplit <= 1;
}

template <int DH> void launch_small(const pq3d_attn_desc& d, hipStream_t s, bool bwd) {
  const size_t lds = small_lds(d, bwd);
  const int slices = bwd ? (max(d.Lq, d.Lk) + SR - 1) / SR : (d.Lq + SR - 1) / SR;
  if (bwd) {
    auto kern = attn_small_bwd_kernel<DH>;
    static std::atomic<unsigned> done{0};   // per (kernel, device)
    if (pq3d_enable_big_lds(kern, 160 * 1024, done)) { (void)hipGetLastError(); }
    hipLaunchKernelGGL(kern, dim3(d.H, d.B, slices), dim3(SNB<DH>::T), lds, s, d);
  } else {
    auto kern = attn_small_fwd_kernel<DH>;
    static std::atomic<unsigned> done{0};   // per (kernel, device)
    if (pq3d_enable_big_lds(kern, 160 * 1024, done)) { (void)hipGetLastError(); }
    hipLaunchKernelGGL(kern, dim3(d.H, d.B, slices), dim3(SN<DH>::T), lds, s, d);
  }
}

}  // namespace

// fp32 attention with at most 128 queries and keys (the decoder's self-attention): returns false when the call is not of
// that shape (attention.hip's general kernels run instead).
bool pq3d_attn_small_try(const pq3d_attn_desc& d, hipStream_t s, bool bwd) {
  if (!small_ok(d)) return false;
  if (small_lds(d, bwd) > 160 * 1024) return false;   // the head's tiles must fit the LDS of one CU
  if (bwd && !(d.dout && d.dq && d.dk && d.dv && d.delta)) return false;
  if (d.dh == 16) launch_small<16>(d, s, bwd);
  else if (d.dh == 32) launch_small<32>(d, s, bwd);
  else launch_small<64>(d, s, bwd);
  return true;
}
