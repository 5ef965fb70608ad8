// Whole-K GEMM tiles for the small-M launches of the query side (M = B*N_q rows: projections, FFN, heads and their input
// gradients).  Those launches are latency chains, not throughput problems: with the 64x64-tile kernel of gemm.hip an
// 800-row product runs on 52 workgroups, one wave per SIMD, and walks K in four dependent (load -> convert -> LDS -> barrier
// -> MFMA -> barrier) steps -- 13.9k cycles in-kernel for 0.1 GFLOP.  Here a workgroup of 8 waves takes a 32x64 (or 64x64)
// output tile and stages the operands for up to 256 k elements AT ONCE: every global load of the tile is issued in one batch
// (one memory round trip instead of four), converted once (split-bf16 hi/lo or single bf16), and the MFMAs then run
// back to back out of LDS behind a single barrier.  Twice the workgroups (32-row tiles: 100 for [800 x 256]), two waves per
// SIMD, one round trip.  Longer K (FFN second layer, K-concatenated groups, split-K) walks 256-wide chunks with the next
// chunk's loads in flight during the MFMAs of the current one.
//
// Same MFMA sequence per accumulator as gemm_fast_kernel (k ascending in steps of 32; split-bf16 terms in the order
// lo*hi, hi*lo, hi*hi), same epilogue code (gemm_common.h) -> bit-identical results; tests/test_gpu_ops.py compares them.
// Layouts: C = A.B^T with B [N][K] row-major ("NN" in the kernel tables: forward projections) and C = A.B with B [K][N]
// (transB: input gradients; single bf16 only, fragments through the transposing LDS read).
#include <atomic>
#include <cstdlib>

#include "gemm_common.h"

namespace {

constexpr int WT = 512;        // threads per workgroup (8 waves)
constexpr int TN = 64;         // output columns per workgroup
constexpr int LDKN = TN + 8;   // row of the [k][n] tile of a transposed B operand

// KC = k elements staged per chunk (256: one round trip for d = 256; 128: half the LDS -> more workgroups per CU)
template <int TM, int KC> struct WkShape {
  static constexpr int LDR = KC + 8;                    // bf16 per LDS row of a row-major operand tile (132 / 68 dwords = 4 mod 64 banks)
  static constexpr int CPR = KC / 8;                    // 8-element chunks per operand row
  static constexpr int WAVES_M = TM / 16;               // 2 (TM = 32) or 4 (TM = 64): one 16-row MFMA block per wave
  static constexpr int WAVES_N = 8 / WAVES_M;           // 4 or 2
  static constexpr int NJ = TN / (16 * WAVES_N);        // 16-column MFMA blocks per wave: 1 or 2
  static constexpr int NA = TM * (KC / 8) / WT;         // 8-element A chunks per thread: 2 or 4
  static constexpr int NB = TN * (KC / 8) / WT;         // 8-element B chunks per thread: 4
  static constexpr int NV = TM * TN / WT;               // epilogue columns per thread: 4 or 8
  static constexpr int CLD = TN + 4;                    // padded fp32 row of the transposed C tile
};

template <bool X3, bool TRB, int TM, int KC> constexpr size_t wk_lds_bytes() {
  return (size_t)(X3 ? 2 : 1) * ((size_t)TM * (KC + 8) + (TRB ? (size_t)KC * LDKN : (size_t)TN * (KC + 8))) * sizeof(bf16_t);
}

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

template <bool X3, typename TA, typename TB, bool TRB, bool HA2, int TM, int KC>
__global__ __launch_bounds__(WT) void gemm_wk_kernel(const pq3d_kdesc d) {
  typedef WkShape<TM, KC> S;
  constexpr int LDR = S::LDR, CPR = S::CPR;
  static_assert(!X3 || (!TRB && sizeof(TA) == 4 && sizeof(TB) == 4), "split-bf16: row-major fp32 operands");
  static_assert(!HA2 || sizeof(TA) == 4, "addend needs an fp32 primary");
  extern __shared__ __attribute__((aligned(16))) unsigned char wk_smem[];
  constexpr int ASZ = TM * LDR, BSZ = TRB ? KC * LDKN : TN * LDR;
  bf16_t* const Ah = (bf16_t*)wk_smem;
  bf16_t* const Bh = Ah + ASZ;
  bf16_t* const Al = Bh + BSZ;   // X3 only
  bf16_t* const Bl = Al + ASZ;
  static_assert(wk_lds_bytes<X3, TRB, TM, KC>() >= sizeof(float) * TM * S::CLD, "C tile must fit in the staging LDS");

  // kernel-argument prefetch (see gemm_fast_kernel): every scalar the kernel uses in one batch of scalar loads
  GPtrs gp;
  const int gspec = min((int)blockIdx.z, PQ3D_MAX_GROUPS - 1);   // z is never remapped (tile_index_plane)
  gp.load(d, gspec);
  asm volatile("" ::"s"(d.M), "s"(d.N), "s"(d.K), "s"(d.splitk), "s"(d.kconcat), "s"(d.lda), "s"(d.ldb), "s"(d.ldc),
               "s"(d.strideC), "s"(d.alpha), "s"(d.act), "s"(d.act_grad), "s"(d.dtC), "s"(d.dtC2), "s"(d.dtAux), "s"(d.dtBias),
               "s"(d.row_fill), "s"(d.row_scale), "s"(d.row_fill_flag), "s"(d.mask_out), "s"(gp.A), "s"(gp.A2), "s"(gp.B), "s"(d.xcd_order));
  asm volatile("" ::"s"(gp.bias), "s"(gp.aux), "s"(gp.C), "s"(gp.C2), "s"(gp.row_mask));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int wm = (wave / S::WAVES_N) * 16, wn = (wave % S::WAVES_N) * (16 * S::NJ);
  const int ng = d.kconcat > 0 ? d.kconcat : 1;
  int zz = blockIdx.z, split = 0;
  if (d.splitk > 1) { split = zz % d.splitk; zz /= d.splitk; }
  const int g0 = zz * ng;
  const TileIdx ti = tile_index_plane(d.xcd_order);   // the tiles one XCD receives are neighbours in its plane (common.h)
  const int m0 = ti.x * TM, n0 = ti.y * TN;
  const int nck = (d.K + KC - 1) / KC;
  int c0 = 0, c1 = nck;
  if (d.splitk > 1) {
    const int per = (nck + d.splitk - 1) / d.splitk;
    c0 = split * per;
    c1 = min(nck, c0 + per);
    if (c0 >= c1) return;
  }
  if (g0 != gspec) gp.load(d, g0);
  const int ncl = c1 - c0, nit = ncl * ng;

  f32x4 acc[S::NJ];
#pragma unroll
  for (int j = 0; j < S::NJ; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- one register set: the loads of chunk t+1 are issued right after chunk t has been written to LDS
  Raw<TA, 8> ra[S::NA];
  Raw<float, 8> ra2[HA2 ? S::NA : 1];
  Raw<TB, 8> rb[S::NB];
  bool oka[S::NA], okb[S::NB];
  float s2 = 0.f;
  const bool ua = ((d.lda | d.K) & 7) != 0;   // unaligned A rows (host: fp32 A, transposed B only)
  int gi = g0, ci = c0;   // next (group, chunk) to load
  const void *pA = gp.A, *pA2 = gp.A2, *pB = gp.B;
  auto issue = [&]() {
    const int k0 = ci * KC;
    const TA* Ab = (const TA*)pA;
    const TB* Bb = (const TB*)pB;
    s2 = pA2 ? 1.f : 0.f;
#pragma unroll
    for (int i = 0; i < S::NA; ++i) {
      const int c = tid + i * WT, row = c / CPR, k = k0 + (c % CPR) * 8;
      oka[i] = k < d.K;
      const long off = (long)min(m0 + row, d.M - 1) * d.lda + (oka[i] ? k : 0);
      if constexpr (sizeof(TA) == 4 && !HA2 && TRB) {
        if (ua) {   // uniform: rows of an unaligned length (class logits: 201 columns) -- element loads, zeros past K
          float t[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) t[j] = (k + j < d.K) ? ((const float*)Ab)[off + j] : 0.f;
          ra[i].a = make_float4(t[0], t[1], t[2], t[3]);
          ra[i].b = make_float4(t[4], t[5], t[6], t[7]);
          continue;
        }
      }
      ra[i].load(Ab + off);
      if constexpr (HA2) ra2[i].load(pA2 ? (const float*)pA2 + off : (const float*)pA + off);   // no addend: finite filler, scaled by 0
    }
#pragma unroll
    for (int i = 0; i < S::NB; ++i) {
      const int c = tid + i * WT;
      long off;
      if constexpr (!TRB) {
        const int row = c / CPR, k = k0 + (c % CPR) * 8;
        okb[i] = k < d.K;
        off = (long)min(n0 + row, d.N - 1) * d.ldb + (okb[i] ? k : 0);
      } else {
        const int k = k0 + (c >> 3);
        okb[i] = k < d.K;
        off = (long)(okb[i] ? k : 0) * d.ldb + min(n0 + (c & 7) * 8, d.N - 8);
      }
      rb[i].load(Bb + off);
    }
    if (++ci == c1) {   // next group of a K-concatenated product (pointer loads are uniform scalar loads)
      ci = c0; ++gi;
      if (gi < g0 + ng) { const pq3d_kgroup& q = d.gp[gi]; pA = q.A; pA2 = q.A2; pB = q.B; }
    }
  };
  auto put = [&]() {
#pragma unroll
    for (int i = 0; i < S::NA; ++i) {
      const int c = tid + i * WT, o = (c / CPR) * LDR + (c % CPR) * 8;
      if constexpr (!X3 && !HA2 && sizeof(TA) == 2) {
        u32x4 p = __builtin_bit_cast(u32x4, ra[i]);
        if (!oka[i]) p = (u32x4){0, 0, 0, 0};
        *(u32x4*)&Ah[o] = p;
      } else {
        float v[8];
        ra[i].to_float(v);
        if constexpr (HA2) {
          float w[8];
          ra2[i].to_float(w);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += s2 * w[j];
        }
        if (!oka[i]) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = 0.f;
        }
        if constexpr (X3) {
          u32x4 hi, lo;
          split_hi_lo(v, hi, lo);
          *(u32x4*)&Ah[o] = hi;
          *(u32x4*)&Al[o] = lo;
        } else {
          *(u32x4*)&Ah[o] = pack_frag<bf16_t>(v);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < S::NB; ++i) {
      const int c = tid + i * WT;
      const int o = TRB ? (c >> 3) * LDKN + (c & 7) * 8 : (c / CPR) * LDR + (c % CPR) * 8;
      if constexpr (!X3 && sizeof(TB) == 2) {
        u32x4 p = __builtin_bit_cast(u32x4, rb[i]);
        if (!okb[i]) p = (u32x4){0, 0, 0, 0};
        *(u32x4*)&Bh[o] = p;
      } else {
        float v[8];
        rb[i].to_float(v);
        if (!okb[i]) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = 0.f;
        }
        if constexpr (X3) {
          u32x4 hi, lo;
          split_hi_lo(v, hi, lo);
          *(u32x4*)&Bh[o] = hi;
          *(u32x4*)&Bl[o] = lo;
        } else {
          *(u32x4*)&Bh[o] = pack_frag<bf16_t>(v);
        }
      }
    }
  };

  issue();
  // the bias row in accumulator layout, requested right behind the first operand loads (no dependent round trip later)
  const bool bias_early = gp.bias != nullptr && d.dtBias == PQ3D_F32 && d.alpha == 1.f && d.splitk <= 1;
  float bcol[S::NJ];
#pragma unroll
  for (int j = 0; j < S::NJ; ++j) bcol[j] = bias_early ? ((const float*)gp.bias)[min(n0 + wn + j * 16 + li, d.N - 1)] : 0.f;

  int kc_cur = c0;   // chunk index of the tile being multiplied
  for (int it = 0; it < nit; ++it) {
    if (it > 0) __syncthreads();   // the previous chunk's fragment reads are done
    put();
    __syncthreads();
    const int kspan = min(KC, d.K - kc_cur * KC);
    if (++kc_cur == c1) kc_cur = c0;
    if (it + 1 < nit) issue();
    const int nks = (kspan + 31) >> 5;
#pragma unroll
    for (int ks = 0; ks < KC / 32; ++ks) {
      if (ks < nks) {   // uniform
        const int oa = (wm + li) * LDR + ks * 32 + lg * 8;
        const u32x4 ah = *(const u32x4*)&Ah[oa];
        u32x4 al;
        if constexpr (X3) al = *(const u32x4*)&Al[oa];
#pragma unroll
        for (int j = 0; j < S::NJ; ++j) {
          u32x4 bh, bl;
          if constexpr (TRB) {
            bh = km_frag(Bh, LDKN, wn + j * 16, ks, li, lg);
          } else {
            const int ob = (wn + j * 16 + li) * LDR + ks * 32 + lg * 8;
            bh = *(const u32x4*)&Bh[ob];
            if constexpr (X3) bl = *(const u32x4*)&Bl[ob];
          }
          if constexpr (X3) {
            Mma<bf16_t>::mma(acc[j], al, bh);
            Mma<bf16_t>::mma(acc[j], ah, bl);
          }
          Mma<bf16_t>::mma(acc[j], ah, bh);
        }
      }
    }
  }

  if (d.splitk > 1) {   // atomics straight from the C-layout registers (16 consecutive addresses per lane group)
    float* C = (float*)gp.C;
#pragma unroll
    for (int j = 0; j < S::NJ; ++j) {
      const int col = n0 + wn + j * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm + lg * 4 + r;
        if (row < d.M && col < d.N) unsafeAtomicAdd(C + (long)row * d.ldc + col, acc[j][r] * d.alpha);
      }
    }
    return;
  }
  __syncthreads();   // operand tiles are dead: the transposed C tile overlays them
  float* const Ct = (float*)wk_smem;
#pragma unroll
  for (int j = 0; j < S::NJ; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) Ct[(wm + lg * 4 + r) * S::CLD + wn + j * 16 + li] = (acc[j][r] + bcol[j]) * d.alpha;
  __syncthreads();
  constexpr int TPR = TN / S::NV;   // threads per output row
  const int lrow = tid / TPR, lcol = (tid % TPR) * S::NV;
  const int row = m0 + lrow, col = n0 + lcol;
  if (row >= d.M || col >= d.N) return;
  float v[S::NV];
#pragma unroll
  for (int j = 0; j < S::NV; j += 4) { const float4 t = *(const float4*)&Ct[lrow * S::CLD + lcol + j]; v[j] = t.x; v[j + 1] = t.y; v[j + 2] = t.z; v[j + 3] = t.w; }
  epi_row<S::NV>(d, gp, v, g0, 0, row, col, bias_early);
}

bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

std::atomic<int> g_wk_enable{1};   // the option word of pq3d_gemm_set_wk (bit 0 = on)
int g_wk_max_m = 2048;

template <bool X3, typename TA, typename TB, bool TRB, bool HA2, int TM, int KC>
int wk_launch(const pq3d_gemm_desc& d, const pq3d_kdesc& kd, hipStream_t s) {
  auto kern = gemm_wk_kernel<X3, TA, TB, TRB, HA2, TM, KC>;
  constexpr size_t lds = wk_lds_bytes<X3, TRB, TM, KC>();
  static std::atomic<unsigned> attr_done{0};   // per (kernel instantiation, device)
  if (int e = pq3d_enable_big_lds(kern, (int)lds, attr_done)) return e;
  const int kc = d.kconcat > 0 ? d.kconcat : 1;
  const dim3 grid((d.M + TM - 1) / TM, (d.N + TN - 1) / TN, (d.groups / kc) * (d.splitk > 1 ? d.splitk : 1));
  if (g_wk_enable.load() & (1 << 9)) {   // A/B switch of the probes: hardware tile order
    hipLaunchKernelGGL(kern, grid, dim3(WT), lds, s, kd);
    return 0;
  }
  pq3d_kdesc k2 = kd;
  const long kl = (long)d.K * kc / (d.splitk > 1 ? d.splitk : 1);   // reduction length one workgroup walks
  k2.xcd_order = plane_xcd_order((int)grid.x, (int)grid.y, (long)TM * kl * (long)(sizeof(TA) + (HA2 ? 4 : 0)), (long)TN * kl * (long)sizeof(TB));
  hipLaunchKernelGGL(kern, grid, dim3(WT), lds, s, k2);
  return 0;
}

// Tile plan.  What matters for these launches is that ALL workgroups are resident at once (one round): a second round
// costs a whole in-kernel latency chain.  Residency per CU: 160 KB of LDS / the kernel's footprint, at most two 8-wave
// workgroups.  Among the plans that fit in one round the cheapest is the one with the fewest staged chunks per workgroup
// (K = 256 in one chunk), then 32-row tiles (no padded rows at M = 800, half the conversion work per thread).
struct WkPlan { int tm, kc; };
size_t wk_lds(bool x3, bool trb, int tm, int kc) {
  return (size_t)(x3 ? 2 : 1) * ((size_t)tm * (kc + 8) + (trb ? (size_t)kc * LDKN : (size_t)TN * (kc + 8))) * 2;
}
bool wk_plan(const pq3d_gemm_desc& d, bool x3, int opt, WkPlan* out) {
  const int force_tm = (opt >> 4) & 3, force_kc = (opt >> 6) & 3;
  const bool multi_round = (opt >> 8) & 1;
  const int kcn = d.kconcat > 0 ? d.kconcat : 1, sk = d.splitk > 1 ? d.splitk : 1;
  float best = 1e30f;
  bool found = false;
  for (int tm = 32; tm <= 64; tm *= 2)
    for (int kc = 256; kc >= 128; kc /= 2) {
      if (force_tm && tm != (force_tm == 1 ? 32 : 64)) continue;
      if (force_kc && kc != (force_kc == 1 ? 128 : 256)) continue;
      const long wgs = (long)((d.M + tm - 1) / tm) * ((d.N + TN - 1) / TN) * (d.groups / kcn) * sk;
      const size_t lds = wk_lds(x3, d.transB != 0, tm, kc);
      const long per_cu = lds > 80 * 1024 ? 1 : 2;
      const long rounds = (wgs + 256 * per_cu - 1) / (256 * per_cu);
      if (rounds > 1 && !multi_round) continue;
      const int nck = (d.K + kc - 1) / kc;
      const int chunks = kcn * ((nck + sk - 1) / sk);
      const float cost = (float)rounds * (1.5f + chunks + (tm == 64 ? 0.4f : 0.f)) + (per_cu == 2 && wgs > 256 ? 0.3f : 0.f);
      if (cost < best) { best = cost; out->tm = tm; out->kc = kc; found = true; }
    }
  return found;
}

template <bool X3, typename TA, typename TB, bool TRB, bool HA2>
int wk_launch_plan(const pq3d_gemm_desc& d, const pq3d_kdesc& kd, hipStream_t s, WkPlan p) {
  if (p.tm == 32) {
    if (p.kc == 256) return wk_launch<X3, TA, TB, TRB, HA2, 32, 256>(d, kd, s);
    return wk_launch<X3, TA, TB, TRB, HA2, 32, 128>(d, kd, s);
  }
  if (p.kc == 256) return wk_launch<X3, TA, TB, TRB, HA2, 64, 256>(d, kd, s);
  return wk_launch<X3, TA, TB, TRB, HA2, 64, 128>(d, kd, s);
}

}  // namespace

extern "C" int pq3d_gemm_set_wk(int options, int max_m) {
  g_wk_enable.store(options < 0 ? 0 : options);
  if (max_m > 0) g_wk_max_m = max_m;
  return 0;
}

// Returns true when the whole-K kernel took the launch (*err = 0, or a hipError_t with the error text set), false when the
// call is outside its domain.
bool pq3d_gemm_wk_try(const pq3d_gemm_desc& d, const pq3d_kdesc& kd, hipStream_t s, int* err) {
  *err = 0;
  const int en = g_wk_enable.load();
  if (!(en & 1)) return false;
  // A rows of an unaligned length (K % 8 or lda % 8: the class head's 201 logits as the reduction of its input gradient):
  // taken with element loads when A is fp32 without an addend and B is the transposed (k-major, aligned) operand
  const bool ua = ((d.K | d.lda) & 7) != 0;
  if (d.transA || d.batch != 1 || d.M > g_wk_max_m || d.M < 1 || d.K < 8) return false;
  if (ua && !(d.transB && d.dtA == PQ3D_F32 && d.ct == PQ3D_BF16 && d.splitk <= 1 && d.kconcat <= 1)) return false;
  if (d.ct != PQ3D_BF16 && d.ct != PQ3D_BF16X3) return false;
  const bool x3 = d.ct == PQ3D_BF16X3;
  if (x3 && (d.transB || d.dtA != PQ3D_F32 || d.dtB != PQ3D_F32 || d.splitk > 1)) return false;
  if (d.dtB != PQ3D_F32) return false;   // weights are fp32 parameters on every small-M product of the path
  if (d.transB && (d.N % 8 || d.N < 8)) return false;
  if ((!ua && d.lda % 8) || d.ldb % 8) return false;
  if (d.splitk > 1 && (d.dtC != PQ3D_F32 || d.kconcat > 1)) return false;
  if (d.splitk > 1) {   // the split-K branch adds alpha * acc with atomics and has no epilogue: refuse anything that needs one
    if (d.act || d.act_grad || d.row_scale || d.row_fill_flag || d.mask_out || (d.drop.p > 0.f && d.drop.seed)) return false;
    for (int g = 0; g < d.groups; ++g)
      if (d.bias[g] || d.aux[g] || d.C2[g] || d.row_mask[g]) return false;
  }
  bool a2 = false;
  for (int g = 0; g < d.groups; ++g) {
    if ((!ua && !aligned16(d.A[g])) || (((uintptr_t)d.A[g]) & 3) || !aligned16(d.B[g]) || d.B2[g] || d.colsum[g]) return false;
    if (ua && d.A2[g]) return false;
    if (d.A2[g]) { a2 = true; if (!aligned16(d.A2[g]) || d.dtA2 != PQ3D_F32 || d.dtA != PQ3D_F32) return false; }
  }
  if (a2 && d.transB) return false;
  WkPlan p;
  if (!wk_plan(d, x3, en, &p)) return false;   // more than one round of workgroups: the 4-wave pipeline kernel is better there
  int e;
  if (x3) {
    e = a2 ? wk_launch_plan<true, float, float, false, true>(d, kd, s, p) : wk_launch_plan<true, float, float, false, false>(d, kd, s, p);
  } else if (!d.transB) {
    if (d.dtA == PQ3D_F32) e = a2 ? wk_launch_plan<false, float, float, false, true>(d, kd, s, p) : wk_launch_plan<false, float, float, false, false>(d, kd, s, p);
    else e = wk_launch_plan<false, bf16_t, float, false, false>(d, kd, s, p);
  } else {
    if (d.dtA == PQ3D_F32) e = wk_launch_plan<false, float, float, true, false>(d, kd, s, p);
    else e = wk_launch_plan<false, bf16_t, float, true, false>(d, kd, s, p);
  }
  if (e) { pq3d_set_error(hipGetErrorString((hipError_t)e)); *err = e; }
  return true;
}
