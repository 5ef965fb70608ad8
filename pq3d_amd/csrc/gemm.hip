// Grouped / batched MFMA GEMM with fused prologue (A + A2, B + B2, dtype conversion) and epilogue
// (bias, activation, activation-gradient, row masks, mask-head fill/threshold, split-K atomics).
// One kernel family serves every nn.Linear forward/backward on the path and the mask-head einsum.
//
// Tiling: 64x64 output tile per 256-thread workgroup (4 waves as 2x2, each wave a 32x32 sub-tile = 2x2 MFMA
// 16x16 tiles), K-step of 128 bytes of compute type per LDS row (64 bf16 / 32 f32).  LDS rows padded by 16 B.
// Staging is register-prefetched: the raw 16-byte global loads of tile t+1 are issued (branch-free, so they all
// go out back to back) before the MFMAs of tile t and are converted/added/packed only when they are written to
// LDS -- the HBM round trip hides under the MFMAs instead of being paid once per load.
//   FAST path  : operands 16-byte aligned, leading dims / K (or M,N for transposed operands) multiples of the
//                chunk -> unconditional vector loads from clamped addresses + select.
//   generic    : guarded scalar loads (K = 3 or 5 projections, odd shapes).
#include "common.h"

namespace {

constexpr int BM = 64, BN = 64, NT = 256;

template <typename CT> struct Tile {
  static constexpr int EPL = Mma<CT>::EPL;
  static constexpr int KSTEP = Mma<CT>::KSTEP;
  static constexpr int BKE = 128 / (int)sizeof(CT);         // k elements per tile
  static constexpr int LDK = BKE + 16 / (int)sizeof(CT);    // padded LDS row (elements)
  static constexpr int CPR = BKE / EPL;                     // 16-byte chunks per row (= 8)
};

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

// ---- FAST stager: branch-free raw loads, conversion at store time ------------------------------------------
template <typename CT, typename TS, typename TS2, bool TR>
struct FastStage {
  typedef Tile<CT> T;
  Raw<TS, T::EPL> r[2];
  Raw<TS2, T::EPL> r2[2];
  bool kvalid[2];

  PQ_DEV void load(const void* base, const void* base2, long off, long ld, int r0, int R, int k0, int K, int tid) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int c = tid + it * NT;
      long idx;
      if (!TR) {
        const int row = c / T::CPR, kc = c % T::CPR;
        const int gr = min(r0 + row, R - 1), gk = k0 + kc * T::EPL;
        kvalid[it] = gk < K;
        idx = off + (long)gr * ld + (kvalid[it] ? gk : 0);
      } else {
        constexpr int RC = BM / T::EPL;
        const int kk = c / RC, rc = c % RC;
        const int gk = k0 + kk, gr = min(r0 + rc * T::EPL, R - T::EPL);
        kvalid[it] = gk < K;
        idx = off + (long)(kvalid[it] ? gk : 0) * ld + gr;
      }
      r[it].load((const TS*)base + idx);
      if (base2) r2[it].load((const TS2*)base2 + idx);
    }
  }
  PQ_DEV void store(CT* lds, bool has2, int tid) const {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int c = tid + it * NT;
      float v[T::EPL];
      r[it].to_float(v);
      if (has2) {
        float w[T::EPL];
        r2[it].to_float(w);
#pragma unroll
        for (int j = 0; j < T::EPL; ++j) v[j] += w[j];
      }
      if (!kvalid[it]) {
#pragma unroll
        for (int j = 0; j < T::EPL; ++j) v[j] = 0.f;
      }
      if (!TR) {
        const int row = c / T::CPR, kc = c % T::CPR;
        *(u32x4*)&lds[row * T::LDK + kc * T::EPL] = pack_frag<CT>(v);
      } else {
        constexpr int RC = BM / T::EPL;
        const int kk = c / RC, rc = c % RC;
#pragma unroll
        for (int j = 0; j < T::EPL; ++j) lds[(rc * T::EPL + j) * T::LDK + kk] = Cvt<CT>::from(v[j]);
      }
    }
  }
};

// ---- generic stager: guarded scalar loads, runtime dtypes --------------------------------------------------
template <typename CT, bool TR>
struct SlowStage {
  typedef Tile<CT> T;
  u32x4 reg[2];
  PQ_DEV void load(const void* base, const void* base2, int dt, int dt2, long off, long ld, int r0, int R, int k0,
                   int K, int tid) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int c = tid + it * NT;
      float v[T::EPL];
#pragma unroll
      for (int j = 0; j < T::EPL; ++j) v[j] = 0.f;
      long idx = 0;
      int valid = 0;
      if (!TR) {
        const int row = c / T::CPR, kc = c % T::CPR;
        const int gr = r0 + row, gk = k0 + kc * T::EPL;
        if (gr < R && gk < K) { valid = min(T::EPL, K - gk); idx = off + (long)gr * ld + gk; }
      } else {
        constexpr int RC = BM / T::EPL;
        const int kk = c / RC, rc = c % RC;
        const int gk = k0 + kk, gr = r0 + rc * T::EPL;
        if (gk < K && gr < R) { valid = min(T::EPL, R - gr); idx = off + (long)gk * ld + gr; }
      }
#pragma unroll
      for (int j = 0; j < T::EPL; ++j) {
        if (j < valid) {
          v[j] = load_elem(base, dt, idx + j);
          if (base2) v[j] += load_elem(base2, dt2, idx + j);
        }
      }
      reg[it] = pack_frag<CT>(v);
    }
  }
  PQ_DEV void store(CT* lds, int tid) const {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int c = tid + it * NT;
      if (!TR) {
        const int row = c / T::CPR, kc = c % T::CPR;
        *(u32x4*)&lds[row * T::LDK + kc * T::EPL] = reg[it];
      } else {
        constexpr int RC = BM / T::EPL;
        const int kk = c / RC, rc = c % RC;
#pragma unroll
        for (int j = 0; j < T::EPL; ++j) {
          if constexpr (sizeof(CT) == 2)
            lds[(rc * T::EPL + j) * T::LDK + kk] = (CT)((reg[it][j >> 1] >> (16 * (j & 1))) & 0xffffu);
          else
            lds[(rc * T::EPL + j) * T::LDK + kk] = __uint_as_float(reg[it][j]);
        }
      }
    }
  }
};

template <typename CT>
PQ_DEV void mma_tile(f32x4 (&acc)[2][2], const CT* As, const CT* Bs, int wm, int wn, int li, int lg) {
  typedef Tile<CT> T;
#pragma unroll
  for (int ks = 0; ks < T::BKE / T::KSTEP; ++ks) {
    u32x4 fa[2], fb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) fa[i] = *(const u32x4*)&As[(wm + i * 16 + li) * T::LDK + ks * T::KSTEP + lg * T::EPL];
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[j] = *(const u32x4*)&Bs[(wn + j * 16 + li) * T::LDK + ks * T::KSTEP + lg * T::EPL];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) Mma<CT>::mma(acc[i][j], fa[i], fb[j]);
  }
}

// Epilogue.  All global reads it needs (bias, row flags, activation-gradient operand) are gathered into registers
// FIRST, in straight-line code, so they are issued together and waited for once -- not one round trip per element.
PQ_DEV void epilogue(const pq3d_gemm_desc& d, const f32x4 (&acc)[2][2], int g, int z, int m0, int n0, int wm, int wn,
                     int li, int lg) {
  void* C = d.C[g];
  const long offC = (long)z * d.strideC;
  if (d.splitk > 1) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn + j * 16 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + wm + i * 16 + lg * 4 + r;
          if (row < d.M && col < d.N) unsafeAtomicAdd((float*)C + offC + (long)row * d.ldc + col, acc[i][j][r] * d.alpha);
        }
      }
    return;
  }
  void* C2 = d.C2[g];
  const void* aux = d.aux[g];
  const void* bias = d.bias[g];
  const uint8_t* rmask = d.row_mask[g];
  float bv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = min(n0 + wn + j * 16 + li, d.N - 1);
    bv[j] = bias ? load_elem(bias, d.dtBias, col) : 0.f;
  }
  float rs[2][4];      // per-row multiplier (row_mask * row_scale)
  bool rfill[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long ri = (long)z * d.M + min(m0 + wm + i * 16 + lg * 4 + r, d.M - 1);
      float m = 1.f;
      if (rmask) m = rmask[ri] ? 1.f : 0.f;
      if (d.row_scale) m *= d.row_scale[ri];
      rs[i][r] = m;
      rfill[i][r] = d.row_fill_flag ? d.row_fill_flag[ri] != 0 : false;
    }
  float av[2][2][4];
  if (d.act_grad) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = min(m0 + wm + i * 16 + lg * 4 + r, d.M - 1), col = min(n0 + wn + j * 16 + li, d.N - 1);
          av[i][j][r] = load_elem(aux, d.dtAux, offC + (long)row * d.ldc + col);
        }
  }
  const bool row_ops = rmask || d.row_scale;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn + j * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm + i * 16 + lg * 4 + r;
        if (row >= d.M || col >= d.N) continue;
        float v = acc[i][j][r] * d.alpha + bv[j];
        const long ci = offC + (long)row * d.ldc + col;
        if (C2) store_elem(C2, d.dtC2, ci, v);
        if (d.act == PQ3D_ACT_RELU) v = fmaxf(v, 0.f);
        else if (d.act == PQ3D_ACT_GELU) v = gelu_f(v);
        if (d.act_grad == PQ3D_ACT_RELU) v = av[i][j][r] > 0.f ? v : 0.f;
        else if (d.act_grad == PQ3D_ACT_GELU) v *= gelu_grad_f(av[i][j][r]);
        else if (d.act_grad == PQ3D_ACT_ADD) v += av[i][j][r];
        if (row_ops) v *= rs[i][r];
        if (rfill[i][r]) v = d.row_fill;
        store_elem(C, d.dtC, ci, v);
        if (d.mask_out) d.mask_out[((long)z * d.N + col) * d.M + row] = (1.f / (1.f + __expf(-v)) < 0.5f) ? 1 : 0;
      }
    }
  }
}

struct BlockCoords {
  int g, z, m0, n0, kt0, kt1, ng;
  bool active;
};
template <typename CT> PQ_DEV BlockCoords block_coords(const pq3d_gemm_desc& d) {
  typedef Tile<CT> T;
  BlockCoords b;
  b.ng = d.kconcat > 0 ? d.kconcat : 1;           // groups walked inside the K loop
  b.g = (blockIdx.z / d.batch) * b.ng;            // first group of this output
  b.z = blockIdx.z % d.batch;
  const int tiles_m = (d.M + BM - 1) / BM;
  b.m0 = (blockIdx.x % tiles_m) * BM;
  b.n0 = (blockIdx.x / tiles_m) * BN;
  const int nkt = (d.K + T::BKE - 1) / T::BKE;
  b.kt0 = 0; b.kt1 = nkt; b.active = true;
  if (d.splitk > 1) {
    const int per = (nkt + d.splitk - 1) / d.splitk;
    b.kt0 = blockIdx.y * per;
    b.kt1 = min(nkt, b.kt0 + per);
    b.active = b.kt0 < b.kt1;
  }
  return b;
}

template <typename CT, typename TA, typename TA2, typename TB, typename TB2, bool TRA, bool TRB>
__global__ __launch_bounds__(NT) void gemm_fast_kernel(const pq3d_gemm_desc d) {
  typedef Tile<CT> T;
  __shared__ __attribute__((aligned(16))) CT As[BM * T::LDK];
  __shared__ __attribute__((aligned(16))) CT Bs[BN * T::LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const BlockCoords b = block_coords<CT>(d);
  if (!b.active) return;
  const long offA = (long)b.z * d.strideA, offB = (long)b.z * d.strideB;
  const int nk = b.kt1 - b.kt0, nit = nk * b.ng;

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  FastStage<CT, TA, TA2, TRA> sa;
  FastStage<CT, TB, TB2, TRB> sb;
  bool hasA2 = false, hasB2 = false;
  auto load_it = [&](int it) {
    const int gg = b.g + it / nk;
    const int kt = b.kt0 + it % nk;
    hasA2 = d.A2[gg] != nullptr;
    hasB2 = d.B2[gg] != nullptr;
    sa.load(d.A[gg], d.A2[gg], offA, d.lda, b.m0, d.M, kt * T::BKE, d.K, tid);
    sb.load(d.B[gg], d.B2[gg], offB, d.ldb, b.n0, d.N, kt * T::BKE, d.K, tid);
  };
  load_it(0);
  for (int it = 0; it < nit; ++it) {
    sa.store(As, hasA2, tid);
    sb.store(Bs, hasB2, tid);
    __syncthreads();
    if (it + 1 < nit) load_it(it + 1);
    mma_tile<CT>(acc, As, Bs, wm, wn, li, lg);
    __syncthreads();
  }
  epilogue(d, acc, b.g, b.z, b.m0, b.n0, wm, wn, li, lg);
}

template <typename CT, bool TRA, bool TRB>
__global__ __launch_bounds__(NT) void gemm_slow_kernel(const pq3d_gemm_desc d) {
  typedef Tile<CT> T;
  __shared__ __attribute__((aligned(16))) CT As[BM * T::LDK];
  __shared__ __attribute__((aligned(16))) CT Bs[BN * T::LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const BlockCoords b = block_coords<CT>(d);
  if (!b.active) return;
  const long offA = (long)b.z * d.strideA, offB = (long)b.z * d.strideB;
  const int nk = b.kt1 - b.kt0, nit = nk * b.ng;

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  SlowStage<CT, TRA> sa;
  SlowStage<CT, TRB> sb;
  auto load_it = [&](int it) {
    const int gg = b.g + it / nk;
    const int kt = b.kt0 + it % nk;
    sa.load(d.A[gg], d.A2[gg], d.dtA, d.dtA2, offA, d.lda, b.m0, d.M, kt * T::BKE, d.K, tid);
    sb.load(d.B[gg], d.B2[gg], d.dtB, d.dtB2, offB, d.ldb, b.n0, d.N, kt * T::BKE, d.K, tid);
  };
  load_it(0);
  for (int it = 0; it < nit; ++it) {
    sa.store(As, tid);
    sb.store(Bs, tid);
    __syncthreads();
    if (it + 1 < nit) load_it(it + 1);
    mma_tile<CT>(acc, As, Bs, wm, wn, li, lg);
    __syncthreads();
  }
  epilogue(d, acc, b.g, b.z, b.m0, b.n0, wm, wn, li, lg);
}

bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// Can every group use unconditional 16-byte loads?
template <typename CT> bool fast_ok(const pq3d_gemm_desc& d) {
  constexpr int EPL = Mma<CT>::EPL;
  if (d.transA && !d.transB) return false;
  if (d.lda % EPL || d.ldb % EPL || d.strideA % EPL || d.strideB % EPL) return false;
  if ((!d.transA || !d.transB) && (d.K % EPL)) return false;
  if (d.transA && (d.M % EPL || d.M < EPL)) return false;
  if (d.transB && (d.N % EPL || d.N < EPL)) return false;
  for (int g = 0; g < d.groups; ++g) {
    if (!aligned16(d.A[g]) || !aligned16(d.B[g])) return false;
    if (d.A2[g] && (!aligned16(d.A2[g]) || d.dtA2 != PQ3D_F32)) return false;
    if (d.B2[g] && (!aligned16(d.B2[g]) || d.dtB2 != PQ3D_F32)) return false;
  }
  return true;
}

#define LAUNCH(...) hipLaunchKernelGGL((__VA_ARGS__), grid, dim3(NT), 0, s, d)

template <typename CT, typename TA, typename TB>
void launch_fast_layout(const pq3d_gemm_desc& d, dim3 grid, hipStream_t s) {
  // optional addends A2 / B2 are always fp32 on the fast path (residual-stream position encodings)
  if (!d.transA && !d.transB) LAUNCH(gemm_fast_kernel<CT, TA, float, TB, float, false, false>);
  else if (!d.transA && d.transB) LAUNCH(gemm_fast_kernel<CT, TA, float, TB, float, false, true>);
  else LAUNCH(gemm_fast_kernel<CT, TA, float, TB, float, true, true>);
}

template <typename CT> void launch_slow(const pq3d_gemm_desc& d, dim3 grid, hipStream_t s) {
  if (!d.transA && !d.transB) LAUNCH(gemm_slow_kernel<CT, false, false>);
  else if (!d.transA && d.transB) LAUNCH(gemm_slow_kernel<CT, false, true>);
  else if (d.transA && d.transB) LAUNCH(gemm_slow_kernel<CT, true, true>);
  else LAUNCH(gemm_slow_kernel<CT, true, false>);
}

}  // namespace

extern "C" int pq3d_gemm(const pq3d_gemm_desc* dp, void* stream) {
  PQ_CHECK_ARG(dp != nullptr, "pq3d_gemm: null descriptor");
  pq3d_gemm_desc d = *dp;
  PQ_CHECK_ARG(d.M >= 0 && d.N >= 0 && d.K >= 0, "pq3d_gemm: negative dims");
  PQ_CHECK_ARG(d.groups >= 1 && d.groups <= PQ3D_MAX_GROUPS, "pq3d_gemm: groups out of range");
  PQ_CHECK_ARG(d.batch >= 1, "pq3d_gemm: batch < 1");
  PQ_CHECK_ARG(d.ct == PQ3D_F32 || d.ct == PQ3D_BF16, "pq3d_gemm: bad compute type");
  if (d.M == 0 || d.N == 0) return 0;
  const int kc = d.kconcat > 0 ? d.kconcat : 1;
  PQ_CHECK_ARG(d.groups % kc == 0, "pq3d_gemm: groups must be a multiple of kconcat");
  for (int g = 0; g < d.groups; ++g) {
    PQ_CHECK_ARG(d.A[g] && d.B[g] && (d.C[g] || (g % kc) != 0), "pq3d_gemm: null A/B/C");
    PQ_CHECK_ARG(!d.act_grad || d.aux[g] || (g % kc) != 0, "pq3d_gemm: act_grad needs aux");
  }
  if (d.splitk < 1) d.splitk = 1;
  PQ_CHECK_ARG(!(kc > 1 && d.splitk > 1), "pq3d_gemm: kconcat and split-K are exclusive");
  hipStream_t s = (hipStream_t)stream;
  if (d.splitk > 1) {
    PQ_CHECK_ARG(d.dtC == PQ3D_F32, "pq3d_gemm: split-K needs fp32 C");
    PQ_CHECK_ARG(d.ldc == d.N && (d.batch == 1 || d.strideC == (int64_t)d.M * d.N),
                 "pq3d_gemm: split-K needs contiguous C");
    for (int g = 0; g < d.groups && !d.accumulate; ++g) {
      hipError_t e = hipMemsetAsync(d.C[g], 0, sizeof(float) * (size_t)d.batch * d.M * d.N, s);
      if (e != hipSuccess) { pq3d_set_error(hipGetErrorString(e)); return (int)e; }
    }
  }
  const int tiles = ((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN);
  dim3 grid(tiles, d.splitk, (d.groups / kc) * d.batch);
  if (d.ct == PQ3D_BF16) {
    if (fast_ok<bf16_t>(d)) {
      const bool af = d.dtA == PQ3D_F32, bf = d.dtB == PQ3D_F32;
      if (af && bf) launch_fast_layout<bf16_t, float, float>(d, grid, s);
      else if (af && !bf) launch_fast_layout<bf16_t, float, bf16_t>(d, grid, s);
      else if (!af && bf) launch_fast_layout<bf16_t, bf16_t, float>(d, grid, s);
      else launch_fast_layout<bf16_t, bf16_t, bf16_t>(d, grid, s);
    } else {
      launch_slow<bf16_t>(d, grid, s);
    }
  } else {
    if (fast_ok<float>(d) && d.dtA == PQ3D_F32 && d.dtB == PQ3D_F32) launch_fast_layout<float, float, float>(d, grid, s);
    else launch_slow<float>(d, grid, s);
  }
  PQ_LAUNCH_CHECK();
  return 0;
}
