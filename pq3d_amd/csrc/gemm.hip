// Grouped / batched MFMA GEMM with fused prologue (A + A2, dtype conversion) and epilogue
// (bias, activation, activation-gradient, row masks, mask-head fill/threshold, split-K atomics).
// One kernel family serves every nn.Linear forward/backward on the path and the mask-head einsum.
//
// Tiling: 64x64 output tile per 256-thread workgroup (4 waves as 2x2, each wave a 32x32 sub-tile = 2x2 MFMA
// 16x16 tiles), K-step of 128 bytes of compute type per LDS row (64 bf16 / 32 f32), register-prefetched so the
// next tile's global loads overlap the current tile's MFMAs.  LDS rows are padded by 16 B.
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

// Stage one operand tile (ROWS x BKE) from global into two packed 16-byte registers per thread.
// Non-transposed: element (r,k) at base[off + r*ld + k].  Transposed: at base[off + k*ld + r].
template <typename CT, bool TR>
PQ_DEV void stage_load(u32x4 (&reg)[2], const void* base, const void* base2, int dt, int dt2, long off, long ld,
                       int r0, int R, int k0, int K, int tid) {
  typedef Tile<CT> T;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int c = tid + it * NT;
    float v[T::EPL];
#pragma unroll
    for (int j = 0; j < T::EPL; ++j) v[j] = 0.f;
    if (!TR) {
      const int row = c / T::CPR, kc = c % T::CPR;
      const int gr = r0 + row, gk = k0 + kc * T::EPL;
      if (gr < R && gk < K) {
        const int valid = min(T::EPL, K - gk);
        load_elems<T::EPL>(base, dt, off + (long)gr * ld + gk, valid, v);
        if (base2) {
          float w[T::EPL];
          load_elems<T::EPL>(base2, dt2, off + (long)gr * ld + gk, valid, w);
#pragma unroll
          for (int j = 0; j < T::EPL; ++j) v[j] += w[j];
        }
      }
    } else {
      constexpr int RC = BM / T::EPL;  // row-chunks per k (BM == BN)
      const int kk = c / RC, rc = c % RC;
      const int gk = k0 + kk, gr = r0 + rc * T::EPL;
      if (gk < K && gr < R) {
        const int valid = min(T::EPL, R - gr);
        load_elems<T::EPL>(base, dt, off + (long)gk * ld + gr, valid, v);
        if (base2) {
          float w[T::EPL];
          load_elems<T::EPL>(base2, dt2, off + (long)gk * ld + gr, valid, w);
#pragma unroll
          for (int j = 0; j < T::EPL; ++j) v[j] += w[j];
        }
      }
    }
    reg[it] = pack_frag<CT>(v);
  }
}

template <typename CT, bool TR>
PQ_DEV void stage_store(const u32x4 (&reg)[2], CT* lds, int tid) {
  typedef Tile<CT> T;
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

template <typename CT, bool TA, bool TB>
__global__ __launch_bounds__(NT) void gemm_kernel(const pq3d_gemm_desc d) {
  typedef Tile<CT> T;
  __shared__ __attribute__((aligned(16))) CT As[BM * T::LDK];
  __shared__ __attribute__((aligned(16))) CT Bs[BN * T::LDK];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const int g = d.kconcat ? 0 : blockIdx.z / d.batch, z = d.kconcat ? blockIdx.z : blockIdx.z % d.batch;
  const int tiles_m = (d.M + BM - 1) / BM;
  const int m0 = (blockIdx.x % tiles_m) * BM, n0 = (blockIdx.x / tiles_m) * BN;

  const int nkt = (d.K + T::BKE - 1) / T::BKE;
  int kt0 = 0, kt1 = nkt;
  if (d.splitk > 1) {
    const int per = (nkt + d.splitk - 1) / d.splitk;
    kt0 = blockIdx.y * per;
    kt1 = min(nkt, kt0 + per);
    if (kt0 >= kt1) return;
  }

  const int ng = d.kconcat ? d.groups : 1;  // groups walked inside the K loop
  const long offA = (long)z * d.strideA, offB = (long)z * d.strideB;
  const int nit = (kt1 - kt0) * ng;         // flattened (group, k-tile) iterations

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  u32x4 ra[2], rb[2];
  auto load_it = [&](int it) {
    const int gg = d.kconcat ? it / (kt1 - kt0) : g;
    const int kt = kt0 + (d.kconcat ? it % (kt1 - kt0) : it);
    stage_load<CT, TA>(ra, d.A[gg], d.A2[gg], d.dtA, d.dtA2, offA, d.lda, m0, d.M, kt * T::BKE, d.K, tid);
    stage_load<CT, TB>(rb, d.B[gg], d.B2[gg], d.dtB, d.dtB2, offB, d.ldb, n0, d.N, kt * T::BKE, d.K, tid);
  };
  load_it(0);

  for (int it = 0; it < nit; ++it) {
    stage_store<CT, TA>(ra, As, tid);
    stage_store<CT, TB>(rb, Bs, tid);
    __syncthreads();
    if (it + 1 < nit) load_it(it + 1);
#pragma unroll
    for (int ks = 0; ks < T::BKE / T::KSTEP; ++ks) {
      u32x4 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        fa[i] = *(const u32x4*)&As[(wm + i * 16 + li) * T::LDK + ks * T::KSTEP + lg * T::EPL];
#pragma unroll
      for (int j = 0; j < 2; ++j)
        fb[j] = *(const u32x4*)&Bs[(wn + j * 16 + li) * T::LDK + ks * T::KSTEP + lg * T::EPL];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) Mma<CT>::mma(acc[i][j], fa[i], fb[j]);
    }
    __syncthreads();
  }

  // ------------------------------------------------------------------ epilogue
  void* C = d.C[g];
  void* C2 = d.C2[g];
  const void* aux = d.aux[g];
  const void* bias = d.bias[g];
  const uint8_t* rmask = d.row_mask[g];
  const long offC = (long)z * d.strideC;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn + j * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm + i * 16 + lg * 4 + r;
        if (row >= d.M || col >= d.N) continue;
        float v = acc[i][j][r] * d.alpha;
        const long ci = offC + (long)row * d.ldc + col;
        if (d.splitk > 1) {
          unsafeAtomicAdd((float*)C + ci, v);
          continue;
        }
        if (bias) v += load_elem(bias, d.dtBias, col);
        if (C2) store_elem(C2, d.dtC2, ci, v);
        if (d.act == PQ3D_ACT_RELU) v = fmaxf(v, 0.f);
        else if (d.act == PQ3D_ACT_GELU) v = gelu_f(v);
        if (d.act_grad == PQ3D_ACT_RELU) v = load_elem(aux, d.dtAux, ci) > 0.f ? v : 0.f;
        else if (d.act_grad == PQ3D_ACT_GELU) v *= gelu_grad_f(load_elem(aux, d.dtAux, ci));
        const long ri = (long)z * d.M + row;
        if (rmask && rmask[ri] == 0) v = 0.f;
        if (d.row_scale) v *= d.row_scale[ri];
        if (d.row_fill_flag && d.row_fill_flag[ri]) v = d.row_fill;
        store_elem(C, d.dtC, ci, v);
        if (d.mask_out) d.mask_out[((long)z * d.N + col) * d.M + row] = (1.f / (1.f + __expf(-v)) < 0.5f) ? 1 : 0;
      }
    }
  }
}

template <typename CT>
int launch_ct(const pq3d_gemm_desc& d, dim3 grid, hipStream_t s) {
  if (!d.transA && !d.transB) hipLaunchKernelGGL((gemm_kernel<CT, false, false>), grid, dim3(NT), 0, s, d);
  else if (!d.transA && d.transB) hipLaunchKernelGGL((gemm_kernel<CT, false, true>), grid, dim3(NT), 0, s, d);
  else if (d.transA && d.transB) hipLaunchKernelGGL((gemm_kernel<CT, true, true>), grid, dim3(NT), 0, s, d);
  else hipLaunchKernelGGL((gemm_kernel<CT, true, false>), grid, dim3(NT), 0, s, d);
  PQ_LAUNCH_CHECK();
  return 0;
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
  for (int g = 0; g < d.groups; ++g) {
    PQ_CHECK_ARG(d.A[g] && d.B[g] && (d.C[g] || (d.kconcat && g > 0)), "pq3d_gemm: null A/B/C");
    PQ_CHECK_ARG(!d.act_grad || d.aux[g], "pq3d_gemm: act_grad needs aux");
  }
  PQ_CHECK_ARG(!(d.kconcat && d.splitk > 1), "pq3d_gemm: kconcat and split-K are exclusive");
  if (d.splitk < 1) d.splitk = 1;
  hipStream_t s = (hipStream_t)stream;
  if (d.splitk > 1) {
    PQ_CHECK_ARG(d.dtC == PQ3D_F32, "pq3d_gemm: split-K needs fp32 C");
    PQ_CHECK_ARG(d.ldc == d.N && (d.batch == 1 || d.strideC == (int64_t)d.M * d.N),
                 "pq3d_gemm: split-K needs contiguous C");
    for (int g = 0; g < d.groups; ++g) {
      hipError_t e = hipMemsetAsync(d.C[g], 0, sizeof(float) * (size_t)d.batch * d.M * d.N, s);
      if (e != hipSuccess) { pq3d_set_error(hipGetErrorString(e)); return (int)e; }
    }
  }
  const int tiles = ((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN);
  dim3 grid(tiles, d.splitk, (d.kconcat ? 1 : d.groups) * d.batch);
  return d.ct == PQ3D_BF16 ? launch_ct<bf16_t>(d, grid, s) : launch_ct<float>(d, grid, s);
}
