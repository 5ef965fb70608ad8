// 128 x 128-tile split-bf16 (PQ3D_BF16X3) NT GEMM for the BIG query-side products: C[m][n] = sum_k A[m][k] B[n][k] with fp32
// operands (optional fp32 addend A2: x + pos), fp32-grade results from the bf16 matrix cores (hi.hi + hi.lo + lo.hi, 3
// MFMAs per term pair), the full fused epilogue of gemm.hip (epi_row).  Chosen inside pq3d_gemm when the launch has >= 512
// such tiles: the shipped stage-2 decoder runs its projections and FFN at M = 128 scenes x 80 objects = 10240 rows.
//
// Why a second tile shape: the 64 x 64 kernel reads 4 + 4 hi / lo fragments from LDS for 12 MFMAs (a 32 x 32 wave tile) and
// writes 32 KB of converted operands per 64-wide k-tile -- 768 LDS cycles per workgroup against 384 MFMA cycles per SIMD,
// i.e. LDS-bound at best half of the matrix pipe (measured: 22 % at M = 10240, N = K = 768).  A 64 x 64 wave tile reads
// 8 + 8 fragments for 48 MFMAs: LDS and MFMA time per k-step balance.  Same hi / lo values, same k order and the same
// three-term order per accumulator as gemm.hip's mma_tile_x3, so the results are bit-identical to the 64 x 64 path.
// Bound: MFMA (3 x 2 M N K flops per group); algorithmic bytes (M + N) K 4 [+ M K 4 with an addend] + M N 4.
#include "gemm_common.h"

namespace {

constexpr int XM = 128, XN = 128, XK = 32, XLD = XK + 8;   // padded LDS row: 80 B -> conflict-free 16-byte fragment reads
constexpr int XCLD = XN + 4;                                // fp32 row of the transposed half C tile

template <bool HA2> struct XStage {
  Raw<float, 8> a[2], a2[HA2 ? 2 : 1], b[2];
};

PQ_DEV void split_store(const float (&v)[8], bf16_t* hi_t, bf16_t* lo_t, int o) {
  const u32x4 hi = pack_frag<bf16_t>(v);
  float w[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    w[2 * j] = v[2 * j] - __uint_as_float(hi[j] << 16);
    w[2 * j + 1] = v[2 * j + 1] - __uint_as_float(hi[j] & 0xffff0000u);
  }
  *(u32x4*)&hi_t[o] = hi;
  *(u32x4*)&lo_t[o] = pack_frag<bf16_t>(w);
}

template <bool HA2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_x3_128_kernel(const pq3d_kdesc d) {
  __shared__ __attribute__((aligned(16))) bf16_t S[4 * XM * XLD];   // Ah | Al | Bh | Bl (40 KB); epilogue: fp32 [64][XCLD]
  static_assert(sizeof(bf16_t) * 4 * XM * XLD >= sizeof(float) * (XM / 2) * XCLD, "half C tile must fit in the staging LDS");
  bf16_t* const Ah = S;
  bf16_t* const Al = S + XM * XLD;
  bf16_t* const Bh = S + 2 * XM * XLD;
  bf16_t* const Bl = S + 3 * XM * XLD;
  const TileIdx ti = tile_index(d.xcd_order);
  const int g = ti.z, m0 = ti.x * XM, n0 = ti.y * XN;
  GPtrs gp;
  gp.load(d, g);
#ifndef PQ3D_NO_KARG_PIN
  asm volatile("" ::"s"(d.M), "s"(d.N), "s"(d.K), "s"(d.lda), "s"(d.ldb), "s"(d.ldc), "s"(d.alpha), "s"(d.act), "s"(d.act_grad),
               "s"(d.dtC), "s"(d.dtC2), "s"(d.dtAux), "s"(d.dtBias), "s"(d.row_fill), "s"(d.row_scale), "s"(d.row_fill_flag),
               "s"(d.mask_out), "s"(gp.A), "s"(gp.A2), "s"(gp.B), "s"(gp.bias), "s"(gp.aux), "s"(gp.C), "s"(gp.C2),
               "s"(gp.row_mask));
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

  // staging: thread -> 2 chunks of 8 k per operand (chunk c = tid + 256 it: row c / 4, k chunk c % 4)
  const float* pa[2];
  const float* pa2[2];
  const float* pb[2];
  int lofs[2];
  const float scale2 = HA2 && gp.A2 ? 1.f : 0.f;   // a group without an addend inside an addend launch re-reads A with weight 0
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int c = tid + it * 256, row = c >> 2, kc = c & 3;
    const long ao = (long)min(m0 + row, d.M - 1) * d.lda + kc * 8;   // rows past M: clamped duplicates (never stored)
    pa[it] = (const float*)gp.A + ao;
    pa2[it] = HA2 && gp.A2 ? (const float*)gp.A2 + ao : pa[it];
    pb[it] = (const float*)gp.B + (long)(n0 + row) * d.ldb + kc * 8;
    lofs[it] = row * XLD + kc * 8;
  }
  auto load = [&](XStage<HA2>& s, int t) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      s.a[it].load(pa[it] + t * XK);
      if constexpr (HA2) s.a2[it].load(pa2[it] + t * XK);
      s.b[it].load(pb[it] + t * XK);
    }
  };
  auto put = [&](const XStage<HA2>& s) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      float v[8];
      s.a[it].to_float(v);
      if constexpr (HA2) {
        float w[8];
        s.a2[it].to_float(w);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += scale2 * w[j];
      }
      split_store(v, Ah, Al, lofs[it]);
      s.b[it].to_float(v);
      split_store(v, Bh, Bl, lofs[it]);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto mult = [&]() {
    u32x4 ah[4], al[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int o = (wm + i * 16 + li) * XLD + lg * 8;
      ah[i] = *(const u32x4*)&Ah[o];
      al[i] = *(const u32x4*)&Al[o];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int o = (wn + j * 16 + li) * XLD + lg * 8;
      const u32x4 bh = *(const u32x4*)&Bh[o], bl = *(const u32x4*)&Bl[o];
#pragma unroll
      for (int i = 0; i < 4; ++i) {   // term order of gemm.hip's mma_tile_x3
        Mma<bf16_t>::mma(acc[i][j], al[i], bh);
        Mma<bf16_t>::mma(acc[i][j], ah[i], bl);
        Mma<bf16_t>::mma(acc[i][j], ah[i], bh);
      }
    }
  };

  // two register stages: k-tiles t + 1 and t + 2 are in flight while tile t is multiplied.  Every load below is
  // UNCONDITIONAL (past the end the last tile is requested again and never used): with a branch around a stage's loads
  // the compiler cannot count the loads in flight and waits for vmcnt(0) -- i.e. for the stage it has just issued --
  // before every conversion.
  // The same goes for the bias row (read from the B operand with weight 0 when there is none to add early) and for the
  // loop body: K % 64 == 0 (host), so both halves always run.
  const bool bias_early = gp.bias != nullptr && d.dtBias == PQ3D_F32 && d.alpha == 1.f;   // gemm.hip's rule (no split-K here)
  const float* bsrc = bias_early ? (const float*)gp.bias + n0 : (const float*)gp.B;
  float bcol[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bcol[j] = bsrc[wn + j * 16 + li];
  XStage<HA2> s0, s1;
  const int nk = d.K / XK, last = nk - 1;
  load(s0, 0);
  __builtin_amdgcn_sched_barrier(0);   // in this order: swapped (the scheduler did), the loop head must wait for vmcnt(0) on every trip
  load(s1, 1);
  __builtin_amdgcn_sched_barrier(0);
  // sched_barrier(0) at every phase boundary: left alone, the scheduler hoists the conversion arithmetic of the NEXT stage
  // above this stage's MFMA phase (to overlap it) -- which needs that stage's loads complete, i.e. vmcnt(0) at the loop
  // head and one k-tile of prefetch distance instead of two; and it sinks the loads behind the MFMA phase.
#define X3_PHASE() __builtin_amdgcn_sched_barrier(0)
  for (int t = 0; t < nk; t += 2) {
    put(s0);
    __syncthreads();
    X3_PHASE();
    load(s0, min(t + 2, last));
    X3_PHASE();
    mult();
    __syncthreads();
    X3_PHASE();
    put(s1);
    __syncthreads();
    X3_PHASE();
    load(s1, min(t + 3, last));
    X3_PHASE();
    mult();
    __syncthreads();
    X3_PHASE();
  }
#undef X3_PHASE
  if (bias_early) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] += bcol[j];
  }

  // epilogue: the rows leave in two halves (one per pair of waves) through LDS, every thread then owns 16 contiguous
  // columns of a row twice -> gemm.hip's fused epilogue on whole 16-byte pieces
  float* Cf = (float*)S;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (wm == h * 64) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) Cf[(i * 16 + 4 * lg + r) * XCLD + wn + j * 16 + li] = acc[i][j][r] * d.alpha;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int idx = tid + p * 256, lrow = idx >> 3, lcol = (idx & 7) * 16;
      const int row = m0 + h * 64 + lrow;
      if (row < d.M) {
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; j += 4) {
          const float4 q = *(const float4*)&Cf[lrow * XCLD + lcol + j];
          v[j] = q.x; v[j + 1] = q.y; v[j + 2] = q.z; v[j + 3] = q.w;
        }
        epi_row<16>(d, gp, v, g, 0, row, n0 + lcol, bias_early);
      }
    }
    __syncthreads();
  }
}

}  // namespace

// Eligibility beyond what gemm.hip's split-bf16 branch has already established (NT layout, fp32 A / B, 16-byte aligned
// rows, no split-K): plain grouped launch, whole tiles along N and K, enough tiles to fill the chip twice.
bool pq3d_gemm_x3_128_try(const pq3d_gemm_desc& d, const pq3d_kdesc& kd, hipStream_t s, bool a2) {
  if (d.kconcat > 1 || d.batch != 1 || d.splitk > 1) return false;
  if (d.N % XN || d.K % (2 * XK) || d.M < XM) return false;   // an even number of k-tiles: the kernel's loop has no tail
  const long tiles = (long)((d.M + XM - 1) / XM) * (d.N / XN) * d.groups;
  if (tiles < 512) return false;
  pq3d_kdesc k = kd;
  k.xcd_order = xcd_order_for(tiles, (long)d.N * d.K * 4, shared_a_run(d));
  const dim3 grid((d.M + XM - 1) / XM, d.N / XN, d.groups);
  if (a2) hipLaunchKernelGGL((gemm_x3_128_kernel<true>), grid, dim3(256), 0, s, k);
  else hipLaunchKernelGGL((gemm_x3_128_kernel<false>), grid, dim3(256), 0, s, k);
  return true;
}
