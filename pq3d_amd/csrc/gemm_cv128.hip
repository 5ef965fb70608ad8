// 128 x 128-tile GEMMs for the BIG products whose operands are stored in fp32 (or bf16 activations against fp32 weights)
// and converted in flight -- the query side of a large batch (the shipped stage-2 decoder: M = 128 scenes x 80 objects =
// 10240 rows), chosen inside pq3d_gemm when the launch has enough such tiles:
//   X3        C[m][n] = sum_k A[m][k] B[n][k], fp32 operands (optional fp32 addend A2: x + pos), split-bf16 arithmetic
//             (hi.hi + hi.lo + lo.hi: fp32-grade results from the bf16 matrix cores) -- the forward projections and FFN
//   bf16, NT  the same product with operands rounded to bf16 once
//   bf16, NN  C[m][n] = sum_k A[m][k] B[k][n] (B row-major [k][n]: dX = dY W), A fp32 or bf16 -- the input gradients
// all with gemm.hip's full fused epilogue (epi_row: bias, activation, dropout, activation gradient / "+ aux", row masks,
// second output), and split-K requests served as ONE pass (C = / += product: deterministic, no
// atomics -- at these sizes the rows alone fill the chip).
//
// Why a second tile shape: these launches are bound by the L2 -> CU operand stream, not by HBM or the matrix pipe.  A
// 64 x 64 tile loads (64 + 64) x 4 B of fp32 operands per k for 2 x 64 x 64 flops = 16 flop/B (measured 360 TFLOP/s at
// M = 10240, N = K = 768, i.e. ~22 TB/s out of the L2s); 128 x 128 doubles the flops per byte.  For X3 also the LDS side:
// a 32 x 32 wave tile reads 4 + 4 hi / lo fragments for 12 MFMAs, a 64 x 64 wave tile 8 + 8 for 48.
// Same rounding points, same k order, same term order per accumulator as gemm.hip's kernels -> bit-identical results
// (tests/test_gpu_ops.py compares with the same product launched in 1024-row pieces, which stay on those kernels).
// Bound: L2 operand stream; algorithmic bytes per group (M + N) K 4 [+ M K 4 with an addend] + M N 4.
#include "gemm_common.h"

namespace {

constexpr int XM = 128, XN = 128, XK = 32;
constexpr int XLD = XK + 8;      // [row][k] tiles: 80 B rows -> conflict-free 16-byte fragment reads
constexpr int XLT = XN + 16;     // [k][n] tile of a row-major B (NN): 288 B rows -> conflict-free transposing reads
constexpr int XCLD = XN + 4;     // fp32 row of the transposed half C tile

template <typename TA, bool HA2> struct XStage {
  Raw<TA, 8> a[2];
  Raw<float, 8> a2[HA2 ? 2 : 1], b[2];
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

// X3: split-bf16 (fp32 A and B, NT only).  TRB: B is row-major [k][n].  TA: storage type of A (float / bf16).
template <bool X3, bool HA2, bool TRB, typename TA>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_cv128_kernel(const pq3d_kdesc d) {
  static_assert(!X3 || (!TRB && sizeof(TA) == 4), "split-bf16: NT, fp32 operands");
  static_assert(!HA2 || (!TRB && sizeof(TA) == 4), "addend: NT, fp32 A");
  constexpr int NA = X3 ? 2 : 1;                                     // hi (+ lo) tiles per operand
  constexpr int A_EL = NA * XM * XLD, B_EL = TRB ? XK * XLT : NA * XN * XLD;
  constexpr int OP_B = (int)sizeof(bf16_t) * (A_EL + B_EL), C_B = (int)sizeof(float) * (XM / 2) * XCLD;
  __shared__ __attribute__((aligned(16))) bf16_t S[(OP_B > C_B ? OP_B : C_B) / 2];   // operands; epilogue: fp32 [64][XCLD]
  bf16_t* const Ah = S;
  bf16_t* const Al = S + XM * XLD;          // X3 only
  bf16_t* const Bh = S + A_EL;
  bf16_t* const Bl = Bh + XN * XLD;         // X3 only
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

  // staging: thread -> 2 chunks of 8 elements per operand.  [row][k] tiles: chunk c = tid + 256 it -> row c / 4, k chunk
  // c % 4; the [k][n] tile of a row-major B: k row c / 16, n chunk c % 16.
  long aoff[2], boff[2];
  int alds[2], blds[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int c = tid + it * 256, row = c >> 2, kq = c & 3;
    aoff[it] = (long)min(m0 + row, d.M - 1) * d.lda + kq * 8;   // rows past M: clamped duplicates (never stored)
    alds[it] = row * XLD + kq * 8;
    if constexpr (TRB) {
      const int kk = c >> 4, nq = c & 15;
      boff[it] = (long)kk * d.ldb + n0 + nq * 8;
      blds[it] = kk * XLT + nq * 8;
    } else {
      boff[it] = (long)(n0 + row) * d.ldb + kq * 8;
      blds[it] = alds[it];
    }
  }
  const long bstep = TRB ? (long)XK * d.ldb : (long)XK;
  const int nk = d.K / XK, last = nk - 1;
  // a group without an addend inside an addend launch re-reads A with weight 0 (gemm.hip's rule)
  const float scale2 = HA2 && gp.A2 ? 1.f : 0.f;
  const TA* const A = (const TA*)gp.A;
  const float* const A2 = HA2 && gp.A2 ? (const float*)gp.A2 : (const float*)gp.A;
  const float* const B = (const float*)gp.B;
  auto load = [&](XStage<TA, HA2>& s, int t) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      s.a[it].load(A + aoff[it] + t * XK);
      if constexpr (HA2) s.a2[it].load(A2 + aoff[it] + t * XK);
      s.b[it].load(B + boff[it] + t * bstep);
    }
  };
  auto put = [&](const XStage<TA, HA2>& s) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      float v[8];
      if constexpr (sizeof(TA) == 2) {
        *(u32x4*)&Ah[alds[it]] = s.a[it].a;   // already bf16: straight through
      } else {
        s.a[it].to_float(v);
        if constexpr (HA2) {
          float w[8];
          s.a2[it].to_float(w);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += scale2 * w[j];
        }
        if constexpr (X3) split_store(v, Ah, Al, alds[it]);
        else *(u32x4*)&Ah[alds[it]] = pack_frag<bf16_t>(v);
      }
      s.b[it].to_float(v);
      if constexpr (X3) split_store(v, Bh, Bl, blds[it]);
      else *(u32x4*)&Bh[blds[it]] = pack_frag<bf16_t>(v);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto mult = [&]() {
    u32x4 ah[4], al[X3 ? 4 : 1];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int o = (wm + i * 16 + li) * XLD + lg * 8;
      ah[i] = *(const u32x4*)&Ah[o];
      if constexpr (X3) al[i] = *(const u32x4*)&Al[o];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      u32x4 bh, bl;
      if constexpr (TRB) {
        bh = km_frag(Bh, XLT, wn + j * 16, 0, li, lg);
      } else {
        const int o = (wn + j * 16 + li) * XLD + lg * 8;
        bh = *(const u32x4*)&Bh[o];
        if constexpr (X3) bl = *(const u32x4*)&Bl[o];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (X3) {   // term order of gemm.hip's mma_tile_x3
          Mma<bf16_t>::mma(acc[i][j], al[i], bh);
          Mma<bf16_t>::mma(acc[i][j], ah[i], bl);
        }
        Mma<bf16_t>::mma(acc[i][j], ah[i], bh);
      }
    }
  };

  // Two register stages: k-tiles t + 1 and t + 2 are in flight while tile t is multiplied.  Every load is UNCONDITIONAL
  // (past the end the last tile is requested again and never used) and pinned in program order by sched_barrier(0):
  // with a branch around a stage's loads, with the two prologue stages swapped, or with the next stage's conversion
  // arithmetic hoisted above the MFMA phase (the scheduler did all three), the compiler cannot count the loads in flight
  // and waits for vmcnt(0) at the loop head.  The same goes for the bias row (read from the B operand when there is
  // none to add early) and for the loop body: an even number of k-tiles (host), so both halves always run.
#define CV_PHASE() __builtin_amdgcn_sched_barrier(0)
  const bool bias_early = gp.bias != nullptr && d.dtBias == PQ3D_F32 && d.alpha == 1.f;   // gemm.hip's rule (no split-K here)
  const float* bsrc = bias_early ? (const float*)gp.bias + n0 : (const float*)gp.B;
  float bcol[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bcol[j] = bsrc[wn + j * 16 + li];
  XStage<TA, HA2> s0, s1;
  CV_PHASE();
  load(s0, 0);
  CV_PHASE();
  load(s1, 1);
  CV_PHASE();
  for (int t = 0; t < nk; t += 2) {
    put(s0);
    __syncthreads();
    CV_PHASE();
    load(s0, min(t + 2, last));
    CV_PHASE();
    mult();
    __syncthreads();
    CV_PHASE();
    put(s1);
    __syncthreads();
    CV_PHASE();
    load(s1, min(t + 3, last));
    CV_PHASE();
    mult();
    __syncthreads();
    CV_PHASE();
  }
#undef CV_PHASE
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

// Eligibility is decided here so that pq3d_gemm stays the single entry point (gemm.hip calls this for its split-bf16
// branch and, after the bf16 x bf16 128-tile kernels, for every other product).
bool pq3d_gemm_cv128_try(const pq3d_gemm_desc& d, const pq3d_kdesc& kd, hipStream_t s) {
  const bool x3 = d.ct == PQ3D_BF16X3;
  if (d.ct != PQ3D_BF16 && !x3) return false;
  if (d.transA || d.batch != 1 || d.dtB != PQ3D_F32 || (d.dtA != PQ3D_F32 && d.dtA != PQ3D_BF16)) return false;
  if (x3 && (d.transB || d.dtA != PQ3D_F32)) return false;
  if (d.kconcat > 1) return false;
  if (d.N % XN || d.K % (2 * XK) || d.M < XM) return false;   // an even number of k-tiles: the kernel's loop has no tail
  if (d.lda % 8 || d.ldb % 8) return false;
  // split-K requests are served as one pass: C = product, or (accumulate) C += product through the "+ aux" epilogue
  // reading C itself -- every element is read and written by the same thread.  Both need an otherwise plain epilogue.
  if (d.splitk > 1) {
    if (d.dtC != PQ3D_F32 || d.act || d.act_grad || (d.drop.p > 0.f && d.drop.seed) || d.row_scale || d.row_fill_flag ||
        d.mask_out) return false;
    for (int g = 0; g < d.groups; ++g)
      if (d.bias[g] || d.aux[g] || d.C2[g] || d.row_mask[g]) return false;
  }
  bool a2 = false;
  for (int g = 0; g < d.groups; ++g) {
    if (d.B2[g] || d.colsum[g]) return false;
    if (d.A2[g]) { a2 = true; if (d.dtA != PQ3D_F32 || d.dtA2 != PQ3D_F32 || d.transB || (((uintptr_t)d.A2[g]) & 15)) return false; }
    if ((((uintptr_t)d.A[g]) | ((uintptr_t)d.B[g])) & 15) return false;
  }
  // enough tiles to give every CU two workgroups, or one with a long reduction
  const long tiles = (long)((d.M + XM - 1) / XM) * (d.N / XN) * d.groups;
  if (tiles < 512 && !(tiles >= 256 && d.K >= 1024)) return false;
  pq3d_kdesc k = kd;
  if (d.splitk > 1) {
    k.splitk = 1;
    if (d.accumulate) {
      k.act_grad = PQ3D_ACT_ADD;
      k.dtAux = PQ3D_F32;
      for (int g = 0; g < d.groups; ++g) k.gp[g].aux = d.C[g];
    }
  }
  k.xcd_order = xcd_order_for(tiles, (long)d.N * d.K * 4, shared_a_run(d));
  const dim3 grid((d.M + XM - 1) / XM, d.N / XN, d.groups);
#define CV_LAUNCH(...) hipLaunchKernelGGL((gemm_cv128_kernel<__VA_ARGS__>), grid, dim3(256), 0, s, k)
  const bool af = d.dtA == PQ3D_F32;
  if (x3) { if (a2) CV_LAUNCH(true, true, false, float); else CV_LAUNCH(true, false, false, float); }
  else if (d.transB) { if (af) CV_LAUNCH(false, false, true, float); else CV_LAUNCH(false, false, true, bf16_t); }
  else if (a2) CV_LAUNCH(false, true, false, float);
  else if (af) CV_LAUNCH(false, false, false, float);
  else CV_LAUNCH(false, false, false, bf16_t);
#undef CV_LAUNCH
  return true;
}
