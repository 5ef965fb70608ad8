// 128-row-tile bf16 GEMMs for the big streaming products of the path (chosen inside pq3d_gemm when both operands are
// bf16 and the launch still fills the chip):
//   gemm_nt128_kernel : C_o[m][n] = act( sum_{g in o} sum_k A_g[m][k] B_g[n][k] + bias[n] ) [+ aux],  k contiguous in A and B;
//                       hoisted K/V projections, their input gradients (K-concatenated over layers, fp32 out, "+ aux"),
//                       the PointNet++ tokenizer's SharedMLP layers (ReLU)
//   gemm_tt128_kernel : C[m][n] += sum_k A[k][m] B[k][n]  (weight gradients dW = dY^T X, split-K atomics, bias gradient)
//
// Why a second tile shape: at K = 256 a 64x64 tile loads 64 KB of operands for 8 KB of output (8 B of L2 traffic per
// output byte -- 786 MB for the c2 hoisted projection, which is L2-bandwidth time, not MFMA time); 128x128 halves that.
// Only the plain cases live here (no prologue adds, GELU, masks) so the register budget goes to the 4x4 accumulator
// block per wave (64 registers, 3 waves per SIMD) instead of options; everything else stays in gemm.hip.  The k order
// (one accumulator per output, 32-wide MFMA steps in sequence) is gemm.hip's, so the NT kernel produces the same bits.
// K % 64 == 0, N % 128 == 0 (M % 128 == 0 for TT).  Bound: L2 -> LDS traffic / HBM write of C; algorithmic bytes per
// group: (M + N) * K * 2 + M * N * {2, 4}.
#include <algorithm>

#include "common.h"

namespace {

constexpr int TM = 128, TN = 128, TK = 64, LDT = TK + 8;   // padded LDS row: 144 B -> conflict-free 16-byte fragment reads
constexpr int LDC = TN + 8;                                 // bf16 C staging row

// Operand staging: direct global -> LDS DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass), one
// 1 KB piece per wave instruction = 8 rows of the unpadded [128][64] tile.  The DMA writes lane-linear, so the
// bank-conflict swizzle lives in the SOURCE address and in the fragment read: 16-byte slot s of row r holds k chunk
// s ^ ((r >> 1) & 7) -- the 16 lanes of a fragment read (rows r0 .. r0 + 15, one chunk) then hit 16 distinct slots of the
// 64 banks.  Single LDS buffer, two barriers per k slice: the overlap of loads and MFMAs comes from 4 workgroups per CU
// (~100 VGPRs, 34 KB of LDS) instead of from staging registers (136 VGPRs, 3 per CU before).
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

// Tile shapes (rows x columns per workgroup; 4 waves as 2 x 2):
//   128 x 128  the streaming products with thousands of tiles (hoisted K/V projection: 3072 at config 2)
//   128 x 64   launches with < 3 workgroups of 128 x 128 per CU (the K-concatenated input-gradient products: 384 tiles at
//              config 2): twice the workgroups for the same k loop.  Measured per launch: 30.4 us (128 x 128), 26.5 us
//              (64 x 128), 27.3 us (128 x 64; kept: the streamed operand A takes 2/3 of each LDS stage)
// OUT: 0 = bf16 C, 1 = fp32 C (+ aux), 2 = bf16 hi / lo PLANES of the fp32 result (C = bf16(v), C2 = bf16(v - C): the K / V tensors
// of compute mode 'bf16x3', whose split-bf16 product arrives here as 3 K-concatenated bf16 groups hi.hi + lo.hi + hi.lo)
template <int OUT, int TMT, int TNT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_nt128_kernel(const pq3d_kdesc d) {
  constexpr bool F32OUT = OUT == 1;
#ifndef PQ3D_NO_KARG_PIN
  // kernel-argument prefetch (see gemm_fast_kernel): one batch of scalar loads for every descriptor scalar used below
  asm volatile("" ::"s"(d.M), "s"(d.N), "s"(d.K), "s"(d.kconcat), "s"(d.lda), "s"(d.ldb), "s"(d.ldc), "s"(d.alpha), "s"(d.act),
               "s"(d.act_grad), "s"(d.groups), "s"(d.xcd_order));
#endif
  constexpr int MI = TMT / 32, NJ = TNT / 32;                // 16 x 16 MFMA tiles per wave: MI x NJ
  constexpr int LDCT = TNT + 8, LDFT = TNT + 4, HR = TMT / 2;   // bf16 / fp32 C staging rows; rows per fp32 half
  constexpr int OPB = sizeof(bf16_t) * (TMT + TNT) * TK;
  constexpr int STB = F32OUT ? (int)sizeof(float) * HR * LDFT : (int)sizeof(bf16_t) * TMT * LDCT;
  constexpr int SM_BYTES = OPB > STB ? OPB : STB;
  __shared__ __attribute__((aligned(16))) bf16_t As[SM_BYTES / 2];
  bf16_t* const Bs = As + TMT * TK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const int wm = (wave >> 1) * (TMT / 2), wn = (wave & 1) * (TNT / 2);
  const int kc = d.kconcat > 0 ? d.kconcat : 1;   // kc consecutive groups are concatenated along K into one output
  const TileIdx ti = tile_index(d.xcd_order);   // hardware order, or the XCD-aware order of a big launch (common.h)
  const int g = ti.z * kc, m0 = ti.x * TMT, n0 = ti.y * TNT;
  const bf16_t* A = (const bf16_t*)d.gp[g].A;
  const bf16_t* B = (const bf16_t*)d.gp[g].B;
  const int nk1 = d.K / TK, nkt = nk1 * kc;

  // DMA piece p of this thread: LDS slot (p * 256 + tid) = row slot / 8, position slot % 8 <- k chunk pos ^ swizzle(row)
  int aoff[MI], boff[NJ];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int slot = p * 256 + tid, row = slot >> 3, c = (slot & 7) ^ ((row >> 1) & 7);
    if (p < MI) aoff[p] = min(m0 + row, d.M - 1) * (int)d.lda + c * 8;   // rows past M: clamped duplicates (never stored)
    if (p < NJ) boff[p] = (n0 + row) * (int)d.ldb + c * 8;
  }
  auto stage = [&](int t) {
    int kt = t;
    if (kc > 1) {   // uniform: switch to the operands of the group this k-tile belongs to
      const int gi = t / nk1;
      kt = t - gi * nk1;
      A = (const bf16_t*)d.gp[g + gi].A;
      B = (const bf16_t*)d.gp[g + gi].B;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if (p < MI)
        __builtin_amdgcn_global_load_lds((gptr_t*)(A + aoff[p] + kt * TK), (lptr_t*)(As + (p * 256 + wave * 64) * 8), 16, 0, 0);
      if (p < NJ)
        __builtin_amdgcn_global_load_lds((gptr_t*)(B + boff[p] + kt * TK), (lptr_t*)(Bs + (p * 256 + wave * 64) * 8), 16, 0, 0);
    }
  };
  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int swz = li >> 1;   // (row >> 1) & 7 of this lane's fragment rows (row = 16 * tile + li)
  for (int kt = 0; kt < nkt; ++kt) {
    stage(kt);
    __syncthreads();   // the compiler drains the DMA queue (vmcnt(0)) before the barrier
#pragma unroll
    for (int ks = 0; ks < TK / 32; ++ks) {
      u32x4 af[MI], bf[NJ];
      const int ch = ((ks * 4 + lg) ^ swz) * 8;
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = *(const u32x4*)&As[(wm + i * 16 + li) * TK + ch];
#pragma unroll
      for (int j = 0; j < NJ; ++j) bf[j] = *(const u32x4*)&Bs[(wn + j * 16 + li) * TK + ch];
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) Mma<bf16_t>::mma(acc[i][j], af[i], bf[j]);
    }
    __syncthreads();
  }

  // epilogue: + bias, transpose through LDS (the operand tiles are dead: all fragment reads are behind the barrier
  // above) so that every row leaves in whole 16-byte pieces of contiguous columns
  const float* bias = (const float*)d.gp[g].bias;
  const bool relu = d.act == PQ3D_ACT_RELU;
  if constexpr (!F32OUT) {
    bf16_t* Ct = As;   // [TMT][LDCT] bf16
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = wn + j * 16 + li;
      const float bn = bias ? bias[n0 + col] : 0.f;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[i][j][r] * d.alpha + bn;
          v = relu ? fmaxf(v, 0.f) : v;
          const bf16_t hi = f2bf(v);
          Ct[(wm + i * 16 + 4 * lg + r) * LDCT + col] = hi;
          if constexpr (OUT == 2) acc[i][j][r] = v - bf2f(hi);   // the residual, rounded and stored by the second pass
        }
    }
    __syncthreads();
    bf16_t* C = (bf16_t*)d.gp[g].C;
    constexpr int TPR = TNT / 8, RPP = 256 / TPR;   // threads per row (16-byte pieces), rows per pass
    const int crow = tid / TPR, cch = (tid % TPR) * 8;
#pragma unroll
    for (int p = 0; p < TMT / RPP; ++p) {
      const int row = p * RPP + crow;
      if (m0 + row < d.M) *(u32x4*)(C + (long)(m0 + row) * d.ldc + n0 + cch) = *(const u32x4*)&Ct[row * LDCT + cch];
    }
    if constexpr (OUT == 2) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int col = wn + j * 16 + li;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) Ct[(wm + i * 16 + 4 * lg + r) * LDCT + col] = f2bf(acc[i][j][r]);
      }
      __syncthreads();
      bf16_t* C2 = (bf16_t*)d.gp[g].C2;
#pragma unroll
      for (int p = 0; p < TMT / RPP; ++p) {
        const int row = p * RPP + crow;
        if (m0 + row < d.M) *(u32x4*)(C2 + (long)(m0 + row) * d.ldc + n0 + cch) = *(const u32x4*)&Ct[row * LDCT + cch];
      }
    }
  } else {
    float* Cf = (float*)As;   // [HR][LDFT] fp32: the rows leave in two halves (one per pair of waves)
    float* C = (float*)d.gp[g].C;
    const float* aux = d.act_grad == PQ3D_ACT_ADD ? (const float*)d.gp[g].aux : nullptr;
    constexpr int TPR = TNT / 4, RPP = 256 / TPR;
    const int crow = tid / TPR, cch = (tid % TPR) * 4;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (wm == h * HR) {   // the two waves that own this half
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int col = wn + j * 16 + li;
          const float bn = bias ? bias[n0 + col] : 0.f;
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float v = acc[i][j][r] * d.alpha + bn;
              Cf[(i * 16 + 4 * lg + r) * LDFT + col] = relu ? fmaxf(v, 0.f) : v;
            }
        }
      }
      __syncthreads();
#pragma unroll
      for (int p = 0; p < HR / RPP; ++p) {
        const int row = p * RPP + crow;
        if (m0 + h * HR + row < d.M) {
          const long off = (long)(m0 + h * HR + row) * d.ldc + n0 + cch;
          float4 v = *(const float4*)&Cf[row * LDFT + cch];
          if (aux) {   // act_grad == ADD: C = product + aux (gradient accumulation), read as whole 16-byte pieces
            const float4 x = *(const float4*)(aux + off);
            v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
          }
          *(float4*)(C + off) = v;
        }
      }
      __syncthreads();
    }
  }
}

// ---- TT variant: weight gradients  C[m][n] (fp32) += alpha * sum_k A[k][m] * B[k][n]  (dW = dY^T X), split-K with
// atomics onto a zeroed / accumulating C, optional fused bias gradient colsum[m] += alpha * sum_k A[k][m].
// Both operands are row-major [K, *] bf16 (k strided): the tiles stay in that orientation in LDS ([k][m], one 16-byte
// write per chunk) and the MFMA fragments come from the transposing LDS read, as in gemm.hip's TT path.
constexpr int LDM = TM + 8;   // padded [k][m] row (bf16 elements)
typedef short v4i16_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16_t lds_v4i16_t;
PQ_DEV u32x4 km_frag128(const bf16_t* tile, int r0, int ks, int li, int lg) {
  const bf16_t* p0 = tile + (ks * 32 + 8 * lg + (li >> 2)) * LDM + r0 + 4 * (li & 3);
  const v4i16_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_t*)p0);
  const v4i16_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_t*)(p0 + 4 * LDM));
  const u32x2 lo = __builtin_bit_cast(u32x2, a), hi = __builtin_bit_cast(u32x2, b);
  return (u32x4){lo.x, lo.y, hi.x, hi.y};
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void gemm_tt128_kernel(const pq3d_kdesc d,
                                                                                                    const int nsplit) {
#ifndef PQ3D_NO_KARG_PIN
  // kernel-argument prefetch (see gemm_fast_kernel): one batch of scalar loads for every descriptor scalar used below
  asm volatile("" ::"s"(d.M), "s"(d.N), "s"(d.K), "s"(d.kconcat), "s"(d.lda), "s"(d.ldb), "s"(d.ldc), "s"(d.alpha), "s"(d.act),
               "s"(d.act_grad), "s"(d.groups), "s"(d.xcd_order));
#endif
  __shared__ __attribute__((aligned(16))) bf16_t As[TK * LDM];
  __shared__ __attribute__((aligned(16))) bf16_t Bs[TK * LDM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const TileIdx ti = tile_index(d.xcd_order);   // XCD-aware: the M/128 x N/128 tiles of one (group, k-split) share an L2
  const int split = ti.z % nsplit, g = ti.z / nsplit, m0 = ti.x * TM, n0 = ti.y * TN;
  const int nkt = d.K / TK, per = (nkt + nsplit - 1) / nsplit;
  const int kt0 = split * per, kt1 = min(nkt, kt0 + per);
  if (kt0 >= kt1) return;
  const bf16_t* A = (const bf16_t*)d.gp[g].A;
  const bf16_t* B = (const bf16_t*)d.gp[g].B;
  // staging: thread -> (k row = tid / 16 of a 16-row pass, 16-byte chunk = tid % 16 of the 128-wide m / n slice)
  const int srow = tid >> 4, sch = (tid & 15) * 8;
  const long astep = 16 * d.lda, bstep = 16 * d.ldb;
  const bf16_t* ap = A + (long)srow * d.lda + m0 + sch;
  const bf16_t* bp = B + (long)srow * d.ldb + n0 + sch;
  u32x4 ra[4], rb[4];
  auto gload = [&](int kt) {
    const bf16_t* a0 = ap + (long)kt * TK * d.lda;
    const bf16_t* b0 = bp + (long)kt * TK * d.ldb;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      ra[p] = *(const u32x4*)(a0 + p * astep);
      rb[p] = *(const u32x4*)(b0 + p * bstep);
    }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float* cs_out = ti.y == 0 ? d.gp[g].colsum : nullptr;   // uniform per block
  // fused bias gradient colsum[m] = sum_k A[k][m]: one extra MFMA per A fragment against an all-ones B fragment (every
  // column of the result holds the column sums; the matrix pipe is idle 85 % of the time here) instead of 64 scalar LDS
  // reads per k slice on two of the four waves -- those waves set the pace of half the workgroups
  const bool do_cs = cs_out != nullptr && wn == 0;
  const u32x4 ones = (u32x4){0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
  f32x4 accb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  gload(kt0);
  for (int kt = kt0; kt < kt1; ++kt) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      *(u32x4*)&As[(p * 16 + srow) * LDM + sch] = ra[p];
      *(u32x4*)&Bs[(p * 16 + srow) * LDM + sch] = rb[p];
    }
    __syncthreads();
    if (kt + 1 < kt1) gload(kt + 1);
#pragma unroll
    for (int ks = 0; ks < TK / 32; ++ks) {
      u32x4 af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = km_frag128(As, wm + i * 16, ks, li, lg);
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = km_frag128(Bs, wn + j * 16, ks, li, lg);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Mma<bf16_t>::mma(acc[i][j], af[i], bf[j]);
      if (do_cs) {   // wave-uniform
#pragma unroll
        for (int i = 0; i < 4; ++i) Mma<bf16_t>::mma(accb[i], af[i], ones);
      }
    }
    __syncthreads();
  }
  if (do_cs && li == 0) {   // C layout: lane (column li, rows 4 lg + r); every column holds the same sums
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) unsafeAtomicAdd(&cs_out[m0 + wm + i * 16 + 4 * lg + r], accb[i][r] * d.alpha);
  }
  float* C = (float*)d.gp[g].C;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* crow = C + (long)(m0 + wm + i * 16 + 4 * lg + r) * d.ldc + n0 + wn + li;
#pragma unroll
      for (int j = 0; j < 4; ++j) unsafeAtomicAdd(crow + j * 16, acc[i][j][r] * d.alpha);
    }
}

}  // namespace

// Weight-gradient products (split-K launches; C already zeroed by pq3d_gemm unless it accumulates).  The split factor
// is this kernel's own: ~1.5 workgroups per CU with at least 4 k-tiles each.
bool pq3d_gemm_tt128_try(const pq3d_gemm_desc& d, const pq3d_kdesc& kd, hipStream_t s) {
  if (d.ct != PQ3D_BF16 || d.dtA != PQ3D_BF16 || d.dtB != PQ3D_BF16 || d.dtC != PQ3D_F32) return false;
  if (!d.transA || !d.transB || d.batch != 1 || d.kconcat > 1 || d.splitk <= 1 || d.act || d.act_grad) return false;
  if (d.M % TM || d.N % TN || d.K % TK || d.K < 8 * TK || d.ldc != d.N) return false;
  if (d.lda % 8 || d.ldb % 8) return false;
  if (d.row_scale || d.row_fill_flag || d.mask_out) return false;
  for (int g = 0; g < d.groups; ++g) {
    if (d.A2[g] || d.B2[g] || d.C2[g] || d.aux[g] || d.row_mask[g] || d.bias[g]) return false;
    if ((((uintptr_t)d.A[g]) | ((uintptr_t)d.B[g])) & 15) return false;
  }
  const long tiles = (long)(d.M / TM) * (d.N / TN) * d.groups;
  const int nkt = d.K / TK;
  int nsplit = (int)((384 + tiles - 1) / tiles);   // measured (tools/probes/dw_bench.py): 1.5 workgroups per CU; more splits lose to the atomics
  if (nsplit > nkt / 4) nsplit = nkt / 4;
  if (nsplit < 1) nsplit = 1;
  if (tiles * nsplit < 256) return false;   // too small to fill the chip: the 64x64 tile spreads it better
  pq3d_kdesc k = kd;
  // the (M/128) x (N/128) tiles of one (group, k-split) read the same 2 operand slabs: on one L2 whenever there are enough
  // of them to matter, co-resident or not (hardware order deals them to 8 L2s -- 5.7 x the bytes at 6 x 6 tiles)
  // The (M/128) x (N/128) tiles of one (group, k-split) read the same 2 operand slabs: on one L2 whenever there are enough of
  // them to matter, co-resident or not (hardware order deals them to 8 L2s: 5.7 x the bytes at 6 x 6 tiles, measured 3507 ->
  // 1912 MB per launch at the shipped stage-2 shape).  2 x 2 tiles (config 2) stay in hardware order, which by the
  // arithmetic of x + 2 y + 4 z mod 8 already keeps one memory's 8 groups of a tile position on one L2 (158 MB; 250 MB
  // when walked in XCD order with the groups as runs -- measured, rejected).
  const long per_plane = (long)(d.M / TM) * (d.N / TN);
  k.xcd_order = xcd_order_for(per_plane >= 9 ? PQ3D_XCD_MIN : 0, 1L << 40, 1);
  hipLaunchKernelGGL(gemm_tt128_kernel, dim3(d.M / TM, d.N / TN, d.groups * nsplit), dim3(256), 0, s, k, nsplit);
  return true;
}

// Eligibility is decided here so that pq3d_gemm stays the single entry point (gemm.hip calls this first).
bool pq3d_gemm_nt128_try(const pq3d_gemm_desc& d, const pq3d_kdesc& kd, hipStream_t s) {
  if (d.ct != PQ3D_BF16 || d.dtA != PQ3D_BF16 || d.dtB != PQ3D_BF16) return false;
  if (d.dtC != PQ3D_BF16 && d.dtC != PQ3D_F32) return false;
  const int kc = d.kconcat > 0 ? d.kconcat : 1;
  if (d.transA || d.transB || d.batch != 1 || d.splitk > 1 || (d.act != PQ3D_ACT_NONE && d.act != PQ3D_ACT_RELU)) return false;
  const bool add = d.act_grad == PQ3D_ACT_ADD;   // "+ aux" epilogue: fp32 output and fp32 aux only
  const bool planes = d.act_grad == PQ3D_ACT_PLANES;   // bf16 hi / lo planes of the result: C and C2 both bf16
  if (planes && !(d.dtC == PQ3D_BF16 && d.dtC2 == PQ3D_BF16)) return false;
  if (d.act_grad && !planes && !(add && d.dtC == PQ3D_F32 && d.dtAux == PQ3D_F32 && d.act == PQ3D_ACT_NONE)) return false;
  if (d.M < TM || d.N % TN || d.K % TK || d.K < TK) return false;
  if (d.lda % 8 || d.ldb % 8 || d.ldc % 8) return false;
  if ((long)d.M * d.lda >= (1L << 31) || (long)d.N * d.ldb >= (1L << 31)) return false;
  if (d.row_scale || d.row_fill_flag || d.mask_out || (d.drop.p > 0.f && d.drop.seed)) return false;
  for (int g = 0; g < d.groups; ++g) {
    if (d.A2[g] || d.B2[g] || d.row_mask[g] || d.colsum[g]) return false;
    if (planes ? ((g % kc) == 0 && (!d.C2[g] || (((uintptr_t)d.C2[g]) & 15))) : d.C2[g] != nullptr) return false;
    if (d.aux[g] && !(add && (g % kc) == 0 && (((uintptr_t)d.aux[g]) & 15) == 0)) return false;
    if (add && (g % kc) == 0 && !d.aux[g]) return false;
    if (d.bias[g] && d.dtBias != PQ3D_F32) return false;
    if ((((uintptr_t)d.A[g]) | ((uintptr_t)d.B[g])) & 15) return false;
    if ((g % kc) == 0 && (((uintptr_t)d.C[g]) & 15)) return false;
  }
  // worth it only when the launch still fills the chip: >= 2 workgroups per CU, or >= 1 per CU with a long K loop
  const long tiles = (long)((d.M + TM - 1) / TM) * (d.N / TN) * (d.groups / kc);
  const long nkt = (long)(d.K / TK) * kc;
  if (!planes && tiles < 512 && !(tiles >= 256 && nkt >= 16)) return false;   // (the plane epilogue exists only here: any size)   // (128 tiles x 48 k-tiles measured equal to the 64x64 tile)
  pq3d_kdesc k = kd;
  // Groups that read one row operand (one memory's tokens against the K / V weights of every layer) become one run of
  // z-planes (common.h): the group order of a plain launch carries no meaning, so sort by the A pointer first.
  int run = 1;
  if (kc == 1 && d.groups > 1) {
    std::stable_sort(k.gp, k.gp + d.groups, [](const pq3d_kgroup& a, const pq3d_kgroup& b) { return (uintptr_t)a.A < (uintptr_t)b.A; });
    const void* ap[PQ3D_MAX_GROUPS];
    for (int g = 0; g < d.groups; ++g) ap[g] = k.gp[g].A;
    run = uniform_run(ap, d.groups);
  }
  if (tiles < 768) {   // fewer than 3 workgroups per CU: 128 x 64 tiles (see the kernel's header)
    const dim3 grid((d.M + TM - 1) / TM, d.N / 64, d.groups / kc);
    k.xcd_order = xcd_order_for(2 * tiles, (long)d.N * d.K * 2 * kc, run);
    if (d.dtC == PQ3D_F32) hipLaunchKernelGGL((gemm_nt128_kernel<1, 128, 64>), grid, dim3(256), 0, s, k);
    else if (planes) hipLaunchKernelGGL((gemm_nt128_kernel<2, 128, 64>), grid, dim3(256), 0, s, k);
    else hipLaunchKernelGGL((gemm_nt128_kernel<0, 128, 64>), grid, dim3(256), 0, s, k);
    return true;
  }
  const dim3 grid((d.M + TM - 1) / TM, d.N / TN, d.groups / kc);
  k.xcd_order = xcd_order_for(tiles, (long)d.N * d.K * 2 * kc, run);
  if (d.dtC == PQ3D_F32) hipLaunchKernelGGL((gemm_nt128_kernel<1, 128, 128>), grid, dim3(256), 0, s, k);
  else if (planes) hipLaunchKernelGGL((gemm_nt128_kernel<2, 128, 128>), grid, dim3(256), 0, s, k);
  else hipLaunchKernelGGL((gemm_nt128_kernel<0, 128, 128>), grid, dim3(256), 0, s, k);
  return true;
}
