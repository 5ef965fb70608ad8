// Split-bf16 product of PRE-SPLIT operands, result as bf16 hi / lo planes -- the hoisted key/value projection of compute mode
// 'bf16x3' (reference arithmetic: the k / v rows of nn.MultiheadAttention's in_proj, query_encoder.py:268-270, 288-307):
//     v[m][n]  = sum_k (A_hi + A_lo)[m][k] (B_hi + B_lo)[n][k] + bias[n]      (A_lo B_lo dropped: 2^-18 relative)
//     C[m][n]  = bf16(v),   C2[m][n] = bf16(v - C)
// pq3d_gemm takes it as ct = PQ3D_BF16X3, dtA = dtB = PQ3D_BF16 with A2 / B2 = the residual planes and act_grad = PQ3D_ACT_PLANES.
// Round 6 first ran it as three K-concatenated bf16 groups on gemm_nt128_kernel (lo.hi | hi.lo | hi.hi): 6 operand tiles streamed
// and 3 x the fragment reads per 3 MFMAs, and 30 groups per launch = three launches at config 2 (174 us against 44.5 us for the
// single-bf16 projection).  Here a workgroup stages the FOUR planes of a k slice once and every fragment feeds 3 (A) / 3 (B)
// MFMAs: 4 tiles and 16 fragment reads per 48 MFMAs, one launch for all (layer, memory, k|v) groups.
// Tile 128 x 128, k slices of 32 (32 KB of LDS: 4 workgroups per CU cover the single-buffer barriers), LDS-DMA staging with the
// bank swizzle in the source address (64-byte rows: slot s of row r holds chunk s ^ ((r >> 2) & 3)), XCD-aware tile order.
// Round 6, last session: the MFMA operands are swapped (C^T tiles: a lane holds 4 consecutive columns of one row), so the epilogue
// stages 8 bytes per LDS write instead of 2: 107 -> 101.6 us at config 2, 607 -> 595 us for config 5's large launch.  Isolating
// switches (profiles/NOTES_r06.md section 5): skeleton 30 + MFMAs 34 + operand DMA 30 + stores 24 us -- the parts add up.
// Bound: the write of C + C2 (an fp32 tensor's bytes) / L2 -> LDS operand traffic; algorithmic bytes per group
// (M + N) K 4 + M N 4.
#include <algorithm>

#include "common.h"

namespace {

constexpr int XM = 128, XN = 128, XTK = 32;
constexpr int XLDC = XN + 8;   // bf16 C staging row

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void gemm_x3p128_kernel(const pq3d_kdesc d) {
#ifndef PQ3D_NO_KARG_PIN
  asm volatile("" ::"s"(d.M), "s"(d.N), "s"(d.K), "s"(d.lda), "s"(d.ldb), "s"(d.ldc), "s"(d.alpha), "s"(d.groups), "s"(d.xcd_order));
#endif
  constexpr int OPB = 4 * XM * XTK * 2, STB = XM * XLDC * 2;
  __shared__ __attribute__((aligned(16))) bf16_t sm[(OPB > STB ? OPB : STB) / 2];
  bf16_t* const Ah = sm;
  bf16_t* const Al = Ah + XM * XTK;
  bf16_t* const Bh = Al + XM * XTK;
  bf16_t* const Bl = Bh + XN * XTK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const TileIdx ti = tile_index(d.xcd_order);
  const int g = ti.z, m0 = ti.x * XM, n0 = ti.y * XN;
  const bf16_t* const A = (const bf16_t*)d.gp[g].A;
  const bf16_t* const A2 = (const bf16_t*)d.gp[g].A2;
  const bf16_t* const B = (const bf16_t*)d.gp[g].B;
  const bf16_t* const B2 = (const bf16_t*)d.gp[g].B2;
  const int nkt = d.K / XTK;

  // DMA piece p of this thread: LDS slot (p * 256 + tid) = row slot / 4, position slot % 4 <- k chunk pos ^ ((row >> 2) & 3)
  int aoff[2], boff[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int slot = p * 256 + tid, row = slot >> 2, c = (slot & 3) ^ ((row >> 2) & 3);
    aoff[p] = min(m0 + row, d.M - 1) * (int)d.lda + c * 8;   // rows past M: clamped duplicates (never stored)
    boff[p] = (n0 + row) * (int)d.ldb + c * 8;
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int ch = (lg ^ ((li >> 2) & 3)) * 8;   // this lane's chunk slot inside its fragment rows (row = 16 t + li: (row >> 2) & 3 = li >> 2)

  for (int kt = 0; kt < nkt; ++kt) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int lo_ = (p * 256 + wave * 64) * 8;
      __builtin_amdgcn_global_load_lds((gptr_t*)(A + aoff[p] + kt * XTK), (lptr_t*)(Ah + lo_), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t*)(A2 + aoff[p] + kt * XTK), (lptr_t*)(Al + lo_), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t*)(B + boff[p] + kt * XTK), (lptr_t*)(Bh + lo_), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t*)(B2 + boff[p] + kt * XTK), (lptr_t*)(Bl + lo_), 16, 0, 0);
    }
    __syncthreads();   // the compiler drains the DMA queue (vmcnt(0)) before the barrier
    u32x4 bh[4], bl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bh[j] = *(const u32x4*)&Bh[(wn + j * 16 + li) * XTK + ch];
      bl[j] = *(const u32x4*)&Bl[(wn + j * 16 + li) * XTK + ch];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const u32x4 ah = *(const u32x4*)&Ah[(wm + i * 16 + li) * XTK + ch];
      const u32x4 al = *(const u32x4*)&Al[(wm + i * 16 + li) * XTK + ch];
#pragma unroll
      for (int j = 0; j < 4; ++j) {   // operands swapped: the lane holds C[m = li][n = 4 lg + r] -- 4 consecutive columns of one row
        Mma<bf16_t>::mma(acc[i][j], bh[j], al);
        Mma<bf16_t>::mma(acc[i][j], bl[j], ah);
        Mma<bf16_t>::mma(acc[i][j], bh[j], ah);
      }
    }
    __syncthreads();
  }

  // epilogue: + bias, hi plane then residual plane through one LDS staging tile (rows leave in whole 16-byte pieces); the
  // transposed accumulator puts 4 consecutive columns of a row in a lane: 32 eight-byte staging writes per thread and plane pair
  // where the plain layout needed 128 two-byte ones
  const float* bias = (const float*)d.gp[g].bias;
  bf16_t* const Ct = sm;   // [XM][XLDC]
  constexpr int TPR = XN / 8, RPP = 256 / TPR;
  const int crow = tid / TPR, cch = (tid % TPR) * 8;
#pragma unroll
  for (int pl = 0; pl < 2; ++pl) {
    if (pl) __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = wn + j * 16 + 4 * lg;
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
      if (pl == 0 && bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = bias[n0 + col + r];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bf16_t h[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[i][j][r];
          if (pl == 0) v = v * d.alpha + bv[r];
          h[r] = f2bf(v);
          if (pl == 0) acc[i][j][r] = v - bf2f(h[r]);
        }
        *(u32x2*)&Ct[(wm + i * 16 + li) * XLDC + col] = (u32x2){(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16)};
      }
    }
    __syncthreads();
    bf16_t* const C = (bf16_t*)(pl == 0 ? d.gp[g].C : d.gp[g].C2);
#pragma unroll
    for (int p = 0; p < XM / RPP; ++p) {
      const int row = p * RPP + crow;
      if (m0 + row < d.M) *(u32x4*)(C + (long)(m0 + row) * d.ldc + n0 + cch) = *(const u32x4*)&Ct[row * XLDC + cch];
    }
  }
}

}  // namespace

// pq3d_gemm routes here (gemm.hip): ct PQ3D_BF16X3 with bf16 A / B, A2 / B2 = residual planes, act_grad PQ3D_ACT_PLANES.
bool pq3d_gemm_x3p_try(const pq3d_gemm_desc& d, const pq3d_kdesc& kd, hipStream_t s) {
  if (d.ct != PQ3D_BF16X3 || d.act_grad != PQ3D_ACT_PLANES) return false;
  if (d.dtA != PQ3D_BF16 || d.dtB != PQ3D_BF16 || d.dtA2 != PQ3D_BF16 || d.dtB2 != PQ3D_BF16 || d.dtC != PQ3D_BF16 || d.dtC2 != PQ3D_BF16) return false;
  if (d.transA || d.transB || d.batch != 1 || d.splitk > 1 || d.kconcat > 1 || d.act != PQ3D_ACT_NONE) return false;
  if (d.M < 1 || d.N % XN || d.K % XTK || d.K < XTK) return false;
  if (d.lda % 8 || d.ldb % 8 || d.ldc % 8) return false;
  if ((long)d.M * d.lda >= (1L << 31) || (long)d.N * d.ldb >= (1L << 31)) return false;
  if (d.row_scale || d.row_fill_flag || d.mask_out || (d.drop.p > 0.f && d.drop.seed)) return false;
  for (int g = 0; g < d.groups; ++g) {
    if (!d.A2[g] || !d.B2[g] || !d.C2[g] || d.aux[g] || d.row_mask[g] || d.colsum[g]) return false;
    if (d.bias[g] && d.dtBias != PQ3D_F32) return false;
    if ((((uintptr_t)d.A[g]) | ((uintptr_t)d.A2[g]) | ((uintptr_t)d.B[g]) | ((uintptr_t)d.B2[g]) | ((uintptr_t)d.C[g]) | ((uintptr_t)d.C2[g])) & 15)
      return false;
  }
  pq3d_kdesc k = kd;
  // groups that read one row operand (one memory's tokens against the K / V weights of every layer) become one run of z-planes
  int run = 1;
  if (d.groups > 1) {
    std::stable_sort(k.gp, k.gp + d.groups, [](const pq3d_kgroup& a, const pq3d_kgroup& b) { return (uintptr_t)a.A < (uintptr_t)b.A; });
    const void* ap[PQ3D_MAX_GROUPS];
    for (int g = 0; g < d.groups; ++g) ap[g] = k.gp[g].A;
    run = uniform_run(ap, d.groups);
  }
  const long tiles = (long)((d.M + XM - 1) / XM) * (d.N / XN) * d.groups;
  k.xcd_order = xcd_order_for(tiles, (long)d.N * d.K * 4, run);
  hipLaunchKernelGGL(gemm_x3p128_kernel, dim3((d.M + XM - 1) / XM, d.N / XN, d.groups), dim3(256), 0, s, k);
  return true;
}
