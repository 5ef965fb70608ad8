// 128x128-tile bf16 NT GEMM for the big streaming projections (hoisted K/V projections, mask-head keys):
//   C_g[m][n] (bf16) = sum_k A_g[m][k] * B_g[n][k] + bias_g[n],   A, B bf16 with k contiguous, K % 64 == 0, N % 128 == 0.
//
// Why a second tile shape: at K = 256 a 64x64 tile loads 64 KB of operands for 8 KB of output (8 B of L2 traffic per
// output byte -- 786 MB for the c2 hoisted projection, which is L2-bandwidth time, not MFMA time); 128x128 halves that.
// Only the plain case lives here (no prologue adds, no activation / masks / split-K) so the register budget goes to the
// 4x4 accumulator block per wave (64 VGPRs) instead of options; everything else stays in gemm.hip.  The k order (one
// accumulator per output, 32-wide MFMA steps in sequence) is gemm.hip's, so both kernels produce the same bits.
// Bound: L2 -> LDS traffic / HBM write of C; algorithmic bytes per group: (M + N) * K * 2 + M * N * 2.
#include "common.h"

namespace {

constexpr int TM = 128, TN = 128, TK = 64, LDT = TK + 8;   // padded LDS row: 144 B -> conflict-free 16-byte fragment reads
constexpr int LDC = TN + 8;                                 // bf16 C staging row

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void gemm_nt128_kernel(const pq3d_gemm_desc d) {
  __shared__ __attribute__((aligned(16))) bf16_t As[TM * LDT];
  __shared__ __attribute__((aligned(16))) bf16_t Bs[TN * LDT];
  static_assert(sizeof(bf16_t) * TM * LDC <= sizeof(bf16_t) * (TM + TN) * LDT, "C staging must fit in the operand tiles");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int g = blockIdx.z, m0 = blockIdx.x * TM, n0 = blockIdx.y * TN;
  const bf16_t* A = (const bf16_t*)d.A[g];
  const bf16_t* B = (const bf16_t*)d.B[g];
  const int nkt = d.K / TK;

  // staging: thread -> (row = tid / 8 of a 32-row pass, 16-byte chunk = tid % 8 of the 64-wide k slice)
  const int srow = tid >> 3, sch = (tid & 7) * 8;
  // 32-bit element offsets (eligibility bounds M * lda and N * ldb below 2^31): 5 registers instead of 16 for pointers
  int aoff[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) aoff[p] = min(m0 + p * 32 + srow, d.M - 1) * (int)d.lda + sch;   // rows past M: clamped
  const int boff = (n0 + srow) * (int)d.ldb + sch, bstep = 32 * (int)d.ldb;
  u32x4 ra[4], rb[4];
  auto gload = [&](int kt) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      ra[p] = *(const u32x4*)(A + aoff[p] + kt * TK);
      rb[p] = *(const u32x4*)(B + boff + p * bstep + kt * TK);
    }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  gload(0);
  for (int kt = 0; kt < nkt; ++kt) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      *(u32x4*)&As[(p * 32 + srow) * LDT + sch] = ra[p];
      *(u32x4*)&Bs[(p * 32 + srow) * LDT + sch] = rb[p];
    }
    __syncthreads();
    if (kt + 1 < nkt) gload(kt + 1);   // next k slice in flight behind the MFMAs
#pragma unroll
    for (int ks = 0; ks < TK / 32; ++ks) {
      u32x4 af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *(const u32x4*)&As[(wm + i * 16 + li) * LDT + ks * 32 + lg * 8];
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = *(const u32x4*)&Bs[(wn + j * 16 + li) * LDT + ks * 32 + lg * 8];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Mma<bf16_t>::mma(acc[i][j], af[i], bf[j]);
    }
    __syncthreads();
  }

  // epilogue: + bias, round to bf16, transpose through LDS so that every row leaves as 256 contiguous bytes
  bf16_t* Ct = As;   // [TM][LDC] over both operand tiles (all fragment reads are behind the barrier above)
  const float* bias = (const float*)d.bias[g];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = wn + j * 16 + li;
    const float bn = bias ? bias[n0 + col] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) Ct[(wm + i * 16 + 4 * lg + r) * LDC + col] = f2bf(acc[i][j][r] * d.alpha + bn);
  }
  __syncthreads();
  bf16_t* C = (bf16_t*)d.C[g];
  const int crow = tid >> 4, cch = (tid & 15) * 8;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int row = p * 16 + crow;
    if (m0 + row < d.M) *(u32x4*)(C + (long)(m0 + row) * d.ldc + n0 + cch) = *(const u32x4*)&Ct[row * LDC + cch];
  }
}

}  // namespace

// Eligibility is decided here so that pq3d_gemm stays the single entry point (gemm.hip calls this first).
bool pq3d_gemm_nt128_try(const pq3d_gemm_desc& d, hipStream_t s) {
  if (d.ct != PQ3D_BF16 || d.dtA != PQ3D_BF16 || d.dtB != PQ3D_BF16 || d.dtC != PQ3D_BF16) return false;
  if (d.transA || d.transB || d.batch != 1 || d.kconcat > 1 || d.splitk > 1 || d.act || d.act_grad) return false;
  if (d.M < TM || d.N % TN || d.K % TK || d.K < TK) return false;
  if (d.lda % 8 || d.ldb % 8 || d.ldc % 8) return false;
  if ((long)d.M * d.lda >= (1L << 31) || (long)d.N * d.ldb >= (1L << 31)) return false;
  if (d.row_scale || d.row_fill_flag || d.mask_out || (d.drop.p > 0.f && d.drop.seed)) return false;
  for (int g = 0; g < d.groups; ++g) {
    if (d.A2[g] || d.B2[g] || d.C2[g] || d.aux[g] || d.row_mask[g] || d.colsum[g]) return false;
    if (d.bias[g] && d.dtBias != PQ3D_F32) return false;
    if ((((uintptr_t)d.A[g]) | ((uintptr_t)d.B[g]) | ((uintptr_t)d.C[g])) & 15) return false;
  }
  // worth it only when the launch still fills the chip: at least ~2 workgroups per CU
  const long tiles = (long)((d.M + TM - 1) / TM) * (d.N / TN) * d.groups;
  if (tiles < 512) return false;
  hipLaunchKernelGGL(gemm_nt128_kernel, dim3((d.M + TM - 1) / TM, d.N / TN, d.groups), dim3(256), 0, s, d);
  return true;
}
