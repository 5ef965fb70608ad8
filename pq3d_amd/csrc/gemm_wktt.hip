// Weight-gradient products of the query side, dW[N_out, K_in] += g^T (x [+ x2]) over R = B*N_q rows, with whole-K-chunk
// tiles: the TT layout (both operands k-major: A = g [R][N_out], B = x [R][K_in]) of pq3d_gemm for reductions of a few
// hundred rows.  The 64x64-tile pipeline kernel walks such a reduction in 64-row steps -- 800 rows = 13 dependent
// (load -> convert -> LDS -> barrier -> MFMA -> barrier) steps split over 2-4 workgroups: the five flush launches of a config-2
// backward took 112 us for 10.9 GFLOP.  Here a workgroup of 8 waves stages 256 reduction rows of both operands at once
// (all loads of the chunk in flight together, converted to bf16 once, [k][m] / [k][n] tiles read back with the transposing
// LDS read) and issues the chunk's MFMAs back to back; the next chunk's loads fly during them.  Split-K with atomics into
// the pre-zeroed gradient arena (its own split factor: one round of workgroups), fused bias gradient (column sums of g on
// the matrix pipe, as gemm_fast_kernel).  Single-bf16 operands (every backward product of the path), fp32 accumulate.
#include <atomic>
#include <cstdlib>

#include "gemm_common.h"

namespace {

constexpr int WT = 512;

// TT = tile edge (64 or 128: 128-wide tiles halve the operand re-reads from L2, which bound these products at 64 -- each
// g tile is re-read by every n tile and each x tile by every m tile), KC = reduction rows staged per chunk (256 / 128)
template <typename TA, typename TB, bool HB2, int KC, int TT>
__global__ __launch_bounds__(WT) void gemm_wktt_kernel(const pq3d_kdesc d, const int sk) {
  constexpr int TM = TT, TN = TT, LDT = TT + 8, CPK = TT / 8;   // [k][TT + 8] bf16 tiles of A and B; 8-element chunks per k row
  constexpr int MI = TT / 64, NJ = TT / 32;                     // 16x16 MFMA blocks per wave: 4 x 2 waves of (16 MI) x (16 NJ)
  extern __shared__ __attribute__((aligned(16))) unsigned char tt_smem[];
  bf16_t* const At = (bf16_t*)tt_smem;        // [KC][LDT]: A(k, m0 + col)
  bf16_t* const Bt = At + KC * LDT;           // [KC][LDT]: B(k, n0 + col)
  constexpr int NCH = KC * CPK / WT;          // 8-element chunks per operand per thread (4)
  const int split = blockIdx.z % sk, g = blockIdx.z / sk;
  const pq3d_kgroup& q = d.gp[min(g, PQ3D_MAX_GROUPS - 1)];
  const TA* A = (const TA*)q.A;
  const TB* B = (const TB*)q.B;
  const float* B2 = (const float*)q.B2;
  float* C = (float*)q.C;
  // the tiles one XCD receives are neighbours in the (m, n) plane (common.h tile_index_plane): in hardware order the n-tiles
  // of one m-tile sit gridDim.x ids apart -- on other XCDs, much later -- and every one of them fetches the g slab again
  // (caption LM head, dW [32128 x 512] over 512 rows: 8 n-tiles, 535-643 MB fetched per launch for 66 MB of g)
  const TileIdx ti = tile_index_plane(d.xcd_order);
  float* cs_out = ti.y == 0 ? q.colsum : nullptr;
  asm volatile("" ::"s"(d.M), "s"(d.N), "s"(d.K), "s"(d.lda), "s"(d.ldb), "s"(d.ldc), "s"(d.alpha), "s"(A), "s"(B), "s"(B2), "s"(C), "s"(cs_out));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const int wm = (wave >> 1) * (16 * MI), wn = (wave & 1) * (16 * NJ);   // 4 x 2 waves
  const int m0 = ti.x * TM, n0 = ti.y * TN;
  const int nck = (d.K + KC - 1) / KC, per = (nck + sk - 1) / sk;
  const int c0 = split * per, c1 = min(nck, c0 + per);
  if (c0 >= c1) return;
  f32x4 acc[MI][NJ], accb[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const bool do_cs = cs_out != nullptr && wn == 0;
  Raw<TA, 8> ra[NCH];
  Raw<TB, 8> rb[NCH];
  Raw<float, 8> rb2[HB2 ? NCH : 1];
  bool ok[NCH];
  const float s2 = B2 ? 1.f : 0.f;
  auto issue = [&](int ck) {
    const int k0 = ck * KC;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + i * WT, k = k0 + c / CPK, x = (c % CPK) * 8;
      ok[i] = k < d.K;
      const long ka = ok[i] ? k : 0;
      const long offa = ka * d.lda + min(m0 + x, d.M - 8), offb = ka * d.ldb + min(n0 + x, d.N - 8);
      ra[i].load(A + offa);
      rb[i].load(B + offb);
      if constexpr (HB2) rb2[i].load(B2 ? B2 + offb : (const float*)B + offb);
    }
  };
  auto put = [&]() {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + i * WT, o = (c / CPK) * LDT + (c % CPK) * 8;
      u32x4 pa, pb;
      if constexpr (sizeof(TA) == 2) pa = __builtin_bit_cast(u32x4, ra[i]);
      else { float v[8]; ra[i].to_float(v); pa = pack_frag<bf16_t>(v); }
      if constexpr (sizeof(TB) == 2 && !HB2) pb = __builtin_bit_cast(u32x4, rb[i]);
      else {
        float v[8];
        rb[i].to_float(v);
        if constexpr (HB2) {
          float w[8];
          rb2[i].to_float(w);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += s2 * w[j];
        }
        pb = pack_frag<bf16_t>(v);
      }
      if (!ok[i]) { pa = (u32x4){0, 0, 0, 0}; pb = (u32x4){0, 0, 0, 0}; }
      *(u32x4*)&At[o] = pa;
      *(u32x4*)&Bt[o] = pb;
    }
  };
  issue(c0);
  for (int ck = c0; ck < c1; ++ck) {
    if (ck > c0) __syncthreads();
    put();
    __syncthreads();
    const int nks = (min(KC, d.K - ck * KC) + 31) >> 5;
    if (ck + 1 < c1) issue(ck + 1);
#pragma unroll
    for (int ks = 0; ks < KC / 32; ++ks) {
      if (ks < nks) {   // uniform
        u32x4 a[MI], bfr[NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i) a[i] = km_frag(At, LDT, wm + i * 16, ks, li, lg);
#pragma unroll
        for (int j = 0; j < NJ; ++j) bfr[j] = km_frag(Bt, LDT, wn + j * 16, ks, li, lg);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) Mma<bf16_t>::mma(acc[i][j], a[i], bfr[j]);
        if (do_cs) {
          const u32x4 ones = (u32x4){0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
#pragma unroll
          for (int i = 0; i < MI; ++i) Mma<bf16_t>::mma(accb[i], a[i], ones);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = n0 + wn + j * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm + i * 16 + lg * 4 + r;
        // (a plain read-modify-write for unsplit launches -- one writer per element -- was measured SLOWER than the L2 atomics:
        // 62 vs 59 us average at config 5: the returning loads stall the epilogue, the atomics are fire-and-forget)
        if (row < d.M && col < d.N) unsafeAtomicAdd(C + (long)row * d.ldc + col, acc[i][j][r] * d.alpha);
      }
    }
  if (do_cs && li == 0) {   // every column of accb holds the row sums of the A tile over k
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm + i * 16 + lg * 4 + r;
        if (row < d.M) unsafeAtomicAdd(&cs_out[row], accb[i][r] * d.alpha);
      }
  }
}

bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

template <typename TA, typename TB, bool HB2, int KC, int TT>
int tt_launch_t(const pq3d_gemm_desc& d, const pq3d_kdesc& kd, hipStream_t s) {
  auto kern = gemm_wktt_kernel<TA, TB, HB2, KC, TT>;
  constexpr size_t lds = 2 * (size_t)KC * (TT + 8) * sizeof(bf16_t);
  static std::atomic<unsigned> done{0};
  if (int e = pq3d_enable_big_lds(kern, (int)lds, done)) return e;
  // split factor of this kernel: as many KC-row chunks side by side as keep the launch within one round of workgroups
  // (two ~70 KB workgroups per CU)
  const long tiles = (long)((d.M + TT - 1) / TT) * ((d.N + TT - 1) / TT) * d.groups;
  const int nck = (d.K + KC - 1) / KC;
  int sk = (int)(512 / (tiles > 0 ? tiles : 1));
  if (sk < 1) sk = 1;
  if (sk > nck) sk = nck;
  const dim3 grid((d.M + TT - 1) / TT, (d.N + TT - 1) / TT, d.groups * sk);
  pq3d_kdesc k2 = kd;
  const long kl = ((long)(nck + sk - 1) / sk) * KC;   // reduction rows of one workgroup
  k2.xcd_order = plane_xcd_order((int)grid.x, (int)grid.y, (long)TT * kl * (long)sizeof(TA), (long)TT * kl * (long)(sizeof(TB) + (HB2 ? 4 : 0)));
  hipLaunchKernelGGL(kern, grid, dim3(WT), lds, s, k2, sk);
  return 0;
}
template <typename TA, typename TB, bool HB2>
int tt_launch(const pq3d_gemm_desc& d, const pq3d_kdesc& kd, hipStream_t s) {
  // 64 x 64 tiles, 256 reduction rows per chunk.  (128 x 128 tiles over 128-row chunks -- the same 70 KB of LDS, half the
  // operand re-reads from L2 -- were measured slower at config 2: flush launches 25-33 us against 15-25, one 178-VGPR
  // workgroup per CU and four times the atomics per tile; the instantiation is not built any more)
  return tt_launch_t<TA, TB, HB2, 256, 64>(d, kd, s);
}

}  // namespace

// Returns true when the whole-K weight-gradient kernel took the launch (*err = 0 or a hipError_t), false when the call is
// outside its domain.  The caller has zero-filled C (or accumulates on purpose).
bool pq3d_gemm_wktt_try(const pq3d_gemm_desc& d, const pq3d_kdesc& kd, hipStream_t s, int* err) {
  *err = 0;
  if (!(d.transA && d.transB) || d.ct != PQ3D_BF16 || d.batch != 1 || d.kconcat > 1 || d.splitk < 2 || d.dtC != PQ3D_F32) return false;
  if (d.M < 8 || d.N < 8 || d.M % 8 || d.N % 8 || d.lda % 8 || d.ldb % 8 || d.K < 1) return false;
  if (d.ldc != d.N) return false;
  // this kernel has no epilogue beyond alpha and the fused column sums: anything else belongs to the pipeline kernel
  if (d.act || d.act_grad || d.row_scale || d.row_fill_flag || d.mask_out || (d.drop.p > 0.f && d.drop.seed)) return false;
  for (int g = 0; g < d.groups; ++g)
    if (d.bias[g] || d.aux[g] || d.C2[g] || d.row_mask[g]) return false;
  // long reductions over bf16 x bf16 operands (the K/V projections' weight gradients) belong to the 128x128-tile kernel
  if (d.dtA == PQ3D_BF16 && d.dtB == PQ3D_BF16 && d.K >= 2048) return false;
  bool b2 = false;
  for (int g = 0; g < d.groups; ++g) {
    if (!d.A[g] || !d.B[g] || !d.C[g] || !al16(d.A[g]) || !al16(d.B[g]) || d.A2[g]) return false;
    if (d.B2[g]) { b2 = true; if (!al16(d.B2[g]) || d.dtB2 != PQ3D_F32 || d.dtB != PQ3D_F32) return false; }
  }
  int e;
  const bool af = d.dtA == PQ3D_F32, bf = d.dtB == PQ3D_F32;
  if (b2) e = af ? tt_launch<float, float, true>(d, kd, s) : tt_launch<bf16_t, float, true>(d, kd, s);
  else if (af && bf) e = tt_launch<float, float, false>(d, kd, s);
  else if (af) e = tt_launch<float, bf16_t, false>(d, kd, s);
  else if (bf) e = tt_launch<bf16_t, float, false>(d, kd, s);
  else e = tt_launch<bf16_t, bf16_t, false>(d, kd, s);
  if (e) { pq3d_set_error(hipGetErrorString((hipError_t)e)); *err = e; }
  return true;
}
