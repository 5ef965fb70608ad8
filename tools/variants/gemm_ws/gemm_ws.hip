// Weight-stationary bf16 GEMM for the hoisted key/value projections of the decoder
//   C_g[m][n] = act( sum_k A_g[m][k] W_g[n][k] + bias_g[n] ),   N = K = 256 (= d_model), M = B * N_seg rows, g < 2 * L * M_mem
// (reference: the K / V halves of nn.MultiheadAttention's in_proj over every memory token of every decoder layer,
// modules/layers/transformers.py:189-193 via query_encoder.py:268-307; this repo hoists them out of the layer loop).
//
// Why another kernel: at K = 256 the product is a STREAM -- every output byte needs one input row read once and one weight
// that all rows share.  The 128 x 128-tile kernel (gemm128.hip) moves 8 bytes of L2 -> LDS traffic per output element
// (each tile re-stages its 64 KB of A and its 64 KB of W) and runs at the L2's rate: 48.8 us at config 2 for 113 MB of
// compulsory HBM bytes.  Here a workgroup parks ONE group's whole weight (256 x 256 bf16 = 128 KB of the CU's 160 KB LDS)
// and streams rows past it:
//   * W sits in LDS once per workgroup, chunk-swizzled (16-byte chunk c of row n at position c ^ s(n)) so that the
//     ds_read_b128 of a fragment (16 rows, one k chunk) touches 16 distinct bank groups without padding;
//   * the activation rows go global -> registers directly in MFMA operand shape (lane = row, 16 bytes = 8 consecutive k),
//     prefetched a few k-steps ahead; no LDS staging, no barrier in the main loop;
//   * the product is formed TRANSPOSED (W fragment as the MFMA's A operand, activation fragment as B), so a lane ends up
//     with 8 consecutive output columns of one row -> one 16-byte bf16 store per (16-row tile, 32-column block);
//   * 8 waves as 4 (rows) x 2 (columns), wave tile 64 x 128 (128 accumulator registers): per 32-wide k-step a wave issues
//     32 MFMAs for 8 LDS fragment reads and 4 global loads (LDS at half its rate when the matrix pipe is saturated);
//   * placement: workgroup id = xcd + 8 * (group + groups * sub) -- the hardware deals ids round-robin to the 8 XCDs, so
//     XCD x owns row slice x of EVERY group: the groups that share an input (one memory's tokens x 8 weights) read it
//     through one L2 at about the same time.
// Same arithmetic as gemm_nt128_kernel / gemm_fast_kernel: one fp32 accumulator per output, 32-wide MFMA steps in k order,
// then * alpha + bias, ReLU, round to bf16 -- the bits are the same (tests/test_gpu_ops.py compares them exactly).
// Bound: HBM (write of C); algorithmic bytes per group: (M + N) * K * 2 + M * N * 2.
#include <atomic>

#include "common.h"

namespace {

constexpr int WS_N = 256, WS_K = 256, WS_TM = 128, WS_T = 512, WS_KS = WS_K / 32;
constexpr int WS_CH = WS_K / 8;                       // 16-byte chunks per weight row
constexpr int WS_STAGE = 4 * 8 * 1024;                // hand-over buffer: 4 compute waves x 8 units x 1 KB (half a wave tile)
constexpr int WS_LDS = WS_N * WS_K * 2 + WS_STAGE;    // = 160 KB, the whole LDS of a CU
constexpr int WS_PF = 2, WS_SLOTS = 4;                // activation fragments in flight (k-steps); the ring carries over blocks
static_assert(WS_KS % WS_SLOTS == 0 && WS_PF < WS_SLOTS, "ring");

PQ_DEV int ws_swz(int n) { return (n & 3) | (((n >> 3) & 3) << 2); }

typedef __attribute__((address_space(3))) const u32x4 lds_frag_t;
typedef __attribute__((address_space(3))) u32x4 lds_wfrag_t;
typedef __attribute__((address_space(3))) unsigned char lds_byte_t;

// Waves 0-3 compute (one per SIMD: loads, LDS fragment reads, MFMAs, bias + rounding), waves 4-7 store.  gfx9's vmcnt counts
// loads AND stores and completes them in issue order, so a wave that keeps loading behind its own stores can never have more
// stores in flight than it issues between a load and the load's first use -- at the ~4-6 us write latency of a saturated chip
// that capped every single-role variant of this kernel (and the 128 x 128-tile kernel) at 2.3-2.7 TB/s of output: 225 us at
// config 5 with loads and stores in one wave, 97 us without the loads, 103 us without the stores.  Here the finished tiles
// cross to the store waves through LDS (bf16, 8 KB per compute wave = half a tile at a time, two workgroup barriers per
// half); a store wave never waits for memory at all.
template <bool RELU>
__global__ __launch_bounds__(WS_T) void gemm_ws_kernel(const pq3d_kdesc d, const int nsub, const int rows_per_slice) {
#ifndef PQ3D_NO_KARG_PIN
  asm volatile("" ::"s"(d.M), "s"(d.ldb), "s"(d.ldc), "s"(d.groups));
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char ws_smem[];
  bf16_t* const Ws = (bf16_t*)ws_smem;                       // [256][32 chunks of 8], swizzled
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  // placement: id % 8 = XCD = row slice (of this sub-range); id / 8 = group + groups * sub
  const int wid = (int)blockIdx.x, xcd = wid & 7, rest = wid >> 3;
  const int g = rest % d.groups, sub = rest / d.groups;
  const int slice = sub * 8 + xcd;
  const int r_lo = slice * rows_per_slice, r_hi = min(d.M, r_lo + rows_per_slice);   // multiples of 128 (host)
  if (r_lo >= r_hi) return;
  const bf16_t* const A = (const bf16_t*)d.gp[g].A;          // fragment-major: [M / 16][8 k-steps][64 lanes][8]
  const bf16_t* const B = (const bf16_t*)d.gp[g].B;
  bf16_t* const C = (bf16_t*)d.gp[g].C;
  const float* const bias = (const float*)d.gp[g].bias;
  const int ldc = (int)d.ldc;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_byte_t*)ws_smem, stage0 = lds0 + WS_N * WS_K * 2;
  const int nblk = (r_hi - r_lo) / WS_TM;

  // ---- the group's weight into LDS (once, all 8 waves), swizzled
  {
    const int ldb = (int)d.ldb;
#pragma unroll
    for (int p = 0; p < WS_N * WS_CH / WS_T; p += 8) {
      u32x4 wv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = tid + (p + q) * WS_T, n = c / WS_CH, kc = c % WS_CH;
        wv[q] = *(const u32x4*)(B + n * ldb + kc * 8);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = tid + (p + q) * WS_T, n = c / WS_CH, kc = c % WS_CH;
        *(u32x4*)&Ws[(n * WS_CH + (kc ^ ws_swz(n))) * 8] = wv[q];
      }
    }
  }
  __syncthreads();

  if (wave >= 4) {
    // ================= store waves: wave 4 + s writes what compute wave s hands over =================
    const int cw = wave - 4, wmi = cw >> 1, wni = cw & 1;
    // a store instruction covers 8 rows x 128 bytes (the 64 columns of a half): lane -> (row r = lane / 8, chunk c = lane % 8);
    // the unit (ud, up) that holds it: ud = row / 16, up = c / 4, producer lane (row % 16) + 16 (c % 4)
    const int sr = lane >> 3, sc = lane & 7;
    unsigned rd[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int row = q * 8 + sr;
      rd[q] = stage0 + cw * 8192 + ((row >> 4) * 2 + (sc >> 2)) * 1024 + ((row & 15) + 16 * (sc & 3)) * 16;
    }
    for (int blk = 0; blk < nblk; ++blk) {
      const int r0 = r_lo + blk * WS_TM;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        __syncthreads();                      // B1: the half is in LDS
        u32x4 t[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = *(lds_frag_t*)(uintptr_t)rd[q];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                      // B2: the buffer is free again
        bf16_t* cp = C + (long)(r0 + wmi * 64 + sr) * ldc + 128 * wni + 64 * h + 8 * sc;
#pragma unroll
        for (int q = 0; q < 8; ++q) *(u32x4*)(cp + (long)(q * 8) * ldc) = t[q];
      }
    }
    return;
  }

  // ================= compute waves =================
  const int wmi = wave >> 1, wni = wave & 1;
  unsigned aoff[4];       // element offset of this lane's 16 bytes of m-fragment i of the current / next 128-row block
  auto set_rows = [&](int r0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) aoff[i] = (unsigned)(((r0 + wmi * 64 + i * 16) >> 4) * (16 * WS_K) + lane * 8);
  };
  u32x4 af[WS_SLOTS][4];
  set_rows(r_lo);
#pragma unroll
  for (int p = 0; p < WS_PF; ++p)
#pragma unroll
    for (int i = 0; i < 4; ++i) af[p][i] = *(const u32x4*)(A + aoff[i] + 512 * p);

  // weight fragment j of this wave (MFMA A operand: lane li = output column index inside the tile, lg = k chunk):
  // column n(j, li) = 128 wni + 32 (j >> 1) + 8 (li >> 2) + 4 (j & 1) + (li & 3)  ->  swizzle key s(n) = li, chunk position
  // (4 ks + lg) ^ li = ((4 (ks & 3)) ^ (li & 12)) + (lg ^ (li & 3)) + 16 (ks >> 2).  Byte address = one of FOUR per-lane
  // bases (ks & 3) + a compile-time offset (j, ks >> 2) that fits the DS instruction's 16-bit offset field.  (LDS addresses
  // as integers: an address laundered as a generic pointer turns the reads into flat loads, which count in vmcnt too.)
  unsigned wb[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
    wb[q] = lds0 + (128 * wni + 8 * (li >> 2) + (li & 3)) * (WS_CH * 16) + (((4 * q) ^ (li & 12)) + (lg ^ (li & 3))) * 16;
  auto wfrag = [&](int ks, int j) {
    return *(lds_frag_t*)(uintptr_t)(wb[ks & 3] + (32 * (j >> 1) + 4 * (j & 1)) * (WS_CH * 16) + (ks >> 2) * 256);
  };
  constexpr int WF = 4;    // weight fragments in flight: the one WF positions ahead is requested right behind a fragment's MFMAs
  u32x4 wf[WF];
#pragma unroll
  for (int j = 0; j < WF; ++j) wf[j] = wfrag(0, j);
  const unsigned st = stage0 + wave * 8192 + lane * 16;   // this lane's 16 bytes of unit 0 in the hand-over buffer
  const float* const bl = bias ? bias + 128 * wni + 8 * lg : nullptr;

  for (int blk = 0; blk < nblk; ++blk) {
    const int r0 = r_lo + blk * WS_TM;
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < WS_KS; ++ks) {
      {   // prefetch k-step ks + WS_PF (the next block's rows once this block's are all requested)
        const int pk = ks + WS_PF;
        if (pk == WS_KS) set_rows(blk + 1 < nblk ? r0 + WS_TM : r0);   // past the last block: harmless re-reads
        const int kk = pk < WS_KS ? pk : pk - WS_KS;
#pragma unroll
        for (int i = 0; i < 4; ++i) af[(ks + WS_PF) % WS_SLOTS][i] = *(const u32x4*)(A + aoff[i] + 512 * kk);
      }
      // opaque bases: the weight fragments do not depend on the block (256 registers' worth): without this the compiler
      // hoists their LDS reads out of the block loop and spills
      asm volatile("" : "+v"(wb[(ks + 1) & 3]));
#pragma unroll
      for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int i = 0; i < 4; ++i) Mma<bf16_t>::mma(acc[i][j], wf[j % WF], af[ks % WS_SLOTS][i]);
        wf[j % WF] = wfrag((ks + (j + WF) / 8) & 7, (j + WF) % 8);
        __builtin_amdgcn_sched_barrier(0);   // keep the request HERE (the scheduler otherwise sinks it next to its use)
      }
    }
    // ---- hand-over: two halves of 64 columns; unit (i, up) of a half = 16 rows x 32 columns, 16 bytes per lane
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      u32x4 o[4][2];
#pragma unroll
      for (int up = 0; up < 2; ++up) {
        const int jp = 2 * h + up;
        float bb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (bl) {
          const float4 b0 = *(const float4*)(bl + 32 * jp), b1 = *(const float4*)(bl + 32 * jp + 4);
          bb[0] = b0.x; bb[1] = b0.y; bb[2] = b0.z; bb[3] = b0.w; bb[4] = b1.x; bb[5] = b1.y; bb[6] = b1.z; bb[7] = b1.w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = acc[i][2 * jp][e] + bb[e];
            v[4 + e] = acc[i][2 * jp + 1][e] + bb[4 + e];
          }
          if (RELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          o[i][up] = (u32x4){pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
        }
      }
      // the buffer is free once the store waves have read the half before this one (B2 of that half)
      if (h == 1 || blk > 0) __syncthreads();
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int up = 0; up < 2; ++up) *(lds_wfrag_t*)(uintptr_t)(st + (i * 2 + up) * 1024) = o[i][up];
      __syncthreads();                        // B1: the half is in LDS
    }
  }
  __syncthreads();                            // B2 of the last half (the store waves' count)
}

int g_ws = 1;   // pq3d_gemm_ws: bit 0 on (A/B measurements); the kernel takes fragment-major activations only

}  // namespace

// Eligibility is decided here so that pq3d_gemm stays the single entry point (gemm.hip calls this before gemm128.hip).
bool pq3d_gemm_ws_try(const pq3d_gemm_desc& d, const pq3d_kdesc& kd, hipStream_t s) {
  if (!(g_ws & 1)) return false;
  if (d.ct != PQ3D_BF16 || d.dtA != PQ3D_BF16 || d.dtB != PQ3D_BF16 || d.dtC != PQ3D_BF16) return false;
  if (d.transA || d.transB || d.batch != 1 || d.splitk > 1 || d.kconcat > 1 || d.act_grad) return false;
  if (d.act != PQ3D_ACT_NONE && d.act != PQ3D_ACT_RELU) return false;
  if (d.N != WS_N || d.K != WS_K || d.lda % 8 || d.ldb % 8 || d.ldc % 8) return false;
  if ((long)d.M * d.lda >= (1L << 31) || (long)d.N * d.ldb >= (1L << 31)) return false;
  if (d.row_scale || d.row_fill_flag || d.mask_out || (d.drop.p > 0.f && d.drop.seed)) return false;
  for (int g = 0; g < d.groups; ++g) {
    if (d.A2[g] || d.B2[g] || d.C2[g] || d.row_mask[g] || d.colsum[g] || d.aux[g]) return false;
    if (d.bias[g] && d.dtBias != PQ3D_F32) return false;
    if ((((uintptr_t)d.A[g]) | ((uintptr_t)d.B[g]) | ((uintptr_t)d.C[g])) & 15) return false;
  }
  // one workgroup per (group, row slice): 8 slices (one per XCD) x nsub; worth it when every workgroup streams at least a
  // few 256-row iterations past its weight and the launch covers most of the chip
  int nsub = 32 / d.groups;
  if (nsub < 1) nsub = 1;
  const int slices = 8 * nsub;
  int rps = (d.M + slices - 1) / slices;
  rps = (rps + WS_TM - 1) / WS_TM * WS_TM;
  if (d.M % WS_TM || d.alpha != 1.f || rps < 2 * WS_TM || (long)d.groups * slices < 128) return false;
  auto launch = [&](auto kern, std::atomic<unsigned>& done) {
    if (pq3d_enable_big_lds(kern, WS_LDS, done)) return false;
    hipLaunchKernelGGL(kern, dim3((unsigned)(d.groups * slices)), dim3(WS_T), WS_LDS, s, kd, nsub, rps);
    return true;
  };
  static std::atomic<unsigned> dn[2] = {{0}, {0}};
  if (!(g_ws & 2)) return false;   // (probe switch until the producers write fragment-major activations)
  return d.act == PQ3D_ACT_RELU ? launch(gemm_ws_kernel<true>, dn[0]) : launch(gemm_ws_kernel<false>, dn[1]);
}

extern "C" int pq3d_gemm_ws(int32_t on) { g_ws = on; return 0; }
