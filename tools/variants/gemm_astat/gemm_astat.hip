// A-stationary bf16 GEMM for the hoisted K/V projections (round 4):  C_g[m][n] = sum_k A_f(g)[m][k] B_g[n][k] + bias_g[n]
// for GROUPS THAT SHARE THEIR A OPERAND -- the layer-invariant key / value inputs of a memory feed the K (or V) projection
// of every decoder layer (CrossAttentionLayer.forward_post, query_encoder.py:288-307: k = memory + pos, v = memory; the
// fused executor hoists the projections of all layers out of the layer loop), K = d = 256.
//
// Why: gemm_nt128_kernel gives every 128 x 128 output tile its own workgroup, which stages the tile's 128 x 256 A rows again
// (the 25 MB of inputs were fetched 4.5x at config 2, PMC round 3) and runs its four k slices as four dependent
// {DMA -> barrier -> MFMA -> barrier} steps, hidden only by 4 workgroups per CU: 14 us per workgroup lifetime for 0.8 us of
// MFMA time.  Here a workgroup owns a 128-row A tile for up to 4 output tiles of the groups that share it:
//   * the A tile lives in REGISTERS as MFMA fragments (64 VGPRs of each of the 4 compute waves, 32 rows x all 128 columns
//     per wave): fetched once per 4 tiles, no LDS space or staging for it (a 2 x 2 wave layout -- 128 VGPRs of fragments,
//     half the LDS reads -- spilled);
//   * the B (weight) tiles stream through a 6-stage LDS ring of 128 x 64 slices filled by direct global -> LDS DMA, five
//     slices ahead of the MFMAs, with COUNTED s_waitcnt vmcnt(N) and raw s_barrier (a __syncthreads would drain the queue);
//   * waves 4-7 only STORE: the compute waves leave a finished tile as bf16 in an LDS staging tile and go on with the next
//     one while the store waves write it out in whole 256-byte rows.  The split is what keeps the counted waits exact: on
//     gfx9 the vector-memory counter counts loads and stores, which complete out of order with respect to each other --
//     a wave that has both in flight cannot wait for "the oldest DMA" by count.  Compute waves have only loads, store waves
//     only stores.
// Same MFMA k order per accumulator (8 steps of 32, ascending) and the same epilogue arithmetic as gemm_nt128_kernel /
// gemm_fast_kernel: identical bits (tests/test_gpu_ops.py).  One workgroup per CU (133 KB of LDS, 512 threads).
#include <atomic>

#include "common.h"

namespace {

constexpr int TM = 128, TN = 128, TK = 64, KD = 256;   // tile; k slice; the reduction length this kernel is built for
constexpr int RING = 6, AHEAD = 5;                     // ring stages; slices requested ahead of the one being multiplied
constexpr int MAXT = 4;                                // output tiles per workgroup
constexpr int SLICE = TN * TK;                         // bf16 elements per ring stage (16 KB)
constexpr int LDCS = TN + 8;                           // bf16 row of the C staging tile
constexpr size_t ASTAT_LDS = (size_t)RING * SLICE * 2 + (size_t)TM * LDCS * 2 + (size_t)MAXT * TN * 4;   // ring, C tile, bias rows

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

struct AstatPlan {
  int NT;                         // 128-column tiles per group (N / 128)
  unsigned char order[PQ3D_MAX_GROUPS];   // groups sorted by A pointer: tile t of the launch = (order[t / NT], t % NT)
  unsigned short ent_start[128];  // first tile of work entry e (entries never straddle two A operands)
  unsigned char ent_n[128];       // tiles of entry e (1 .. MAXT)
};

// s_waitcnt vmcnt(4 * ahead): at most `ahead` whole slices (4 DMA instructions per compute wave each) still in flight
PQ_DEV void wait_slices(int ahead) {
  switch (ahead) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
  }
}

#ifdef PQ3D_ASTAT_TIMELINE   // probe builds (tools/probes/astat_timeline.py): 100 MHz stamps of one wave of a few workgroups
__device__ long g_astat_tl[64 * 16];
#define TL(i) do { if (lane == 0 && (wave == 0 || wave == 4) && blockIdx.x < 4 && blockIdx.y < 8) \
                     g_astat_tl[((blockIdx.y * 4 + blockIdx.x) * 2 + (wave >> 2)) * 16 + (i)] = (long)wall_clock64(); } while (0)
#else
#define TL(i) do { } while (0)
#endif

__global__ __launch_bounds__(512) void gemm_nt_astat_kernel(const pq3d_kdesc d, const AstatPlan pl) {
  extern __shared__ __attribute__((aligned(16))) unsigned char astat_smem[];   // ONE LDS object: ring stages, then the C tile
  bf16_t* const ring = (bf16_t*)astat_smem;
  bf16_t* const Cst = ring + RING * SLICE;
  float* const bias_s = (float*)(Cst + TM * LDCS);   // [MAXT][TN]: the bias rows of this workgroup's tiles
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  TL(0);
  const bool loader = wave < 4;   // waves 0-3 issue the DMA (loads only in their queue), waves 4-7 the global stores
  const int m0 = blockIdx.x * TM;
  const int t0 = pl.ent_start[blockIdx.y], nt = pl.ent_n[blockIdx.y], nq = nt * 4;
  const int NT = pl.NT;
  const int wm = (wave >> 2) * 64, wn = (wave & 3) * 32;   // 2 x 4 waves over the tile: 64 rows x 32 columns each

  // per-tile operands (uniform): B rows / bias / C columns of tile t0 + tt
  const bf16_t* Bp[MAXT];
  const float* biasp[MAXT];
  bf16_t* Cp[MAXT];
#pragma unroll
  for (int tt = 0; tt < MAXT; ++tt) {
    const int t = t0 + min(tt, nt - 1), g = pl.order[t / NT], n0 = (t % NT) * TN;
    Bp[tt] = (const bf16_t*)d.gp[g].B + (long)n0 * d.ldb;
    biasp[tt] = d.gp[g].bias ? (const float*)d.gp[g].bias + n0 : nullptr;
    Cp[tt] = (bf16_t*)d.gp[g].C + n0;
  }
  const bf16_t* A = (const bf16_t*)d.gp[pl.order[t0 / NT]].A;

  // DMA pieces of this thread in a slice (compute waves: a quarter of the 1024 16-byte pieces each): LDS slot p * 256 +
  // wave * 64 + lane = (row = slot / 8, position slot % 8) <- k chunk position ^ swizzle(row) of that row
  int boff[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int slot = p * 256 + (wave & 3) * 64 + lane, row = slot >> 3, c = (slot & 7) ^ ((row >> 1) & 7);
    boff[p] = (row * (int)d.ldb + c * 8) * 2;   // byte offset
  }
  // The DMA instruction is issued from inline asm ON PURPOSE: hipcc (ROCm 7.2) waits vmcnt(0) in front of every LDS access
  // that follows a __builtin_amdgcn_global_load_lds it can see (a pending LDS write it cannot disambiguate), which drains the
  // five-slice queue at the first fragment read of every slice.  Hidden in asm, the queue is counted by hand (wait_slices);
  // the compiler's own counting of the ordinary loads (A fragments, issued AFTER the first DMAs and waited for before the
  // loop) stays correct because the counter retires loads in order.
  const unsigned ring_lds = (unsigned)(uintptr_t)(lptr_t*)ring;
  auto issue = [&](int s, const bf16_t* Bt) {   // slice s of the entry: k slice s % 4 of tile s / 4 into ring stage s % RING
    const unsigned dst = __builtin_amdgcn_readfirstlane(ring_lds + (unsigned)(((s % RING) * SLICE + (wave & 3) * 64 * 8) * 2));
    const bf16_t* src = Bt + (s & 3) * TK;   // uniform: SGPR base + one 32-bit byte offset per piece (saddr form)
#pragma unroll
    for (int p = 0; p < 4; ++p)
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(boff[p]), "s"(src),
                   "s"(dst + (unsigned)(p * 256 * 16))
                   : "memory");
  };

  u32x4 af[4][8];          // A fragments of this wave's 64 rows: [16-row block][k step of 32]  (128 VGPRs)
  f32x4 acc[4][2];
  if (loader) {
#pragma unroll
    for (int s = 0; s < AHEAD; ++s)
      if (s < nq) issue(s, Bp[s >> 2]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      af[i][ks] = *(const u32x4*)(A + (long)(m0 + wm + i * 16 + li) * d.lda + ks * 32 + lg * 8);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (!loader) {   // the bias rows into LDS (visible to everybody behind the first barrier)
#pragma unroll
    for (int u = 0; u < MAXT * TN / 256; ++u) {
      const int idx = (tid - 256) + u * 256, tt = idx / TN, col = idx % TN;
      const float* bp = tt == 0 ? biasp[0] : tt == 1 ? biasp[1] : tt == 2 ? biasp[2] : biasp[3];
      bias_s[idx] = bp ? bp[col] : 0.f;
    }
  }
  const bool relu = d.act == PQ3D_ACT_RELU;
  const int swz = li >> 1;
  // store waves: 16 threads per 256-byte row, 16 rows per pass
  const int st = tid - 256, srow = st >> 4, sch = (st & 15) * 8;
  auto store_tile = [&](bf16_t* C) {
#pragma unroll
    for (int p = 0; p < TM / 16; ++p) {
      const int row = p * 16 + srow;
      *(u32x4*)(C + (long)(m0 + row) * d.ldc + sch) = *(const u32x4*)&Cst[row * LDCS + sch];
    }
  };

  auto sel = [&](auto& arr, int k) { return k == 0 ? arr[0] : k == 1 ? arr[1] : k == 2 ? arr[2] : arr[3]; };   // uniform
#pragma unroll 1
  for (int tt = 0; tt < nt; ++tt) {
    {
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) {
        const int q = tt * 4 + sl;
        if (loader) wait_slices(min(nq - 1 - q, AHEAD - 1));     // this wave's pieces of slice q have landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // own LDS reads / writes retired
        if (tt == 1) TL(1 + 3 * sl);
        __builtin_amdgcn_s_barrier();                            // slice q complete; slice q - 1 (and the C tile) released
        if (tt == 1) TL(2 + 3 * sl);
        if (!loader && sl == 0 && tt > 0) store_tile(sel(Cp, tt - 1));   // the previous tile leaves while this one is multiplied
        if (tt == 1 && sl == 0) TL(14);
        {
          if (loader && q + AHEAD < nq) issue(q + AHEAD, sel(Bp, (q + AHEAD) >> 2));
          const bf16_t* Bs = ring + (q % RING) * SLICE;
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const int ch = ((ks * 4 + lg) ^ swz) * 8;
            u32x4 bf[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = *(const u32x4*)&Bs[(wn + j * 16 + li) * TK + ch];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j) Mma<bf16_t>::mma(acc[i][j], af[i][sl * 2 + ks], bf[j]);
          }
          if (tt == 1) TL(3 + 3 * sl);
          if (sl == 3) {   // the tile is complete: + bias, round, leave it in the staging tile for the store waves
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int col = wn + j * 16 + li;
              const float bn = bias_s[tt * TN + col];
#pragma unroll
              for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const float v = acc[i][j][r] * d.alpha + bn;
                  Cst[(wm + i * 16 + 4 * lg + r) * LDCS + col] = f2bf(relu ? fmaxf(v, 0.f) : v);
                  acc[i][j][r] = 0.f;
                }
            }
            if (tt == 1) TL(13);
          }
        }
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (!loader) {
    store_tile(sel(Cp, nt - 1));
  }
}

}  // namespace

#ifdef PQ3D_ASTAT_TIMELINE
extern "C" int pq3d_astat_timeline(long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_astat_tl), sizeof(long) * 64 * 16);
}
#endif

// Takes a plain bf16 NT launch whose groups share A operands (K = 256, bf16 output): returns false when the call is not of
// that shape or too small to fill the chip with one workgroup per CU.
int pq3d_gemm_options();   // gemm_wk.hip: option word of pq3d_gemm_set_wk
bool pq3d_gemm_astat_try(const pq3d_gemm_desc& d, const pq3d_kdesc& kd, hipStream_t s, int* err) {
  *err = 0;
  if (pq3d_gemm_options() & (1 << 9)) return false;   // tests / A-B: the per-tile kernel instead
  if (d.ct != PQ3D_BF16 || d.dtA != PQ3D_BF16 || d.dtB != PQ3D_BF16 || d.dtC != PQ3D_BF16) return false;
  if (d.transA || d.transB || d.batch != 1 || d.splitk > 1 || d.kconcat > 1 || d.act_grad) return false;
  if (d.act != PQ3D_ACT_NONE && d.act != PQ3D_ACT_RELU) return false;
  if (d.K != KD || d.M < TM || d.M % TM || d.N % TN || d.lda % 8 || d.ldb % 8 || d.ldc % 8) return false;
  if ((long)d.N * d.ldb >= (1L << 31)) return false;
  if (d.row_scale || d.row_fill_flag || d.mask_out || (d.drop.p > 0.f && d.drop.seed)) return false;
  for (int g = 0; g < d.groups; ++g) {
    if (d.A2[g] || d.B2[g] || d.C2[g] || d.row_mask[g] || d.colsum[g] || d.aux[g]) return false;
    if (d.bias[g] && d.dtBias != PQ3D_F32) return false;
    if ((((uintptr_t)d.A[g]) | ((uintptr_t)d.B[g]) | ((uintptr_t)d.C[g])) & 15) return false;
  }
  AstatPlan pl;
  pl.NT = d.N / TN;
  // groups sorted by A pointer (stable: the launch's own order inside a family)
  int n = d.groups;
  for (int g = 0; g < n; ++g) pl.order[g] = (unsigned char)g;
  for (int a = 1; a < n; ++a) {
    const unsigned char v = pl.order[a];
    int b = a - 1;
    while (b >= 0 && (uintptr_t)d.A[pl.order[b]] > (uintptr_t)d.A[v]) { pl.order[b + 1] = pl.order[b]; --b; }
    pl.order[b + 1] = v;
  }
  for (int g = n; g < PQ3D_MAX_GROUPS; ++g) pl.order[g] = 0;
  int n_ent = 0, shared = 0;
  for (int a = 0; a < n;) {
    int b = a;
    while (b < n && d.A[pl.order[b]] == d.A[pl.order[a]]) ++b;
    if (b - a > 1) ++shared;
    const int tiles = (b - a) * pl.NT;
    for (int t = 0; t < tiles; t += MAXT) {
      if (n_ent >= 128) return false;
      pl.ent_start[n_ent] = (unsigned short)(a * pl.NT + t);
      pl.ent_n[n_ent] = (unsigned char)(tiles - t < MAXT ? tiles - t : MAXT);
      ++n_ent;
    }
    a = b;
  }
  if (!shared) return false;                          // nothing to keep stationary: the per-tile kernel is the better fit
  const long items = (long)(d.M / TM) * n_ent;
  if (items < 256) return false;                      // one workgroup per CU: fewer items leave CUs idle
  static std::atomic<unsigned> done{0};
  if (int e = pq3d_enable_big_lds(gemm_nt_astat_kernel, (int)ASTAT_LDS, done)) {
    pq3d_set_error(hipGetErrorString((hipError_t)e));
    *err = e;
    return true;
  }
  hipLaunchKernelGGL(gemm_nt_astat_kernel, dim3(d.M / TM, n_ent), dim3(512), ASTAT_LDS, s, kd, pl);
  return true;
}
