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
#include <cstring>

#include "gemm_common.h"

#ifdef PQ3D_DEBUG_TIMING
__device__ long long pq3d_dbg[16];
#define DBG_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) pq3d_dbg[i] = __builtin_readcyclecounter(); } while (0)
extern "C" int pq3d_debug_read(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pq3d_dbg), sizeof(long long) * 16); }
#else
#define DBG_STAMP(i)
#endif

bool pq3d_gemm_nt128_try(const pq3d_gemm_desc& d, const pq3d_kdesc& kd, hipStream_t s);   // gemm128.hip
bool pq3d_gemm_x3p_try(const pq3d_gemm_desc& d, const pq3d_kdesc& kd, hipStream_t s);     // gemm_x3p.hip
bool pq3d_gemm_tt128_try(const pq3d_gemm_desc& d, const pq3d_kdesc& kd, hipStream_t s);   // gemm128.hip
bool pq3d_gemm_cv128_try(const pq3d_gemm_desc& d, const pq3d_kdesc& kd, hipStream_t s);   // gemm_cv128.hip
bool pq3d_gemm_wk_try(const pq3d_gemm_desc& d, const pq3d_kdesc& kd, hipStream_t s, int* err);   // gemm_wk.hip
bool pq3d_gemm_wktt_try(const pq3d_gemm_desc& d, const pq3d_kdesc& kd, hipStream_t s, int* err);   // gemm_wktt.hip

namespace {

constexpr int BM = 64, BN = 64, NT = 256;

template <typename CT> struct Tile {
  static constexpr int EPL = Mma<CT>::EPL;
  static constexpr int KSTEP = Mma<CT>::KSTEP;
  static constexpr int BKE = 128 / (int)sizeof(CT);         // k elements per tile
  static constexpr int LDK = BKE + 16 / (int)sizeof(CT);    // padded LDS row (elements)
  static constexpr int CPR = BKE / EPL;                     // 16-byte chunks per row (= 8)
};

// ---- FAST stager: branch-free raw loads, conversion at store time ------------------------------------------
// Chunk addresses are computed once per (group, block) by Cursor::init and then only advanced by a constant stride.
// HAS2 (optional fp32 addend A2 / B2) is a COMPILE-TIME property of the kernel: with a run-time "is there an
// addend" branch around its loads the compiler cannot count outstanding loads and drains vmcnt to 0 before every
// LDS store, which serialises the software pipeline.  Groups without an addend inside a HAS2 launch re-read the
// primary operand with scale 0.
template <typename CT, typename TS, bool TR, bool HAS2>
struct Cursor {
  typedef Tile<CT> T;
  const TS* p[2];
  const float* p2[2];
  int kofs[2];   // k index of the chunk inside the tile
  long step;     // elements to advance per K-tile
  float scale2;  // 1 if this group has an addend, else 0
  PQ_DEV void init(const void* base, const void* base2, long off, long ld, int r0, int R, int k0, int tid) {
    scale2 = base2 ? 1.f : 0.f;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int c = tid + it * NT;
      long idx;
      if (!TR) {
        const int row = c / T::CPR, kc = c % T::CPR;
        kofs[it] = kc * T::EPL;
        idx = off + (long)min(r0 + row, R - 1) * ld + k0 + kofs[it];
      } else {
        constexpr int RC = BM / T::EPL;
        const int kk = c / RC, rc = c % RC;
        kofs[it] = kk;
        idx = off + (long)(k0 + kk) * ld + min(r0 + rc * T::EPL, R - T::EPL);
      }
      p[it] = (const TS*)base + idx;
      // no addend for this group: point at fp32-readable memory of the same extent only if the primary is fp32;
      // otherwise fall back to the primary's base (always mapped; values are multiplied by 0)
      if (HAS2) p2[it] = base2 ? (const float*)base2 + idx : (const float*)base;
    }
    step = TR ? (long)T::BKE * ld : (long)T::BKE;
    if (HAS2 && !base2) step2_zero = true; else step2_zero = false;
  }
  bool step2_zero;
};

template <typename CT, typename TS, bool TR, bool HAS2>
struct FastStage {
  typedef Tile<CT> T;
  Raw<TS, T::EPL> r[2];
  Raw<float, T::EPL> r2[2];
  bool kvalid[2];
  float scale2;

  // loads the tile the cursor points at (k0 = its first k index) and advances the cursor by one tile
  PQ_DEV void load(Cursor<CT, TS, TR, HAS2>& cur, int k0, int K) {
    scale2 = cur.scale2;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      kvalid[it] = k0 + cur.kofs[it] < K;
      // out-of-range k chunks re-read the chunk's position in k-tile 0 (valid) and are zeroed at store time
      const long back = kvalid[it] ? 0 : (long)(k0 + cur.kofs[it]) * (TR ? cur.step / T::BKE : 1);
      r[it].load(cur.p[it] - back);
      if (HAS2) r2[it].load(cur.step2_zero ? cur.p2[it] : cur.p2[it] - back);
      cur.p[it] += cur.step;
      if (HAS2 && !cur.step2_zero) cur.p2[it] += cur.step;
    }
  }
  PQ_DEV void store(CT* lds, int tid) const {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int c = tid + it * NT;
      u32x4 packed;
      if constexpr (!HAS2 && sizeof(TS) == sizeof(CT)) {
        packed = __builtin_bit_cast(u32x4, r[it]);   // source already in compute type: straight through
        if (!kvalid[it]) packed = (u32x4){0, 0, 0, 0};
      } else {
        float v[T::EPL];
        r[it].to_float(v);
        if (HAS2) {
          float w[T::EPL];
          r2[it].to_float(w);
#pragma unroll
          for (int j = 0; j < T::EPL; ++j) v[j] += scale2 * w[j];
        }
        if (!kvalid[it]) {
#pragma unroll
          for (int j = 0; j < T::EPL; ++j) v[j] = 0.f;
        }
        packed = pack_frag<CT>(v);
      }
      if (!TR) {
        const int row = c / T::CPR, kc = c % T::CPR;
        *(u32x4*)&lds[row * T::LDK + kc * T::EPL] = packed;
      } else {
        constexpr int RC = BM / T::EPL;
        const int kk = c / RC, rc = c % RC;
        if constexpr (sizeof(CT) == 2) {
          // bf16: keep the operand in its native [k][m] orientation (one 16-byte LDS write); the MFMA fragments are
          // fetched with the transposing LDS read (mma_tile)
          *(u32x4*)&lds[kk * T::LDK + rc * T::EPL] = packed;
        } else {
#pragma unroll
          for (int j = 0; j < T::EPL; ++j) lds[(rc * T::EPL + j) * T::LDK + kk] = __uint_as_float(packed[j]);
        }
      }
    }
  }
  // Split-bf16 staging (compute type PQ3D_BF16X3, row-major fp32 operands only): every fp32 element x (+ addend) is written
  // as hi = bf16(x) and lo = bf16(x - hi) into two tiles; the MFMA loop forms hi*hi + hi*lo + lo*hi, i.e. the product
  // of the operands to ~2^-17 relative instead of 2^-9 -- fp32-grade results from the bf16 matrix cores.
  PQ_DEV void store_split(CT* lds_hi, CT* lds_lo, int tid) const {
    static_assert(!TR && sizeof(TS) == 4 && sizeof(CT) == 2, "split staging: row-major fp32 source, bf16 MFMA");
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int c = tid + it * NT;
      float v[T::EPL], w[T::EPL];
      r[it].to_float(v);
      if (HAS2) {
        r2[it].to_float(w);
#pragma unroll
        for (int j = 0; j < T::EPL; ++j) v[j] += scale2 * w[j];
      }
      if (!kvalid[it]) {
#pragma unroll
        for (int j = 0; j < T::EPL; ++j) v[j] = 0.f;
      }
      const u32x4 hi = pack_frag<CT>(v);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        w[2 * j] = v[2 * j] - __uint_as_float(hi[j] << 16);
        w[2 * j + 1] = v[2 * j + 1] - __uint_as_float(hi[j] & 0xffff0000u);
      }
      const u32x4 lo = pack_frag<CT>(w);
      const int row = c / T::CPR, kc = c % T::CPR;
      *(u32x4*)&lds_hi[row * T::LDK + kc * T::EPL] = hi;
      *(u32x4*)&lds_lo[row * T::LDK + kc * T::EPL] = lo;
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

// do_cs: also accb[i] += A fragment i x all-ones fragment -- every column of that product is the row sum of the A tile
// over k (the fused bias gradient of the weight-gradient GEMMs) on the otherwise idle matrix pipe.  (accb by reference,
// never through a pointer: a pointer to the accumulators sends them to scratch memory.)
template <typename CT, bool KMA, bool KMB>
PQ_DEV void mma_tile(f32x4 (&acc)[2][2], const CT* As, const CT* Bs, int wm, int wn, int li, int lg, f32x4 (&accb)[2], bool do_cs) {
  typedef Tile<CT> T;
#pragma unroll
  for (int ks = 0; ks < T::BKE / T::KSTEP; ++ks) {
    u32x4 fa[2], fb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if constexpr (KMA && sizeof(CT) == 2) fa[i] = km_frag((const bf16_t*)As, T::LDK, wm + i * 16, ks, li, lg);
      else fa[i] = *(const u32x4*)&As[(wm + i * 16 + li) * T::LDK + ks * T::KSTEP + lg * T::EPL];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if constexpr (KMB && sizeof(CT) == 2) fb[j] = km_frag((const bf16_t*)Bs, T::LDK, wn + j * 16, ks, li, lg);
      else fb[j] = *(const u32x4*)&Bs[(wn + j * 16 + li) * T::LDK + ks * T::KSTEP + lg * T::EPL];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) Mma<CT>::mma(acc[i][j], fa[i], fb[j]);
    if (do_cs) {   // wave-uniform
      const uint32_t one = sizeof(CT) == 2 ? 0x3F803F80u : 0x3F800000u;
      const u32x4 ones = (u32x4){one, one, one, one};
#pragma unroll
      for (int i = 0; i < 2; ++i) Mma<CT>::mma(accb[i], fa[i], ones);
    }
  }
}

// split-bf16 product of one staged tile: acc += Ahi.Bhi + Ahi.Blo + Alo.Bhi (the lo.lo term is below fp32 round-off)
PQ_DEV void mma_tile_x3(f32x4 (&acc)[2][2], const bf16_t* Ah, const bf16_t* Al, const bf16_t* Bh, const bf16_t* Bl, int wm,
                        int wn, int li, int lg) {
  typedef Tile<bf16_t> T;
#pragma unroll
  for (int ks = 0; ks < T::BKE / T::KSTEP; ++ks) {
    u32x4 ah[2], al[2], bh[2], bl[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int o = (wm + i * 16 + li) * T::LDK + ks * T::KSTEP + lg * T::EPL;
      ah[i] = *(const u32x4*)&Ah[o];
      al[i] = *(const u32x4*)&Al[o];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int o = (wn + j * 16 + li) * T::LDK + ks * T::KSTEP + lg * T::EPL;
      bh[j] = *(const u32x4*)&Bh[o];
      bl[j] = *(const u32x4*)&Bl[o];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        Mma<bf16_t>::mma(acc[i][j], al[i], bh[j]);
        Mma<bf16_t>::mma(acc[i][j], ah[i], bl[j]);
        Mma<bf16_t>::mma(acc[i][j], ah[i], bh[j]);
      }
  }
}

// Epilogue.  The accumulators (MFMA C-layout: 4 consecutive rows x 1 column per lane) are first transposed through
// LDS (the staging tiles are dead by then) so that every thread owns 16 CONTIGUOUS columns of one row: bias / aux /
// row-flag reads and the C / C2 stores become a handful of 16-byte vector operations instead of 16 scalar,
// branch-wrapped element accesses (which cost 14k cycles -- more than the whole K loop -- in the first version).
constexpr int CLD = BN + 4;  // padded fp32 row of the transposed C tile

PQ_DEV void epilogue(const pq3d_kdesc& d, const GPtrs& gp, const f32x4 (&acc)[2][2], float* Ct, int g, int z, int m0, int n0,
                     int wm, int wn, int li, int lg, int tid, bool bias_done = false) {
  if (d.splitk > 1) {
    // split-K: atomics straight from the C-layout registers -- there the 16 lanes of a group hit 16 CONSECUTIVE
    // addresses per instruction (the row-per-thread layout below would scatter every atomic over 64 cache lines)
    float* C = (float*)gp.C + (long)z * d.strideC;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn + j * 16 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + wm + i * 16 + lg * 4 + r;
          if (row < d.M && col < d.N) unsafeAtomicAdd(C + (long)row * d.ldc + col, acc[i][j][r] * d.alpha);
        }
      }
    return;
  }
  // ---- C-layout registers -> LDS [64][CLD] (conflict-free: consecutive lanes hit consecutive banks)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) Ct[(wm + i * 16 + lg * 4 + r) * CLD + wn + j * 16 + li] = acc[i][j][r] * d.alpha;
  __syncthreads();
  const int lrow = tid >> 2, lcol = (tid & 3) * 16;
  const int row = m0 + lrow, col = n0 + lcol;
  if (row >= d.M || col >= d.N) return;
  float v[16];
#pragma unroll
  for (int j = 0; j < 16; j += 4) { const float4 t = *(const float4*)&Ct[lrow * CLD + lcol + j]; v[j] = t.x; v[j + 1] = t.y; v[j + 2] = t.z; v[j + 3] = t.w; }
  epi_row<16>(d, gp, v, g, z, row, col, bias_done);
}


struct BlockCoords {
  int g, z, m0, n0, kt0, kt1, ng;
  bool active;
};
// grid = (m tiles, n tiles, outputs * batch * splitk): no integer division unless batch > 1 or split-K is used
template <typename CT> PQ_DEV BlockCoords block_coords(const pq3d_kdesc& d, const TileIdx& ti) {
  typedef Tile<CT> T;
  BlockCoords b;
  b.ng = d.kconcat > 0 ? d.kconcat : 1;  // groups walked inside the K loop
  int zz = ti.z, split = 0;
  if (d.splitk > 1) { split = zz % d.splitk; zz /= d.splitk; }
  if (d.batch > 1) { b.z = zz % d.batch; zz /= d.batch; } else b.z = 0;
  b.g = zz * b.ng;
  b.m0 = ti.x * BM;
  b.n0 = ti.y * BN;
  const int nkt = (d.K + T::BKE - 1) / T::BKE;
  b.kt0 = 0; b.kt1 = nkt; b.active = true;
  if (d.splitk > 1) {
    const int per = (nkt + d.splitk - 1) / d.splitk;
    b.kt0 = split * per;
    b.kt1 = min(nkt, b.kt0 + per);
    b.active = b.kt0 < b.kt1;
  }
  return b;
}

template <typename CT, typename TA, typename TB, bool TRA, bool TRB, bool HA2, bool HB2, bool X3 = false>
__global__ __launch_bounds__(NT) void gemm_fast_kernel(const pq3d_kdesc d) {
  typedef Tile<CT> T;
  // X3 (split-bf16): hi tiles first, lo tiles behind them -- one contiguous block, so the C tile of the epilogue still
  // overlays the start of the staging LDS
  __shared__ __attribute__((aligned(16))) CT As[(X3 ? 2 : 1) * (BM + BN) * T::LDK];
  CT* const Bs = As + BM * T::LDK;
  CT* const Al = As + (BM + BN) * T::LDK;
  CT* const Bl = Al + BM * T::LDK;
  static_assert(sizeof(CT) * (BM + BN) * T::LDK >= sizeof(float) * BM * CLD, "C tile must fit in the staging LDS");
  static_assert(!X3 || (!TRA && !TRB && sizeof(CT) == 2 && sizeof(TA) == 4 && sizeof(TB) == 4), "X3: NT, fp32 operands");
  DBG_STAMP(0);
  // Kernel-argument prefetch.  The descriptor is 2.7 KB of kernarg memory, cold in the scalar cache at every launch; read
  // on demand (hipcc sinks each scalar load next to its first use, behind the branches that depend on earlier ones) the
  // prologue was FIVE dependent scalar-cache misses (3.5k cycles before the first operand load was issued, measured with
  // the stamps above) and the epilogue several more.  Pin every scalar the kernel will use here: one batch of loads.
  GPtrs gp;
  const TileIdx ti = tile_index(d.xcd_order);   // hardware order, or the XCD-aware order of a big launch (common.h)
  const int gspec = min(ti.z, PQ3D_MAX_GROUPS - 1);   // the group index of a plain launch (no split / batch / concat)
  gp.load(d, gspec);
#ifndef PQ3D_NO_KARG_PIN
  asm volatile("" ::"s"(d.M), "s"(d.N), "s"(d.K), "s"(d.batch), "s"(d.splitk), "s"(d.kconcat), "s"(d.lda), "s"(d.ldb), "s"(d.ldc),
               "s"(d.strideA), "s"(d.strideB), "s"(d.strideC), "s"(d.alpha), "s"(d.act), "s"(d.act_grad), "s"(d.dtC),
               "s"(d.dtC2), "s"(d.dtAux), "s"(d.dtBias), "s"(d.row_fill), "s"(d.row_scale), "s"(d.row_fill_flag),
               "s"(d.mask_out), "s"(gp.A), "s"(gp.A2), "s"(gp.B), "s"(d.xcd_order));
  asm volatile("" ::"s"(gp.B2), "s"(gp.bias), "s"(gp.aux), "s"(gp.C), "s"(gp.C2), "s"(gp.row_mask));
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const BlockCoords b = block_coords<CT>(d, ti);
  if (!b.active) return;
  if (b.g != gspec) gp.load(d, b.g);   // uniform; split-K / batched / K-concatenated launches
  const long offA = (long)b.z * d.strideA, offB = (long)b.z * d.strideB;
  const int nk = b.kt1 - b.kt0, nit = nk * b.ng;

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // two register stages: tile t+1 and t+2 are in flight while tile t is multiplied
  FastStage<CT, TA, TRA, HA2> sa0, sa1;
  FastStage<CT, TB, TRB, HB2> sb0, sb1;
  Cursor<CT, TA, TRA, HA2> ca;
  Cursor<CT, TB, TRB, HB2> cb;
  int lg_ = b.g, lk_ = b.kt0, issued = 0;  // next (group, k-tile) to load
  ca.init(gp.A, gp.A2, offA, d.lda, b.m0, d.M, b.kt0 * T::BKE, tid);
  cb.init(gp.B, gp.B2, offB, d.ldb, b.n0, d.N, b.kt0 * T::BKE, tid);
  auto issue = [&](FastStage<CT, TA, TRA, HA2>& sa, FastStage<CT, TB, TRB, HB2>& sb) {
    sa.load(ca, lk_ * T::BKE, d.K);
    sb.load(cb, lk_ * T::BKE, d.K);
    ++issued;
    if (++lk_ == b.kt1 && issued < nit) {   // next group of a K-concatenated product
      lk_ = b.kt0; ++lg_;
      ca.init(d.gp[lg_].A, d.gp[lg_].A2, offA, d.lda, b.m0, d.M, b.kt0 * T::BKE, tid);
      cb.init(d.gp[lg_].B, d.gp[lg_].B2, offB, d.ldb, b.n0, d.N, b.kt0 * T::BKE, tid);
    }
  };
  // fused bias gradient (weight-gradient GEMMs): colsum[m] += sum_k A(m,k) by the blocks of the first n-tile column, as one
  // extra MFMA per A fragment against an all-ones fragment (mma_tile); saves one column-sum launch per linear layer
  float* cs_out = nullptr;
  if constexpr (TRA) { if (ti.y == 0) cs_out = d.gp[b.g].colsum; }
  f32x4 accb[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
  const bool do_cs = cs_out != nullptr && wn == 0;   // wave-uniform
  issue(sa0, sb0);
  if (nit > 1) issue(sa1, sb1);
  // the bias row in accumulator layout (a lane's 4 rows share its column), requested right behind the first operand
  // loads and added to the accumulators after the k loop: no dependent global round trip in the epilogue
  const bool bias_early = gp.bias != nullptr && d.dtBias == PQ3D_F32 && d.alpha == 1.f && d.splitk <= 1;
  float bcol[2] = {0.f, 0.f};
  if (bias_early) {
#pragma unroll
    for (int j = 0; j < 2; ++j) bcol[j] = ((const float*)gp.bias)[min(b.n0 + wn + j * 16 + li, d.N - 1)];
  }
  DBG_STAMP(1);
  auto put = [&](const FastStage<CT, TA, TRA, HA2>& sa, const FastStage<CT, TB, TRB, HB2>& sb) {
    if constexpr (X3) { sa.store_split(As, Al, tid); sb.store_split(Bs, Bl, tid); }
    else { sa.store(As, tid); sb.store(Bs, tid); }
  };
  auto mult = [&]() {
    if constexpr (X3) mma_tile_x3(acc, (const bf16_t*)As, (const bf16_t*)Al, (const bf16_t*)Bs, (const bf16_t*)Bl, wm, wn, li, lg);
    else mma_tile<CT, TRA, TRB>(acc, As, Bs, wm, wn, li, lg, accb, TRA && do_cs);
  };
  for (int it = 0; it < nit; it += 2) {
    put(sa0, sb0);
    if (it == 0) DBG_STAMP(2);
    __syncthreads();
    if (issued < nit) issue(sa0, sb0);
    mult();
    __syncthreads();
    if (it == 0) DBG_STAMP(3);
    if (it + 1 < nit) {
      put(sa1, sb1);
      __syncthreads();
      if (issued < nit) issue(sa1, sb1);
        mult();
      __syncthreads();
    }
  }
  DBG_STAMP(4);
  if constexpr (TRA) {
    if (do_cs && li == 0) {   // C layout: lane (column li, rows 4 lg + r); every column holds the same sums
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = b.m0 + wm + i * 16 + 4 * lg + r;
          if (row < d.M) unsafeAtomicAdd(&cs_out[row], accb[i][r] * d.alpha);
        }
    }
  }
  if (bias_early) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] += bcol[j];
  }
  epilogue(d, gp, acc, (float*)As, b.g, b.z, b.m0, b.n0, wm, wn, li, lg, tid, bias_early);
  DBG_STAMP(5);
}

template <typename CT, bool TRA, bool TRB>
__global__ __launch_bounds__(NT) void gemm_slow_kernel(const pq3d_kdesc d) {
  typedef Tile<CT> T;
  __shared__ __attribute__((aligned(16))) CT As[BM * T::LDK];
  __shared__ __attribute__((aligned(16))) CT Bs[BN * T::LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const BlockCoords b = block_coords<CT>(d, tile_index(0));
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
  int lg_ = b.g, lk_ = b.kt0;
  auto load_next = [&]() {
    sa.load(d.gp[lg_].A, d.gp[lg_].A2, d.dtA, d.dtA2, offA, d.lda, b.m0, d.M, lk_ * T::BKE, d.K, tid);
    sb.load(d.gp[lg_].B, d.gp[lg_].B2, d.dtB, d.dtB2, offB, d.ldb, b.n0, d.N, lk_ * T::BKE, d.K, tid);
    if (++lk_ == b.kt1) { lk_ = b.kt0; ++lg_; }
  };
  load_next();
  for (int it = 0; it < nit; ++it) {
    sa.store(As, tid);
    sb.store(Bs, tid);
    __syncthreads();
    if (it + 1 < nit) load_next();
    { f32x4 nocs[2]; mma_tile<CT, false, false>(acc, As, Bs, wm, wn, li, lg, nocs, false); }
    __syncthreads();
  }
  GPtrs gp;
  gp.load(d, b.g);
  epilogue(d, gp, acc, (float*)As, b.g, b.z, b.m0, b.n0, wm, wn, li, lg, tid);
}

bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// Can every group use unconditional 16-byte loads?  Addends must be fp32 and accompany an fp32 primary operand.
template <typename CT> bool fast_ok(const pq3d_gemm_desc& d, bool& a2, bool& b2) {
  constexpr int EPL = Mma<CT>::EPL;
  a2 = b2 = false;
  if (d.transA && !d.transB) return false;
  if (d.lda % EPL || d.ldb % EPL || d.strideA % EPL || d.strideB % EPL) return false;
  if ((!d.transA || !d.transB) && (d.K % EPL)) return false;
  if (d.transA && (d.M % EPL || d.M < EPL)) return false;
  if (d.transB && (d.N % EPL || d.N < EPL)) return false;
  for (int g = 0; g < d.groups; ++g) {
    if (!aligned16(d.A[g]) || !aligned16(d.B[g])) return false;
    if (d.A2[g]) { a2 = true; if (!aligned16(d.A2[g]) || d.dtA2 != PQ3D_F32 || d.dtA != PQ3D_F32) return false; }
    if (d.B2[g]) { b2 = true; if (!aligned16(d.B2[g]) || d.dtB2 != PQ3D_F32 || d.dtB != PQ3D_F32) return false; }
  }
  if (a2 && b2) return false;
  if (a2 && (d.transA || d.transB)) return false;    // A2 only occurs in forward projections (NT)
  if (b2 && !(d.transA && d.transB)) return false;   // B2 only in weight-gradient GEMMs (TT)
  return true;
}

#define LAUNCH(...) hipLaunchKernelGGL((__VA_ARGS__), grid, dim3(NT), 0, s, kd)

template <typename CT, typename TA, typename TB>
void launch_fast_layout(const pq3d_gemm_desc& d, const pq3d_kdesc& kd, dim3 grid, hipStream_t s, bool a2, bool b2) {
  if (!d.transA && !d.transB) {
    if constexpr (sizeof(TA) == 4) { if (a2) { LAUNCH(gemm_fast_kernel<CT, TA, TB, false, false, true, false>); return; } }
    LAUNCH(gemm_fast_kernel<CT, TA, TB, false, false, false, false>);
  } else if (!d.transA && d.transB) {
    LAUNCH(gemm_fast_kernel<CT, TA, TB, false, true, false, false>);
  } else {
    if constexpr (sizeof(TB) == 4) { if (b2) { LAUNCH(gemm_fast_kernel<CT, TA, TB, true, true, false, true>); return; } }
    LAUNCH(gemm_fast_kernel<CT, TA, TB, true, true, false, false>);
  }
}

template <typename CT> void launch_slow(const pq3d_gemm_desc& d, const pq3d_kdesc& kd, dim3 grid, hipStream_t s) {
  if (!d.transA && !d.transB) LAUNCH(gemm_slow_kernel<CT, false, false>);
  else if (!d.transA && d.transB) LAUNCH(gemm_slow_kernel<CT, false, true>);
  else if (d.transA && d.transB) LAUNCH(gemm_slow_kernel<CT, true, true>);
  else LAUNCH(gemm_slow_kernel<CT, true, false>);
}

}  // namespace

extern "C" int pq3d_gemm(const pq3d_gemm_desc* dp, void* stream) {
  PQ_DEVICE_GUARD(stream, dp ? dp->A[0] : nullptr);
  PQ_CHECK_ARG(dp != nullptr, "pq3d_gemm: null descriptor");
  pq3d_gemm_desc d = *dp;
  PQ_CHECK_ARG(d.M >= 0 && d.N >= 0 && d.K >= 0, "pq3d_gemm: negative dims");
  PQ_CHECK_ARG(d.groups >= 1 && d.groups <= PQ3D_MAX_GROUPS, "pq3d_gemm: groups out of range");
  PQ_CHECK_ARG(d.batch >= 1, "pq3d_gemm: batch < 1");
  PQ_CHECK_ARG(d.ct == PQ3D_F32 || d.ct == PQ3D_BF16 || d.ct == PQ3D_BF16X3, "pq3d_gemm: bad compute type");
  if (d.M == 0 || d.N == 0) return 0;
  const int kc = d.kconcat > 0 ? d.kconcat : 1;
  PQ_CHECK_ARG(d.groups % kc == 0, "pq3d_gemm: groups must be a multiple of kconcat");
  for (int g = 0; g < d.groups; ++g) {
    PQ_CHECK_ARG(d.A[g] && d.B[g] && (d.C[g] || (g % kc) != 0), "pq3d_gemm: null A/B/C");
    PQ_CHECK_ARG(!d.act_grad || d.act_grad == PQ3D_ACT_ADD || d.act_grad == PQ3D_ACT_PLANES || d.aux[g] || (g % kc) != 0,
                 "pq3d_gemm: act_grad needs aux");
  }
  if (d.splitk < 1) d.splitk = 1;
  PQ_CHECK_ARG(!(kc > 1 && d.splitk > 1), "pq3d_gemm: kconcat and split-K are exclusive");
  PQ_CHECK_ARG(!(d.drop.p > 0.f && d.drop.seed) || d.splitk == 1, "pq3d_gemm: dropout is not available with split-K");
  PQ_CHECK_DROP(d.drop, (int64_t)d.batch * d.M, d.N, "pq3d_gemm");
  bool any_cs = false;
  for (int g = 0; g < d.groups; ++g) any_cs |= d.colsum[g] != nullptr;
  PQ_CHECK_ARG(!any_cs || (d.transA && d.transB && kc == 1 && d.batch == 1 && d.splitk > 1),
               "pq3d_gemm: colsum needs a transA/transB, non-batched, non-concatenated split-K GEMM");
  hipStream_t s = (hipStream_t)stream;
  pq3d_kdesc kd = make_kdesc(d);   // the kernels' compact form of the descriptor (common.h)
  if (d.act_grad == PQ3D_ACT_PLANES) {   // bf16 hi / lo planes of the result: the 128-row-tile kernels' epilogues only
    if (pq3d_gemm_x3p_try(d, kd, s)) {   // pre-split operands (A2 / B2 = residual planes), ct PQ3D_BF16X3: gemm_x3p.hip
      PQ_LAUNCH_CHECK();
      return 0;
    }
    PQ_CHECK_ARG(d.ct == PQ3D_BF16, "pq3d_gemm: PQ3D_ACT_PLANES with ct PQ3D_BF16X3 needs bf16 A / B / A2 / B2 planes, N % 128 == 0, "
                                    "K % 32 == 0 (gemm_x3p.hip)");
    PQ_CHECK_ARG(d.splitk == 1 && pq3d_gemm_nt128_try(d, kd, s),
                 "pq3d_gemm: PQ3D_ACT_PLANES needs a plain bf16 NT product of the 128-row-tile kernel's shape (M >= 128, N % 128 == "
                 "0, K % 64 == 0, >= 256 tiles, bf16 C and C2)");
    PQ_LAUNCH_CHECK();
    return 0;
  }
  int wk_err = 0;
  // small-M launches (the query side): whole-K tiles, gemm_wk.hip -- same bits, a third of the in-kernel latency
  if ((d.splitk == 1 || d.accumulate) && pq3d_gemm_wk_try(d, kd, s, &wk_err)) {
    if (wk_err) return wk_err;
    PQ_LAUNCH_CHECK();
    return 0;
  }
  if (d.ct == PQ3D_BF16X3) {
    // split-bf16: C = A.B^T of fp32 operands to fp32-grade accuracy on the bf16 matrix cores (3 MFMAs per product term
    // pair).  Available for the aligned row-major (NT) layout with fp32 A and B; anything else runs the exact-f32 MFMA
    // path, which has the same accuracy contract.
    bool a2 = false, b2 = false;
    const bool ok = !d.transA && !d.transB && d.dtA == PQ3D_F32 && d.dtB == PQ3D_F32 && d.splitk == 1 && !any_cs &&
                    fast_ok<bf16_t>(d, a2, b2);
    if (ok) {
      if (pq3d_gemm_cv128_try(d, kd, s)) {   // big-M launches: 128 x 128 tiles (gemm_cv128.hip), same bits
        PQ_LAUNCH_CHECK();
        return 0;
      }
      dim3 grid((d.M + BM - 1) / BM, (d.N + BN - 1) / BN, (d.groups / kc) * d.batch);
      kd.xcd_order = xcd_order_for((long)grid.x * grid.y * grid.z, (long)d.N * d.K * 4 * kc, shared_a_run(d));
      if (a2) LAUNCH(gemm_fast_kernel<bf16_t, float, float, false, false, true, false, true>);
      else LAUNCH(gemm_fast_kernel<bf16_t, float, float, false, false, false, false, true>);
      PQ_LAUNCH_CHECK();
      return 0;
    }
    d.ct = kd.ct = PQ3D_F32;
  }
  if (pq3d_gemm_nt128_try(d, kd, s)) {   // plain big bf16 NT products: 128x128 tiles (gemm128.hip), same bits
    PQ_LAUNCH_CHECK();
    return 0;
  }
  if (!any_cs && pq3d_gemm_cv128_try(d, kd, s)) {   // big products with fp32-stored operands: 128x128 tiles (gemm_cv128.hip)
    PQ_LAUNCH_CHECK();
    return 0;
  }
  if (d.splitk > 1) {
    PQ_CHECK_ARG(d.dtC == PQ3D_F32, "pq3d_gemm: split-K needs fp32 C");
    PQ_CHECK_ARG(d.ldc == d.N && (d.batch == 1 || d.strideC == (int64_t)d.M * d.N),
                 "pq3d_gemm: split-K needs contiguous C");
    if (!d.accumulate) {   // one zero-fill launch for every output (not hipMemsetAsync: see common.h ZeroList)
      ZeroList z;
      for (int g = 0; g < d.groups; ++g) {
        z.add(d.C[g], (long)d.batch * d.M * d.N);
        z.add(d.colsum[g], (long)d.M);
        if (z.n + 2 > PQ_ZERO_MAX || g + 1 == d.groups) {
          if (int e = pq3d_zero_launch(z, s)) return e;
          z.n = 0;
        }
      }
    }
  }
  if (d.splitk > 1 && !d.accumulate && pq3d_gemm_wk_try(d, kd, s, &wk_err)) {   // split-K into a freshly zeroed C
    if (wk_err) return wk_err;
    PQ_LAUNCH_CHECK();
    return 0;
  }
  if (pq3d_gemm_tt128_try(d, kd, s)) {   // big bf16 weight-gradient products: 128x128 tiles, own split factor
    PQ_LAUNCH_CHECK();
    return 0;
  }
  if (pq3d_gemm_wktt_try(d, kd, s, &wk_err)) {   // weight gradients over short reductions: 256-row chunks at once (gemm_wktt.hip)
    if (wk_err) return wk_err;
    PQ_LAUNCH_CHECK();
    return 0;
  }
  dim3 grid((d.M + BM - 1) / BM, (d.N + BN - 1) / BN, (d.groups / kc) * d.batch * d.splitk);
  // XCD-aware tile order (common.h; read by the fast kernels only).  Weight-gradient layouts (transA): super-rows of 8.
  kd.xcd_order = xcd_order_for((long)grid.x * grid.y * grid.z,
                               d.transA ? (1L << 40) : (long)d.N * d.K * (d.dtB == PQ3D_F32 ? 4 : 2) * kc, shared_a_run(d));
  bool a2 = false, b2 = false;
  if (d.ct == PQ3D_BF16) {
    if (fast_ok<bf16_t>(d, a2, b2)) {
      const bool af = d.dtA == PQ3D_F32, bf = d.dtB == PQ3D_F32;
      if (af && bf) launch_fast_layout<bf16_t, float, float>(d, kd, grid, s, a2, b2);
      else if (af && !bf) launch_fast_layout<bf16_t, float, bf16_t>(d, kd, grid, s, a2, b2);
      else if (!af && bf) launch_fast_layout<bf16_t, bf16_t, float>(d, kd, grid, s, a2, b2);
      else launch_fast_layout<bf16_t, bf16_t, bf16_t>(d, kd, grid, s, a2, b2);
    } else {
      PQ_CHECK_ARG(!any_cs, "pq3d_gemm: colsum needs the aligned fast path");
      launch_slow<bf16_t>(d, kd, grid, s);
    }
  } else {
    if (fast_ok<float>(d, a2, b2) && d.dtA == PQ3D_F32 && d.dtB == PQ3D_F32)
      launch_fast_layout<float, float, float>(d, kd, grid, s, a2, b2);
    else { PQ_CHECK_ARG(!any_cs, "pq3d_gemm: colsum needs the aligned fast path"); launch_slow<float>(d, kd, grid, s); }
  }
  PQ_LAUNCH_CHECK();
  return 0;
}
