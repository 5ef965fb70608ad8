// Residual-add + LayerNorm (merged over M branches), forward and backward.  HBM-bound: one wave per row at a time,
// the row lives in registers (d <= 2048 -> <= 32 values per lane), DPP reductions for the statistics, fp32 math
// throughout.  Algorithmic bytes per row: (1 + M) reads + 1 write of d elements.
// Column ownership: VEC (d == 64 * PL, PL a multiple of 4, 16-byte aligned operands): lane owns PL / 4 pieces of 4
// consecutive columns, piece k at columns 256 k + 4 lane -- every row access is a 16-byte load/store and every wave
// instruction covers 1 KB of consecutive bytes (round 4; before, a lane owned PL CONSECUTIVE columns: for PL > 4 the
// 16-byte pieces of one instruction sat PL * 4 bytes apart -- 3 x the cache lines per instruction at d = 768, and the
// 3-branch backward's dx atomics scattered the same way: 0.92 ms per launch at R = 10240); otherwise lane owns columns
// lane + 64 j (4-byte accesses, any d).
// Large R: waves walk rows with a grid stride and keep the NEXT row's loads in flight.
#include <cstdlib>

#include "common.h"

namespace {

constexpr int MAXPL = 32;  // values per lane -> d <= 2048 (round 4: 1152-wide MLP heads of a 768-wide decoder ran into 1024)
constexpr int WPB = 4;     // waves (rows in flight) per block

struct RowStats { float mean, rstd; };

template <int PL, bool VEC> PQ_DEV int colof(int lane, int j) { return VEC ? (j >> 2) * 256 + lane * 4 + (j & 3) : lane + 64 * j; }

template <int PL, bool VEC>
PQ_DEV void load_row(const void* p, int dt, long base, int lane, int d, float (&v)[PL]) {
  if constexpr (VEC) {
    if (dt == PQ3D_F32) {
      const float4* q = (const float4*)((const float*)p + base + lane * 4);
#pragma unroll
      for (int k = 0; k < PL / 4; ++k) { const float4 t = q[64 * k]; v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w; }
    } else {
      const uint2* q = (const uint2*)((const bf16_t*)p + base + lane * 4);
#pragma unroll
      for (int k = 0; k < PL / 4; ++k) {
        const uint2 t = q[64 * k];
        v[4 * k] = __uint_as_float(t.x << 16); v[4 * k + 1] = __uint_as_float(t.x & 0xffff0000u);
        v[4 * k + 2] = __uint_as_float(t.y << 16); v[4 * k + 3] = __uint_as_float(t.y & 0xffff0000u);
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < PL; ++j) {
      const int c = lane + 64 * j;
      v[j] = c < d ? load_elem(p, dt, base + c) : 0.f;
    }
  }
}
template <int PL, bool VEC>
PQ_DEV void store_row(void* p, int dt, long base, int lane, int d, const float (&v)[PL]) {
  if constexpr (VEC) {
    if (dt == PQ3D_F32) {
      float4* q = (float4*)((float*)p + base + lane * 4);
#pragma unroll
      for (int k = 0; k < PL / 4; ++k) q[64 * k] = make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
    } else {
      uint2* q = (uint2*)((bf16_t*)p + base + lane * 4);
#pragma unroll
      for (int k = 0; k < PL / 4; ++k)
        q[64 * k] = make_uint2((unsigned)f2bf(v[4 * k]) | ((unsigned)f2bf(v[4 * k + 1]) << 16),
                          (unsigned)f2bf(v[4 * k + 2]) | ((unsigned)f2bf(v[4 * k + 3]) << 16));
    }
  } else {
#pragma unroll
    for (int j = 0; j < PL; ++j) {
      const int c = lane + 64 * j;
      if (c < d) store_elem(p, dt, base + c, v[j]);
    }
  }
}

template <int PL, bool VEC>
PQ_DEV RowStats row_stats(const float (&v)[PL], int d, int lane, float eps) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < PL; ++j) s += (colof<PL, VEC>(lane, j) < d) ? v[j] : 0.f;
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < PL; ++j) {
    const float t = v[j] - mean;
    q += (colof<PL, VEC>(lane, j) < d) ? t * t : 0.f;
  }
  const float var = wave_sum(q) / (float)d;
  return {mean, 1.f / sqrtf(var + eps)};
}

template <int PL, bool VEC>
__global__ __launch_bounds__(WPB * 64) void add_ln_fwd_kernel(const pq3d_ln_desc d) {
  // kernel-argument prefetch (see gemm_fast_kernel): every descriptor scalar in one batch of scalar loads
#ifndef PQ3D_NO_KARG_PIN
  asm volatile("" ::"s"(d.R), "s"(d.d), "s"(d.M), "s"(d.rows_per_scene), "s"(d.dt_x), "s"(d.dt_o), "s"(d.dt_y), "s"(d.eps), "s"(d.x),
               "s"(d.coef), "s"(d.y), "s"(d.mean), "s"(d.rstd), "s"(d.independent), "s"(d.sum_branches), "s"(d.osum),
               "s"(d.drop.p), "s"(d.drop.seed), "s"(d.o[0]), "s"(d.gamma[0]), "s"(d.beta[0]));
#endif
  const int lane = threadIdx.x & 63;
  const long wave_id = (long)blockIdx.x * WPB + (threadIdx.x >> 6), nwaves = (long)gridDim.x * WPB;
  if (wave_id >= d.R) return;
  // sum_branches: the M inputs are PARTIAL SUMS of one branch (deterministic K-split of the producing GEMM): one
  // LayerNorm of x + dropout(sum_m o_m), statistics / gamma / beta of index 0
  const int mlo = d.independent ? blockIdx.y : 0, mhi = d.independent ? blockIdx.y + 1 : (d.sum_branches ? 1 : d.M);
  const bool drop = drop_on(d.drop);
  const long nscene = d.coef ? d.R / d.rows_per_scene : 1;
  void* yout = d.independent ? d.ys[blockIdx.y] : d.y;
  // the next row's residual and first branch are in flight while this row is normalised
  float nx[PL], no[PL];
  auto fetch = [&](long row) {
    const long base = row * d.d;
    if (d.x) load_row<PL, VEC>(d.x, d.dt_x, base, lane, d.d, nx);
    else {
#pragma unroll
      for (int j = 0; j < PL; ++j) nx[j] = 0.f;
    }
    load_row<PL, VEC>(d.o[mlo], d.dt_o, base, lane, d.d, no);
  };
  fetch(wave_id);
  for (long row = wave_id; row < d.R; row += nwaves) {
    const long base = row * d.d;
    float xr[PL], o0[PL], y[PL];
#pragma unroll
    for (int j = 0; j < PL; ++j) { xr[j] = nx[j]; o0[j] = no[j]; y[j] = 0.f; }
    if (row + nwaves < d.R) fetch(row + nwaves);
    for (int m = mlo; m < mhi; ++m) {
      float v[PL], ov[PL];
      if (m == mlo) {
#pragma unroll
        for (int j = 0; j < PL; ++j) ov[j] = o0[j];
      } else load_row<PL, VEC>(d.o[m], d.dt_o, base, lane, d.d, ov);
      if (d.sum_branches) {
        for (int p = 1; p < d.M; ++p) {
          float t[PL];
          load_row<PL, VEC>(d.o[p], d.dt_o, base, lane, d.d, t);
#pragma unroll
          for (int j = 0; j < PL; ++j) ov[j] += t[j];
        }
        if (d.osum) store_row<PL, VEC>(d.osum, PQ3D_F32, base, lane, d.d, ov);   // kept for the backward pass
      }
      if (drop) {   // uniform branch around pure ALU: residual dropout of branch m (site drop.site + m)
        const DropState ds = drop_init(d.drop, m, d.d);
#pragma unroll
        for (int j = 0; j < PL; ++j)
          ov[j] = drop_keep(ds, (uint32_t)row, (uint32_t)colof<PL, VEC>(lane, j)) ? ov[j] * ds.scale : 0.f;
      }
#pragma unroll
      for (int j = 0; j < PL; ++j) v[j] = (colof<PL, VEC>(lane, j) < d.d) ? xr[j] + ov[j] : 0.f;
      const RowStats st = row_stats<PL, VEC>(v, d.d, lane, d.eps);
      const float w = (d.independent || d.sum_branches) ? 1.f
                      : (d.coef ? d.coef[m * nscene + row / d.rows_per_scene] : 1.f / (float)d.M);
      float gm[PL], bt[PL];
      load_row<PL, VEC>(d.gamma[m], PQ3D_F32, 0, lane, d.d, gm);
      load_row<PL, VEC>(d.beta[m], PQ3D_F32, 0, lane, d.d, bt);
#pragma unroll
      for (int j = 0; j < PL; ++j) y[j] += w * ((v[j] - st.mean) * st.rstd * gm[j] + bt[j]);
      if (lane == 0) {
        d.mean[(long)m * d.R + row] = st.mean;
        d.rstd[(long)m * d.R + row] = st.rstd;
      }
    }
    store_row<PL, VEC>(yout, d.dt_y, base, lane, d.d, y);
  }
}

// Backward: blockIdx.y = branch m; each wave walks rows with a grid stride and keeps the per-lane partial dgamma/dbeta of
// its columns in registers.  dx (sum over branches) is accumulated with atomics only when M > 1.
template <int PL, bool VEC, int NW>
__global__ __launch_bounds__(NW * 64) void add_ln_bwd_kernel(const pq3d_ln_desc d) {
#ifndef PQ3D_NO_KARG_PIN
  asm volatile("" ::"s"(d.R), "s"(d.d), "s"(d.M), "s"(d.rows_per_scene), "s"(d.dt_x), "s"(d.dt_o), "s"(d.x), "s"(d.coef), "s"(d.mean),
               "s"(d.rstd), "s"(d.dy), "s"(d.dx), "s"(d.independent), "s"(d.sum_branches), "s"(d.drop.p), "s"(d.drop.seed));
#endif
#ifndef PQ3D_NO_KARG_PIN
  asm volatile("" ::"s"(d.o[blockIdx.y]), "s"(d.gamma[blockIdx.y]), "s"(d.d_o[blockIdx.y]), "s"(d.dgamma[blockIdx.y]),
               "s"(d.dbeta[blockIdx.y]));
#endif
  __shared__ float red[2][NW][64 * PL];
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.y;
  const long wave_id = (long)blockIdx.x * NW + (threadIdx.x >> 6), nwaves = (long)gridDim.x * NW;
  const long nscene = d.coef ? d.R / d.rows_per_scene : 1;
  float dg[PL], db[PL], gam[PL];
  load_row<PL, VEC>(d.gamma[m], PQ3D_F32, 0, lane, d.d, gam);
#pragma unroll
  for (int j = 0; j < PL; ++j) { dg[j] = 0.f; db[j] = 0.f; }
  // software-pipelined over rows: the three row loads of the NEXT row are in flight while this row is reduced
  const float* dyp = d.independent ? d.dys[m] : d.dy;
  const bool drop = drop_on(d.drop);
  DropState dst;
  if (drop) dst = drop_init(d.drop, m, d.d);
  float nv[PL], ndy[PL];
  unsigned nkeep = 0xffffffffu;   // bit j: column j of the fetched row survived the residual dropout
  auto fetch = [&](long row) {
    const long base = row * d.d;
    float xv[PL], ov[PL];
    if (d.x) load_row<PL, VEC>(d.x, d.dt_x, base, lane, d.d, xv);
    else {
#pragma unroll
      for (int j = 0; j < PL; ++j) xv[j] = 0.f;
    }
    load_row<PL, VEC>(d.o[m], d.dt_o, base, lane, d.d, ov);
    load_row<PL, VEC>(dyp, PQ3D_F32, base, lane, d.d, ndy);
    if (d.dy2) {   // upstream gradient in up to three addends: (dy + dy2) + dy3
      float t[PL];
      load_row<PL, VEC>(d.dy2, PQ3D_F32, base, lane, d.d, t);
#pragma unroll
      for (int j = 0; j < PL; ++j) ndy[j] += t[j];
      if (d.dy3) {
        load_row<PL, VEC>(d.dy3, PQ3D_F32, base, lane, d.d, t);
#pragma unroll
        for (int j = 0; j < PL; ++j) ndy[j] += t[j];
      }
    }
    if (d.sum_branches) {
      for (int p = 1; p < d.M; ++p) {
        float t[PL];
        load_row<PL, VEC>(d.o[p], d.dt_o, base, lane, d.d, t);
#pragma unroll
        for (int j = 0; j < PL; ++j) ov[j] += t[j];
      }
    }
    if (drop) {
      nkeep = 0;
#pragma unroll
      for (int j = 0; j < PL; ++j) {
        const bool k = drop_keep(dst, (uint32_t)row, (uint32_t)colof<PL, VEC>(lane, j));
        nkeep |= (unsigned)k << j;
        ov[j] = k ? ov[j] * dst.scale : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < PL; ++j) nv[j] = xv[j] + ov[j];
  };
  if (wave_id < d.R) fetch(wave_id);
  for (long row = wave_id; row < d.R; row += nwaves) {
    const long base = row * d.d;
    float v[PL], dyr[PL];
#pragma unroll
    for (int j = 0; j < PL; ++j) { v[j] = nv[j]; dyr[j] = ndy[j]; }
    const unsigned keep = nkeep;
    const float mean = d.mean[(long)m * d.R + row], rstd = d.rstd[(long)m * d.R + row];
    const float w = (d.independent || d.sum_branches) ? 1.f
                    : (d.coef ? d.coef[m * nscene + row / d.rows_per_scene] : 1.f / (float)d.M);
    if (row + nwaves < d.R) fetch(row + nwaves);
    float xh[PL], dz[PL];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < PL; ++j) {
      if (colof<PL, VEC>(lane, j) < d.d) {
        const float du = w * dyr[j];
        xh[j] = (v[j] - mean) * rstd;
        dg[j] += du * xh[j];
        db[j] += du;
        dz[j] = du * gam[j];
        s1 += dz[j];
        s2 += dz[j] * xh[j];
      } else { xh[j] = 0.f; dz[j] = 0.f; }
    }
    s1 = wave_sum(s1) / (float)d.d;
    s2 = wave_sum(s2) / (float)d.d;
    float g[PL], go[PL];
#pragma unroll
    for (int j = 0; j < PL; ++j) {
      g[j] = rstd * (dz[j] - s1 - xh[j] * s2);
      go[j] = drop ? (((keep >> j) & 1u) ? g[j] * dst.scale : 0.f) : g[j];
    }
    store_row<PL, VEC>(d.d_o[m], PQ3D_F32, base, lane, d.d, go);
    if (d.dx && !d.independent) {
      if (d.M == 1 || d.sum_branches) store_row<PL, VEC>(d.dx, PQ3D_F32, base, lane, d.d, g);
      else {
#pragma unroll
        for (int j = 0; j < PL; ++j) {
          const int c = colof<PL, VEC>(lane, j);
          if (c < d.d) unsafeAtomicAdd(&d.dx[base + c], g[j]);
        }
      }
    }
  }
  // block-level reduction of the parameter-gradient partials (LDS index = column)
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < PL; ++j) {
    const int c = colof<PL, VEC>(lane, j);
    red[0][wave][c] = dg[j];
    red[1][wave][c] = db[j];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < d.d; c += NW * 64) {   // ONE atomic per column per block
    float sg = 0.f, sb = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) { sg += red[0][w][c]; sb += red[1][w][c]; }
    unsafeAtomicAdd(&d.dgamma[m][c], sg);
    unsafeAtomicAdd(&d.dbeta[m][c], sb);
  }
}

// Merged-branch backward for the big streaming calls (the cross-attention sublayer of a large batch: M = 3 memories, R in the
// thousands): ONE wave handles a row for all MB branches -- x and the upstream gradient are read once, dx = sum_m g_m is
// formed in registers and stored once.  The one-branch-per-block kernel above sums dx with one atomic per element per
// branch onto a zero-filled buffer (R d x (4 + 3 x 8) bytes of extra traffic; 106 us per launch at R = 10240, d = 768).
// Same arithmetic per (row, branch); the parameter-gradient partials are reduced per branch as above.
template <int PL, bool VEC, int NW, int MB>
__global__ __launch_bounds__(NW * 64) void add_ln_bwd_merged_kernel(const pq3d_ln_desc d) {
  __shared__ float red[2][NW][64 * PL];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long wave_id = (long)blockIdx.x * NW + wave, nwaves = (long)gridDim.x * NW;
  const long nscene = d.coef ? d.R / d.rows_per_scene : 1;
  float dg[MB][PL], db[MB][PL];   // (gamma is re-read per row: L1-resident, and 2 MB PL accumulators already fill the budget)
#pragma unroll
  for (int m = 0; m < MB; ++m) {
#pragma unroll
    for (int j = 0; j < PL; ++j) { dg[m][j] = 0.f; db[m][j] = 0.f; }
  }
  const bool drop = drop_on(d.drop);
  for (long row = wave_id; row < d.R; row += nwaves) {
    const long base = row * d.d;
    float xv[PL], dyr[PL], gsum[PL];
    if (d.x) load_row<PL, VEC>(d.x, d.dt_x, base, lane, d.d, xv);
    else {
#pragma unroll
      for (int j = 0; j < PL; ++j) xv[j] = 0.f;
    }
    load_row<PL, VEC>(d.dy, PQ3D_F32, base, lane, d.d, dyr);
    if (d.dy2) {   // upstream gradient in up to three addends: (dy + dy2) + dy3
      float t[PL];
      load_row<PL, VEC>(d.dy2, PQ3D_F32, base, lane, d.d, t);
#pragma unroll
      for (int j = 0; j < PL; ++j) dyr[j] += t[j];
      if (d.dy3) {
        load_row<PL, VEC>(d.dy3, PQ3D_F32, base, lane, d.d, t);
#pragma unroll
        for (int j = 0; j < PL; ++j) dyr[j] += t[j];
      }
    }
#pragma unroll
    for (int j = 0; j < PL; ++j) gsum[j] = 0.f;
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      float ov[PL], gam[PL];
      load_row<PL, VEC>(d.o[m], d.dt_o, base, lane, d.d, ov);
      load_row<PL, VEC>(d.gamma[m], PQ3D_F32, 0, lane, d.d, gam);
      unsigned keep = 0xffffffffu;
      DropState dst;
      if (drop) {
        dst = drop_init(d.drop, m, d.d);
        keep = 0;
#pragma unroll
        for (int j = 0; j < PL; ++j) {
          const bool k = drop_keep(dst, (uint32_t)row, (uint32_t)colof<PL, VEC>(lane, j));
          keep |= (unsigned)k << j;
          ov[j] = k ? ov[j] * dst.scale : 0.f;
        }
      }
      const float mean = d.mean[(long)m * d.R + row], rstd = d.rstd[(long)m * d.R + row];
      const float w = d.coef ? d.coef[m * nscene + row / d.rows_per_scene] : 1.f / (float)d.M;
      float xh[PL], dz[PL];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < PL; ++j) {
        if (colof<PL, VEC>(lane, j) < d.d) {
          const float du = w * dyr[j];
          xh[j] = ((xv[j] + ov[j]) - mean) * rstd;
          dg[m][j] += du * xh[j];
          db[m][j] += du;
          dz[j] = du * gam[j];
          s1 += dz[j];
          s2 += dz[j] * xh[j];
        } else { xh[j] = 0.f; dz[j] = 0.f; }
      }
      s1 = wave_sum(s1) / (float)d.d;
      s2 = wave_sum(s2) / (float)d.d;
      float go[PL];
#pragma unroll
      for (int j = 0; j < PL; ++j) {
        const float g = rstd * (dz[j] - s1 - xh[j] * s2);
        gsum[j] += g;
        go[j] = drop ? (((keep >> j) & 1u) ? g * dst.scale : 0.f) : g;
      }
      store_row<PL, VEC>(d.d_o[m], PQ3D_F32, base, lane, d.d, go);
    }
    store_row<PL, VEC>(d.dx, PQ3D_F32, base, lane, d.d, gsum);
  }
  // block-level reduction of the parameter-gradient partials, one branch after the other (LDS index = column)
#pragma unroll
  for (int m = 0; m < MB; ++m) {
#pragma unroll
    for (int j = 0; j < PL; ++j) {
      const int c = colof<PL, VEC>(lane, j);
      red[0][wave][c] = dg[m][j];
      red[1][wave][c] = db[m][j];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < d.d; c += NW * 64) {   // ONE atomic per column per block
      float sg = 0.f, sb = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) { sg += red[0][w][c]; sb += red[1][w][c]; }
      unsafeAtomicAdd(&d.dgamma[m][c], sg);
      unsafeAtomicAdd(&d.dbeta[m][c], sb);
    }
    __syncthreads();
  }
}

int check_ln(const pq3d_ln_desc& d, bool bwd) {
  PQ_CHECK_ARG(d.R >= 0 && d.d >= 1 && d.d <= 64 * MAXPL, "pq3d_add_ln: d must be in [1,2048]");
  PQ_CHECK_ARG(d.M >= 1 && d.M <= PQ3D_MAX_GROUPS, "pq3d_add_ln: M out of range");
  PQ_CHECK_ARG(d.rows_per_scene >= 1 && (d.R % d.rows_per_scene) == 0, "pq3d_add_ln: R % rows_per_scene != 0");
  PQ_CHECK_ARG(d.mean && d.rstd, "pq3d_add_ln: null mean/rstd");
  PQ_CHECK_DROP(d.drop, d.R, d.d, "pq3d_add_ln");
  PQ_CHECK_ARG(!(d.sum_branches && (d.independent || d.coef)), "pq3d_add_ln: sum_branches excludes independent / coef");
  for (int m = 0; m < d.M; ++m) {
    const bool first = m == 0 || !d.sum_branches;   // sum mode: one gamma / beta / d_o (index 0)
    PQ_CHECK_ARG(d.o[m] && (!first || (d.gamma[m] && d.beta[m])), "pq3d_add_ln: null o/gamma/beta");
    if (bwd && first) PQ_CHECK_ARG(d.d_o[m] && d.dgamma[m] && d.dbeta[m], "pq3d_add_ln_bwd: null grads");
  }
  if (bwd) PQ_CHECK_ARG((!d.dy2 && !d.dy3) || (d.dy2 && !d.independent), "pq3d_add_ln_bwd: dy2 / dy3 (dy3 only with dy2; not with independent branches)");
  if (d.independent) {
    for (int m = 0; m < d.M; ++m) PQ_CHECK_ARG(bwd ? d.dys[m] != nullptr : d.ys[m] != nullptr, "pq3d_add_ln: null ys/dys");
  } else if (bwd) PQ_CHECK_ARG(d.dy != nullptr, "pq3d_add_ln_bwd: null dy");
  else PQ_CHECK_ARG(d.y != nullptr, "pq3d_add_ln_fwd: null y");
  return 0;
}

// VEC: whole rows in 16-byte pieces (d == 64 * PL with PL a multiple of 4, every operand 16-byte aligned)
bool ln_vec_ok(const pq3d_ln_desc& d, bool bwd, bool no_dx_atomics = false) {
  // 64 * PL with PL in {4, 8, 12, 16, 32}
  if (d.d != 256 && d.d != 512 && d.d != 768 && d.d != 1024 && d.d != 2048) return false;
  // the merged-branch backward sums dx with one atomic per element: a lane's 4-column pieces make every atomic instruction
  // touch 64 addresses 16 bytes apart (4 x the cache lines of the lane + 64 j layout).  Fine at d = 256 (measured, config
  // 2), 0.32 ms instead of 0.1 ms per launch at d = 768, R = 10240: wide rows keep the 4-byte layout there.
  if (bwd && !no_dx_atomics && d.d > 256 && d.M > 1 && d.dx && !d.independent && !d.sum_branches) return false;
  auto al = [](const void* p, int dt) { return ((uintptr_t)p & (dt == PQ3D_F32 ? 15 : 7)) == 0; };
  bool ok = al(d.x, d.dt_x);
  const int nm = d.M;
  for (int m = 0; m < nm; ++m) {
    const bool first = m == 0 || !d.sum_branches;
    ok = ok && al(d.o[m], d.dt_o);
    if (first) ok = ok && al(d.gamma[m], PQ3D_F32) && al(d.beta[m], PQ3D_F32);
    if (d.independent) ok = ok && (bwd ? al(d.dys[m], PQ3D_F32) : al(d.ys[m], d.dt_y));
    if (bwd && first) ok = ok && al(d.d_o[m], PQ3D_F32);
  }
  if (!d.independent) ok = ok && (bwd ? al(d.dy, PQ3D_F32) && al(d.dx, PQ3D_F32) && al(d.dy2, PQ3D_F32) && al(d.dy3, PQ3D_F32) : al(d.y, d.dt_y));
  return ok && al(d.osum, PQ3D_F32);
}

#define LN_DISPATCH(LAUNCH, vec)                                                       \
  if (d.d <= 64) { LAUNCH(1, false); }                                                 \
  else if (d.d <= 128) { LAUNCH(2, false); }                                           \
  else if (d.d <= 256) { if (vec) { LAUNCH(4, true); } else { LAUNCH(4, false); } }    \
  else if (d.d <= 512) { if (vec) { LAUNCH(8, true); } else { LAUNCH(8, false); } }    \
  else if (d.d == 768 && vec) { LAUNCH(12, true); }                                     \
  else if (d.d <= 1024) { if (vec) { LAUNCH(16, true); } else { LAUNCH(16, false); } } \
  else { if (vec) { LAUNCH(32, true); } else { LAUNCH(32, false); } }

}  // namespace

extern "C" int pq3d_add_ln_fwd(const pq3d_ln_desc* dp, void* stream) {
  PQ_DEVICE_GUARD(stream, dp ? (dp->x ? dp->x : dp->o[0]) : nullptr);
  PQ_CHECK_ARG(dp != nullptr, "pq3d_add_ln_fwd: null descriptor");
  const pq3d_ln_desc d = *dp;
  if (int e = check_ln(d, false)) return e;
  if (d.R == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  // one row per wave up to 4096 rows; beyond that waves walk rows (next row's loads in flight behind the statistics)
  long nb = (d.R + WPB - 1) / WPB;
  if (nb > 1024) nb = max(1024L, (d.R + 4 * WPB - 1) / (4 * WPB));
  if (nb > 4096) nb = 4096;
  dim3 grid((unsigned)nb, d.independent ? d.M : 1);
  const bool vec = ln_vec_ok(d, false);
#define FWD(PLV, V) hipLaunchKernelGGL((add_ln_fwd_kernel<PLV, V>), grid, dim3(WPB * 64), 0, s, d)
  LN_DISPATCH(FWD, vec)
#undef FWD
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_add_ln_bwd(const pq3d_ln_desc* dp, void* stream) {
  PQ_DEVICE_GUARD(stream, dp ? (dp->x ? dp->x : dp->o[0]) : nullptr);
  PQ_CHECK_ARG(dp != nullptr, "pq3d_add_ln_bwd: null descriptor");
  const pq3d_ln_desc d = *dp;
  if (int e = check_ln(d, true)) return e;
  hipStream_t s = (hipStream_t)stream;
  // 3-branch merged calls (one dx for all branches): the all-branches-per-wave kernel, no dx atomics, no dx zero-fill
  // (config 2: 1.4136 -> 1.3965 ms per step, config 5: 6.77 -> 6.71; interleaved runs on one box)
#ifndef PQ3D_LN_MERGED_MIN_R
#define PQ3D_LN_MERGED_MIN_R 512
#endif
  const bool merged = d.R >= PQ3D_LN_MERGED_MIN_R && d.M == 3 && d.dx && !d.independent && !d.sum_branches &&
                      (d.d <= 512 || (d.d == 768 && ln_vec_ok(d, true, true)));   // wider rows would spill the 6 d / 64 accumulators
  {   // zero the atomics targets in one launch (not hipMemsetAsync: see common.h ZeroList)
    ZeroList z;
    for (int m = 0; m < (d.sum_branches ? 1 : d.M) && !d.accumulate; ++m) { z.add(d.dgamma[m], d.d); z.add(d.dbeta[m], d.d); }
    if (z.full()) { if (int e = pq3d_zero_launch(z, s)) return e; z.n = 0; }
    if (d.R > 0 && d.dx && d.M > 1 && !d.independent && !d.sum_branches && !d.dx_zeroed && !merged) z.add(d.dx, (long)d.R * d.d);
    if (int e = pq3d_zero_launch(z, s)) return e;
  }
  if (d.R == 0) return 0;
  // rows per wave (software-pipelined): 2 for the query-sized calls, more for the big streaming ones (fewer blocks ->
  // fewer parameter-gradient atomics per column)
  // big streaming calls: 8 waves per block, 8 rows per wave -- the parameter-gradient atomics (one per column per block,
  // the 64-byte lines bounce between the XCDs' L2s) set the time, so few fat blocks win: measured at R = 2 x 8192,
  // blocks x waves: 1024x4 43 us, 512x4 28, 256x4 20, 256x8 16, 128x16 15, 64x16 28; query-sized calls: 4 waves x 2 rows
  // (rows wider than 1024 keep the 4-wave kernel at every row count: with 32 values per lane the 8-wave kernel's reduction
  // buffer would be 128 KB of static LDS -- one block per CU and outside the 64 KB every other kernel stays within)
  const bool big = d.R >= 4096 && (merged || d.d <= 1024);
  // query-sized calls: ~100 blocks of 4 waves (measured, rows-per-wave sweep: R = 800: 2 rows per wave 126 us per step, 4: 130,
  // 8: 176; R = 1600: 2: 300, 4: 275, 8: 311 -- more rows per wave = fewer contended parameter-gradient atomics, fewer = more
  // rows in flight)
  const int rpw_small = (int)((d.R + 399) / 400 < 2 ? 2 : ((d.R + 399) / 400 > 8 ? 8 : (d.R + 399) / 400));
  const int rpw = big ? 8 : rpw_small, nw = big ? 8 : 4;
  long nb = (d.R + rpw * nw - 1) / (rpw * nw);
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  dim3 grid((unsigned)nb, d.sum_branches ? 1 : d.M);
  if (merged) {
    grid.y = 1;
    const bool vecm = ln_vec_ok(d, true, true);
    // 8 waves per block also for the query-sized calls (one row per wave at R = 800): measured at config 2, same box, 4
    // waves x 2 rows 1.4155 ms per step (= the atomics kernel), 8 waves 1.409
#define BWDM(PLV, V) hipLaunchKernelGGL((add_ln_bwd_merged_kernel<PLV, V, 8, 3>), grid, dim3(512), 0, s, d)
    if (d.d <= 64) { BWDM(1, false); }
    else if (d.d <= 128) { BWDM(2, false); }
    else if (d.d <= 256) { if (vecm) { BWDM(4, true); } else { BWDM(4, false); } }
    else if (d.d <= 512) { if (vecm) { BWDM(8, true); } else { BWDM(8, false); } }
    else { BWDM(12, true); }
#undef BWDM
    PQ_LAUNCH_CHECK();
    return 0;
  }
  const bool vec = ln_vec_ok(d, true);
#define BWD8(PLV, V) hipLaunchKernelGGL((add_ln_bwd_kernel<PLV, V, 8>), grid, dim3(512), 0, s, d)
#define BWD4(PLV, V) hipLaunchKernelGGL((add_ln_bwd_kernel<PLV, V, 4>), grid, dim3(256), 0, s, d)
  if (big) { LN_DISPATCH(BWD8, vec) } else { LN_DISPATCH(BWD4, vec) }
#undef BWD8
#undef BWD4
  PQ_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------ RMSNorm (T5)
// T5LayerNorm (HF transformers modeling_t5.py, the generation head's third-party body):
// y = x * rsqrt(mean(x^2) + eps) * w -- no mean subtraction, no bias.  One wave per row, fp32.
namespace {
template <int PL>
__global__ __launch_bounds__(WPB * 64) void rmsnorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               float* __restrict__ y, float* __restrict__ rstd, long R, int d,
                                                               float eps) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * WPB + (threadIdx.x >> 6);
  if (row >= R) return;
  float v[PL], s = 0.f;
#pragma unroll
  for (int j = 0; j < PL; ++j) {
    const int c = lane + 64 * j;
    v[j] = c < d ? x[row * d + c] : 0.f;
    s += v[j] * v[j];
  }
  const float r = 1.f / sqrtf(wave_sum(s) / (float)d + eps);
#pragma unroll
  for (int j = 0; j < PL; ++j) {
    const int c = lane + 64 * j;
    if (c < d) y[row * d + c] = v[j] * r * w[c];
  }
  if (lane == 0) rstd[row] = r;
}
// dx = r (g - xh mean(g xh)),  g = dy w,  xh = x r;  dw += sum_rows dy xh
template <int PL>
__global__ __launch_bounds__(WPB * 64) void rmsnorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ rstd, const float* __restrict__ dy,
                                                               float* __restrict__ dx, float* __restrict__ dw, long R, int d,
                                                               const float* __restrict__ dres, const pq3d_dropout dr,
                                                               float* __restrict__ dxm) {
  __shared__ float red[WPB][64 * PL];
  // dxm (optional): a second copy of dx with the dropout mask (x 1/(1-p)) of site `dr` over [R, d] applied -- the gradient
  // the PRECEDING sublayer's output projection needs (x = residual + dropout(o W^T): its dropout's backward), written here
  // instead of by a launch of its own
  DropState dst;
  if (dxm) dst = drop_init(dr, 0, d);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long wave_id = (long)blockIdx.x * WPB + wave, nwaves = (long)gridDim.x * WPB;
  float acc[PL], wv[PL];
#pragma unroll
  for (int j = 0; j < PL; ++j) { acc[j] = 0.f; wv[j] = (lane + 64 * j < d) ? w[lane + 64 * j] : 0.f; }
  for (long row = wave_id; row < R; row += nwaves) {
    const float r = rstd[row];
    float xh[PL], g[PL], s = 0.f;
#pragma unroll
    for (int j = 0; j < PL; ++j) {
      const int c = lane + 64 * j;
      const float dyv = c < d ? dy[row * d + c] : 0.f;
      xh[j] = c < d ? x[row * d + c] * r : 0.f;
      g[j] = dyv * wv[j];
      acc[j] += dyv * xh[j];
      s += g[j] * xh[j];
    }
    s = wave_sum(s) / (float)d;
#pragma unroll
    for (int j = 0; j < PL; ++j) {
      const int c = lane + 64 * j;
      if (c < d) {
        const float v = r * (g[j] - xh[j] * s) + (dres ? dres[row * d + c] : 0.f);   // + the residual branch's gradient
        dx[row * d + c] = v;
        if (dxm) dxm[row * d + c] = drop_keep(dst, (uint32_t)row, (uint32_t)c) ? v * dst.scale : 0.f;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < PL; ++j) red[wave][lane + 64 * j] = acc[j];
  __syncthreads();
  for (int c = threadIdx.x; c < d; c += WPB * 64) {
    float sres = 0.f;
#pragma unroll
    for (int wv_ = 0; wv_ < WPB; ++wv_) sres += red[wv_][c];
    unsafeAtomicAdd(&dw[c], sres);
  }
}
}  // namespace

#define RMS_DISPATCH(kernel, grid, ...)                                                                  \
  if (d <= 64) hipLaunchKernelGGL((kernel<1>), grid, dim3(WPB * 64), 0, s, __VA_ARGS__);                \
  else if (d <= 128) hipLaunchKernelGGL((kernel<2>), grid, dim3(WPB * 64), 0, s, __VA_ARGS__);          \
  else if (d <= 256) hipLaunchKernelGGL((kernel<4>), grid, dim3(WPB * 64), 0, s, __VA_ARGS__);          \
  else if (d <= 512) hipLaunchKernelGGL((kernel<8>), grid, dim3(WPB * 64), 0, s, __VA_ARGS__);          \
  else if (d <= 1024) hipLaunchKernelGGL((kernel<16>), grid, dim3(WPB * 64), 0, s, __VA_ARGS__);       \
  else hipLaunchKernelGGL((kernel<32>), grid, dim3(WPB * 64), 0, s, __VA_ARGS__);

extern "C" int pq3d_rmsnorm_fwd(const float* x, const float* w, float* y, float* rstd, int64_t R, int32_t d, float eps,
                                void* stream) {
  PQ_DEVICE_GUARD(stream, x);
  PQ_CHECK_ARG(x && w && y && rstd && R >= 0 && d >= 1 && d <= 64 * MAXPL, "pq3d_rmsnorm_fwd: bad args (d <= 2048)");
  if (R == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)((R + WPB - 1) / WPB));
  RMS_DISPATCH(rmsnorm_fwd_kernel, grid, x, w, y, rstd, (long)R, d, eps)
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_rmsnorm_bwd(const float* x, const float* w, const float* rstd, const float* dy, float* dx, float* dw,
                                int64_t R, int32_t d, int32_t accumulate, void* stream) {
  return pq3d_rmsnorm_bwd_res(x, w, rstd, dy, nullptr, dx, dw, R, d, accumulate, stream);
}

extern "C" int pq3d_rmsnorm_bwd_res(const float* x, const float* w, const float* rstd, const float* dy, const float* dres,
                                    float* dx, float* dw, int64_t R, int32_t d, int32_t accumulate, void* stream) {
  return pq3d_rmsnorm_bwd_res_drop(x, w, rstd, dy, dres, dx, dw, R, d, accumulate, nullptr, nullptr, stream);
}

extern "C" int pq3d_rmsnorm_bwd_res_drop(const float* x, const float* w, const float* rstd, const float* dy, const float* dres,
                                         float* dx, float* dw, int64_t R, int32_t d, int32_t accumulate, const pq3d_dropout* drop,
                                         float* dxm, void* stream) {
  PQ_DEVICE_GUARD(stream, x);
  PQ_CHECK_ARG(!dxm || (drop && drop->seed && drop->p > 0.f), "pq3d_rmsnorm_bwd_res_drop: dxm needs a dropout site");
  if (dxm) PQ_CHECK_DROP(*drop, R, d, "pq3d_rmsnorm_bwd_res_drop");
  pq3d_dropout dr_v;
  dr_v.p = 0.f; dr_v.site = 0; dr_v.seed = nullptr;
  if (dxm) dr_v = *drop;
  PQ_CHECK_ARG(x && w && rstd && dy && dx && dw && R >= 0 && d >= 1 && d <= 64 * MAXPL, "pq3d_rmsnorm_bwd: bad args");
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate) {
    ZeroList z;
    z.add(dw, d);
    if (int e = pq3d_zero_launch(z, s)) return e;
  }
  if (R == 0) return 0;
  // one row per wave (rows-per-wave sweep at config 5, 19 calls of 512 rows x 512: 1 row per wave 194 us per step, 2: 282, 4: 503 --
  // the rows of a wave are dependent load -> reduce -> store chains; the extra parameter-gradient atomics cost less)
  constexpr int rms_rpw = 1;
  long nb = (R + rms_rpw * WPB - 1) / (rms_rpw * WPB);
  if (nb > 1024) nb = 1024;
  dim3 grid((unsigned)nb);
  RMS_DISPATCH(rmsnorm_bwd_kernel, grid, x, w, rstd, dy, dx, dw, (long)R, d, dres, dr_v, dxm)
  PQ_LAUNCH_CHECK();
  return 0;
}
