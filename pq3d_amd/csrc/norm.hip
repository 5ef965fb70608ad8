// Residual-add + LayerNorm (merged over M branches), forward and backward.  HBM-bound: one wave per row,
// the row lives in registers (d <= 1024 -> <= 16 values per lane), wave shuffles for the statistics,
// fp32 math throughout.  Algorithmic bytes per row: (1 + M) reads + 1 write of d elements.
#include "common.h"

namespace {

constexpr int MAXPL = 16;  // values per lane -> d <= 1024
constexpr int WPB = 4;     // waves (rows in flight) per block

struct RowStats { float mean, rstd; };

template <int PL>
PQ_DEV RowStats row_stats(const float (&v)[PL], int d, int lane, float eps) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < PL; ++j) s += (lane + 64 * j < d) ? v[j] : 0.f;
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < PL; ++j) {
    const float t = v[j] - mean;
    q += (lane + 64 * j < d) ? t * t : 0.f;
  }
  const float var = wave_sum(q) / (float)d;
  return {mean, 1.f / sqrtf(var + eps)};
}

template <int PL>
__global__ __launch_bounds__(WPB * 64) void add_ln_fwd_kernel(const pq3d_ln_desc d) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * WPB + (threadIdx.x >> 6);
  if (row >= d.R) return;
  const long base = row * d.d;
  float xr[PL], y[PL];
#pragma unroll
  for (int j = 0; j < PL; ++j) {
    const int c = lane + 64 * j;
    xr[j] = (d.x && c < d.d) ? load_elem(d.x, d.dt_x, base + c) : 0.f;
    y[j] = 0.f;
  }
  const long scene = row / d.rows_per_scene, nscene = d.R / d.rows_per_scene;
  // sum_branches: the M inputs are PARTIAL SUMS of one branch (deterministic K-split of the producing GEMM): one
  // LayerNorm of x + dropout(sum_m o_m), statistics / gamma / beta of index 0
  const int mlo = d.independent ? blockIdx.y : 0, mhi = d.independent ? blockIdx.y + 1 : (d.sum_branches ? 1 : d.M);
  const bool drop = drop_on(d.drop);
  for (int m = mlo; m < mhi; ++m) {
    float v[PL], ov[PL];
#pragma unroll
    for (int j = 0; j < PL; ++j) {
      const int c = lane + 64 * j;
      ov[j] = c < d.d ? load_elem(d.o[m], d.dt_o, base + c) : 0.f;
    }
    if (d.sum_branches) {
      for (int p = 1; p < d.M; ++p)
#pragma unroll
        for (int j = 0; j < PL; ++j) {
          const int c = lane + 64 * j;
          ov[j] += c < d.d ? load_elem(d.o[p], d.dt_o, base + c) : 0.f;
        }
      if (d.osum) {   // the summed branch, kept for the backward pass (which then reads one tensor instead of M)
#pragma unroll
        for (int j = 0; j < PL; ++j) {
          const int c = lane + 64 * j;
          if (c < d.d) d.osum[base + c] = ov[j];
        }
      }
    }
    if (drop) {   // uniform branch around pure ALU: residual dropout of branch m (site drop.site + m)
      const DropState ds = drop_init(d.drop, m, d.d);
#pragma unroll
      for (int j = 0; j < PL; ++j) ov[j] = drop_keep(ds, (uint32_t)row, (uint32_t)(lane + 64 * j)) ? ov[j] * ds.scale : 0.f;
    }
#pragma unroll
    for (int j = 0; j < PL; ++j) v[j] = (lane + 64 * j < d.d) ? xr[j] + ov[j] : 0.f;
    const RowStats st = row_stats<PL>(v, d.d, lane, d.eps);
    const float w = (d.independent || d.sum_branches) ? 1.f : (d.coef ? d.coef[m * nscene + scene] : 1.f / (float)d.M);
#pragma unroll
    for (int j = 0; j < PL; ++j) {
      const int c = lane + 64 * j;
      if (c < d.d) y[j] += w * ((v[j] - st.mean) * st.rstd * d.gamma[m][c] + d.beta[m][c]);
    }
    if (lane == 0) {
      d.mean[(long)m * d.R + row] = st.mean;
      d.rstd[(long)m * d.R + row] = st.rstd;
    }
  }
  void* yout = d.independent ? d.ys[blockIdx.y] : d.y;
#pragma unroll
  for (int j = 0; j < PL; ++j) {
    const int c = lane + 64 * j;
    if (c < d.d) store_elem(yout, d.dt_y, base + c, y[j]);
  }
}

// Backward: each wave walks rows with a grid stride, keeps per-lane partial dgamma/dbeta for its columns in
// registers for all M branches is too much (M*PL*2) -> loop branches outermost within a row, accumulate the
// parameter grads of one branch at a time into LDS-free registers by making m the OUTER loop of the kernel
// (blockIdx.y = m).  dx (sum over branches) is then accumulated with atomics only when M > 1.
template <int PL>
__global__ __launch_bounds__(WPB * 64) void add_ln_bwd_kernel(const pq3d_ln_desc d) {
  __shared__ float red[2][WPB][64 * PL];
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.y;
  const long wave_id = (long)blockIdx.x * WPB + (threadIdx.x >> 6), nwaves = (long)gridDim.x * WPB;
  const long nscene = d.R / d.rows_per_scene;
  float dg[PL], db[PL], gam[PL];
#pragma unroll
  for (int j = 0; j < PL; ++j) {
    const int c = lane + 64 * j;
    dg[j] = 0.f; db[j] = 0.f;
    gam[j] = c < d.d ? d.gamma[m][c] : 0.f;
  }
  // software-pipelined over rows: the three row loads of the NEXT row are in flight while this row is reduced
  const float* dyp = d.independent ? d.dys[m] : d.dy;
  const bool drop = drop_on(d.drop);
  DropState dst;
  if (drop) dst = drop_init(d.drop, m, d.d);
  float nv[PL], ndy[PL];
  unsigned nkeep = 0xffffffffu;   // bit j: column lane + 64 j of the fetched row survived the residual dropout
  auto fetch = [&](long row) {
    const long base = row * d.d;
    float xv[PL], ov[PL];
#pragma unroll
    for (int j = 0; j < PL; ++j) {
      const int c = lane + 64 * j;
      if (c < d.d) {
        xv[j] = d.x ? load_elem(d.x, d.dt_x, base + c) : 0.f;
        ov[j] = load_elem(d.o[m], d.dt_o, base + c);
        ndy[j] = dyp[base + c];
      } else { xv[j] = 0.f; ov[j] = 0.f; ndy[j] = 0.f; }
    }
    if (d.sum_branches) {
      for (int p = 1; p < d.M; ++p)
#pragma unroll
        for (int j = 0; j < PL; ++j) {
          const int c = lane + 64 * j;
          if (c < d.d) ov[j] += load_elem(d.o[p], d.dt_o, base + c);
        }
    }
    if (drop) {
      nkeep = 0;
#pragma unroll
      for (int j = 0; j < PL; ++j) {
        const bool k = drop_keep(dst, (uint32_t)row, (uint32_t)(lane + 64 * j));
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
      const int c = lane + 64 * j;
      if (c < d.d) {
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
#pragma unroll
    for (int j = 0; j < PL; ++j) {
      const int c = lane + 64 * j;
      if (c < d.d) {
        const float g = rstd * (dz[j] - s1 - xh[j] * s2);
        d.d_o[m][base + c] = drop ? (((keep >> j) & 1u) ? g * dst.scale : 0.f) : g;
        if (d.dx && !d.independent) {
          if (d.M == 1 || d.sum_branches) d.dx[base + c] = g;
          else unsafeAtomicAdd(&d.dx[base + c], g);
        }
      }
    }
  }
  // block-level reduction of the parameter-gradient partials, then ONE atomic per column per block
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < PL; ++j) {
    red[0][wave][lane + 64 * j] = dg[j];
    red[1][wave][lane + 64 * j] = db[j];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < d.d; c += WPB * 64) {
    float sg = 0.f, sb = 0.f;
#pragma unroll
    for (int w = 0; w < WPB; ++w) { sg += red[0][w][c]; sb += red[1][w][c]; }
    unsafeAtomicAdd(&d.dgamma[m][c], sg);
    unsafeAtomicAdd(&d.dbeta[m][c], sb);
  }
}

int check_ln(const pq3d_ln_desc& d, bool bwd) {
  PQ_CHECK_ARG(d.R >= 0 && d.d >= 1 && d.d <= 64 * MAXPL, "pq3d_add_ln: d must be in [1,1024]");
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
  if (d.independent) {
    for (int m = 0; m < d.M; ++m) PQ_CHECK_ARG(bwd ? d.dys[m] != nullptr : d.ys[m] != nullptr, "pq3d_add_ln: null ys/dys");
  } else if (bwd) PQ_CHECK_ARG(d.dy != nullptr, "pq3d_add_ln_bwd: null dy");
  else PQ_CHECK_ARG(d.y != nullptr, "pq3d_add_ln_fwd: null y");
  return 0;
}

#define LN_DISPATCH(kernel, grid)                                                             \
  if (d.d <= 64) hipLaunchKernelGGL((kernel<1>), grid, dim3(WPB * 64), 0, s, d);              \
  else if (d.d <= 128) hipLaunchKernelGGL((kernel<2>), grid, dim3(WPB * 64), 0, s, d);        \
  else if (d.d <= 256) hipLaunchKernelGGL((kernel<4>), grid, dim3(WPB * 64), 0, s, d);        \
  else if (d.d <= 512) hipLaunchKernelGGL((kernel<8>), grid, dim3(WPB * 64), 0, s, d);        \
  else hipLaunchKernelGGL((kernel<16>), grid, dim3(WPB * 64), 0, s, d);

}  // namespace

extern "C" int pq3d_add_ln_fwd(const pq3d_ln_desc* dp, void* stream) {
  PQ_CHECK_ARG(dp != nullptr, "pq3d_add_ln_fwd: null descriptor");
  const pq3d_ln_desc d = *dp;
  if (int e = check_ln(d, false)) return e;
  if (d.R == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)((d.R + WPB - 1) / WPB), d.independent ? d.M : 1);
  LN_DISPATCH(add_ln_fwd_kernel, grid)
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_add_ln_bwd(const pq3d_ln_desc* dp, void* stream) {
  PQ_CHECK_ARG(dp != nullptr, "pq3d_add_ln_bwd: null descriptor");
  const pq3d_ln_desc d = *dp;
  if (int e = check_ln(d, true)) return e;
  hipStream_t s = (hipStream_t)stream;
  {   // zero the atomics targets in one launch (not hipMemsetAsync: see common.h ZeroList)
    ZeroList z;
    for (int m = 0; m < (d.sum_branches ? 1 : d.M) && !d.accumulate; ++m) { z.add(d.dgamma[m], d.d); z.add(d.dbeta[m], d.d); }
    if (z.full()) { if (int e = pq3d_zero_launch(z, s)) return e; z.n = 0; }
    if (d.R > 0 && d.dx && d.M > 1 && !d.independent && !d.sum_branches && !d.dx_zeroed) z.add(d.dx, (long)d.R * d.d);
    if (int e = pq3d_zero_launch(z, s)) return e;
  }
  if (d.R == 0) return 0;
  // two rows per wave (software-pipelined); one atomic per column per block for the parameter gradients
  const int rpw = d.R >= 4096 ? 4 : 2;   // measured: 2 rows/wave is best for R=800, 4 for R=8192 (atomics)
  long nb = (d.R + rpw * WPB - 1) / (rpw * WPB);
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  dim3 grid((unsigned)nb, d.sum_branches ? 1 : d.M);
  LN_DISPATCH(add_ln_bwd_kernel, grid)
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
                                                               float* __restrict__ dx, float* __restrict__ dw, long R, int d) {
  __shared__ float red[WPB][64 * PL];
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
      if (c < d) dx[row * d + c] = r * (g[j] - xh[j] * s);
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
  else hipLaunchKernelGGL((kernel<16>), grid, dim3(WPB * 64), 0, s, __VA_ARGS__);

extern "C" int pq3d_rmsnorm_fwd(const float* x, const float* w, float* y, float* rstd, int64_t R, int32_t d, float eps,
                                void* stream) {
  PQ_CHECK_ARG(x && w && y && rstd && R >= 0 && d >= 1 && d <= 64 * MAXPL, "pq3d_rmsnorm_fwd: bad args (d <= 1024)");
  if (R == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)((R + WPB - 1) / WPB));
  RMS_DISPATCH(rmsnorm_fwd_kernel, grid, x, w, y, rstd, (long)R, d, eps)
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_rmsnorm_bwd(const float* x, const float* w, const float* rstd, const float* dy, float* dx, float* dw,
                                int64_t R, int32_t d, int32_t accumulate, void* stream) {
  PQ_CHECK_ARG(x && w && rstd && dy && dx && dw && R >= 0 && d >= 1 && d <= 64 * MAXPL, "pq3d_rmsnorm_bwd: bad args");
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate) {
    ZeroList z;
    z.add(dw, d);
    if (int e = pq3d_zero_launch(z, s)) return e;
  }
  if (R == 0) return 0;
  long nb = (R + 2 * WPB - 1) / (2 * WPB);
  if (nb > 1024) nb = 1024;
  dim3 grid((unsigned)nb);
  RMS_DISPATCH(rmsnorm_bwd_kernel, grid, x, w, rstd, dy, dx, dw, (long)R, d)
  PQ_LAUNCH_CHECK();
  return 0;
}
