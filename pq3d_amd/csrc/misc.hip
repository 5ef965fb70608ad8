// Small memory-bound kernels of the query-decoder path (all fp32 math, coalesced row-major access).
#include <string.h>

#include "common.h"

namespace {

// ---------------------------------------------------------------- colsum
// grid ceil(N/64); block 1024 = 16 row-lanes x 64 columns; each thread strides over rows with 4 independent loads in
// flight, then a 16-way LDS tree.  No atomics, no memset: deterministic -- except for long ACCUMULATING reductions (R >= 2048
// rows into a running output: bias gradients into the gradient arena), which run as row slices that add atomically.
struct ColsumPtrs { const void* x[PQ3D_MAX_GROUPS]; float* out[PQ3D_MAX_GROUPS]; };
template <typename T>
__global__ __launch_bounds__(1024) void colsum_kernel(const ColsumPtrs cp, long R, long N, long ld, int accumulate) {
  __shared__ float part[16][64];
  const T* x = (const T*)cp.x[blockIdx.y];
  float* out = cp.out[blockIdx.y];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const long col = (long)blockIdx.x * 64 + cx;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  // row slices (gridDim.z > 1; accumulating calls only): a [10240, 768] gradient is 12 column blocks -- a handful of
  // workgroups walking 10240 rows each (55 us at the shipped stage-2 shape); the slices add their partial sums atomically
  const long rchunk = gridDim.z > 1 ? (((R + gridDim.z - 1) / gridDim.z + 15) & ~15L) : R;
  const long r_lo = (long)blockIdx.z * rchunk;
  x += r_lo * ld;
  R = max(0L, min(R - r_lo, rchunk));
  if (col < N) {
    long r = ry;
    for (; r + 48 < R; r += 64) {
      const float a = Cvt<T>::to(x[r * ld + col]), b = Cvt<T>::to(x[(r + 16) * ld + col]);
      const float c = Cvt<T>::to(x[(r + 32) * ld + col]), e = Cvt<T>::to(x[(r + 48) * ld + col]);
      s0 += a; s1 += b; s2 += c; s3 += e;
    }
    for (; r < R; r += 16) s0 += Cvt<T>::to(x[r * ld + col]);
  }
  part[ry][cx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ry == 0 && col < N) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += part[k][cx];
    if (gridDim.z > 1) unsafeAtomicAdd(&out[col], t);
    else out[col] = accumulate ? out[col] + t : t;
  }
}

// ---------------------------------------------------------------- scale rows
__global__ void scale_rows_kernel(const void* x, int dtx, void* y, int dty, long R, long C, const float* scale,
                                  const uint8_t* zero_flag, const uint8_t* keep_mask) {
  const long total = R * C;
  if (dtx == PQ3D_F32 && (C & 7) == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {   // 8 elements of one row per thread
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < total; i += (long)gridDim.x * blockDim.x * 8) {
      const long r = i / C;
      const bool keep = (!zero_flag || !zero_flag[r]) && (!keep_mask || keep_mask[r]);
      const float sc = keep ? (scale ? scale[r] : 1.f) : 0.f;
      const float4 a = *(const float4*)((const float*)x + i), b = *(const float4*)((const float*)x + i + 4);
      // a dropped row is exactly zero (also for non-finite inputs)
      const float v[8] = {keep ? a.x * sc : 0.f, keep ? a.y * sc : 0.f, keep ? a.z * sc : 0.f, keep ? a.w * sc : 0.f,
                          keep ? b.x * sc : 0.f, keep ? b.y * sc : 0.f, keep ? b.z * sc : 0.f, keep ? b.w * sc : 0.f};
      if (dty == PQ3D_BF16) *(u32x4*)((bf16_t*)y + i) = pack_frag<bf16_t>(v);
      else {
        *(float4*)((float*)y + i) = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)((float*)y + i + 4) = make_float4(v[4], v[5], v[6], v[7]);
      }
    }
    return;
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / C;
    const bool keep = (!zero_flag || !zero_flag[r]) && (!keep_mask || keep_mask[r]);
    const float v = keep ? load_elem(x, dtx, i) * (scale ? scale[r] : 1.f) : 0.f;
    store_elem(y, dty, i, v);
  }
}

struct AddCastPtrs { const float* a[PQ3D_MAX_GROUPS]; const float* b[PQ3D_MAX_GROUPS]; void* out[PQ3D_MAX_GROUPS]; };
// 8 elements per thread: two float4 loads per input, one 16-byte (bf16) or two 16-byte (fp32) stores
__global__ void add_cast_kernel(const AddCastPtrs p, int dt_out, long n) {
  const float* a = p.a[blockIdx.y];
  const float* b = p.b[blockIdx.y];
  void* out = p.out[blockIdx.y];
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += (long)gridDim.x * blockDim.x * 8) {
    float v[8];
    if (i + 8 <= n) {
      const float4 a0 = *(const float4*)(a + i), a1 = *(const float4*)(a + i + 4);
      v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
      if (b) {
        const float4 b0 = *(const float4*)(b + i), b1 = *(const float4*)(b + i + 4);
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
      }
      if (dt_out == PQ3D_BF16)
        *(u32x4*)((bf16_t*)out + i) = (u32x4){pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
      else {
        *(float4*)((float*)out + i) = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)((float*)out + i + 4) = make_float4(v[4], v[5], v[6], v[7]);
      }
    } else {
      for (long j = i; j < n; ++j) store_elem(out, dt_out, j, a[j] + (b ? b[j] : 0.f));
    }
  }
}

// hi / lo bf16 planes of (a + b): hi = bf16(v) (round to nearest even), lo = bf16(v - hi) -- the operand form of the split-bf16
// products whose operands are reused by many launches (compute mode 'bf16x3': the memories' tokens and the K / V weights);
// groups of individual lengths, hi may be NULL for a group (its hi plane already exists)
struct SplitPtrs { const float* a[PQ3D_MAX_GROUPS]; const float* b[PQ3D_MAX_GROUPS]; void* hi[PQ3D_MAX_GROUPS]; void* lo[PQ3D_MAX_GROUPS];
                   long n[PQ3D_MAX_GROUPS]; };
__global__ void split_planes_kernel(const SplitPtrs p) {
  const int g = blockIdx.y;
  const float* a = p.a[g];
  const float* b = p.b[g];
  bf16_t* hi = (bf16_t*)p.hi[g];
  bf16_t* lo = (bf16_t*)p.lo[g];
  const long n = p.n[g];
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += (long)gridDim.x * blockDim.x * 8) {
    const float4 a0 = *(const float4*)(a + i), a1 = *(const float4*)(a + i + 4);
    float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    if (b) {
      const float4 b0 = *(const float4*)(b + i), b1 = *(const float4*)(b + i + 4);
      v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
    }
    u32x4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      h[j] = pack_bf2(v[2 * j], v[2 * j + 1]);
      l[j] = pack_bf2(v[2 * j] - __uint_as_float(h[j] << 16), v[2 * j + 1] - __uint_as_float(h[j] & 0xffff0000u));
    }
    if (hi) *(u32x4*)(hi + i) = h;
    *(u32x4*)(lo + i) = l;
  }
}

__global__ void bias_add_rows_kernel(const float* x, const float* bias, float* out, long R, long N) {
  const long n4 = N / 4, total = R * n4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long c = (i % n4) * 4;
    const float4 a = *(const float4*)(x + i * 4), b = *(const float4*)(bias + c);
    *(float4*)(out + i * 4) = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  }
}

__global__ void act_bwd_kernel(const void* dy, int dt_dy, const void* saved, int dt_s, void* dpre, int dt_o, int act,
                               long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float g = load_elem(dy, dt_dy, i), sv = load_elem(saved, dt_s, i);
    float r = g;
    if (act == PQ3D_ACT_RELU) r = sv > 0.f ? g : 0.f;
    else if (act == PQ3D_ACT_GELU) r = g * gelu_grad_f(sv);
    store_elem(dpre, dt_o, i, r);
  }
}

__global__ void fill_cols_kernel(const float* x, float* y, long R, long C, const int32_t* cols, int ncols, float value) {
  const long total = R * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    bool hit = false;
    for (int k = 0; k < ncols; ++k) hit |= (cols[k] == c);
    y[i] = hit ? value : x[i];
  }
}

struct MaskPtrs { const uint8_t* p[PQ3D_MAX_GROUPS]; };
__global__ void mask_inv_den_kernel(MaskPtrs mp, int M, long n, float* inv_den) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float den = 0.f;
  for (int m = 0; m < M; ++m) den += mp.p[m][i] ? 0.f : 1.f;
  inv_den[i] = 1.f / (den + 1e-8f);
}

// ---------------------------------------------------------------- pairwise locs
// grid (ceil(L/8), B), block 256.  Every block recomputes the scene's max distance (L*L <= 4e4 pairs).
__global__ void pairwise_locs_kernel(const float* centers, long cs, float* out, int L, float eps) {
  __shared__ float red[4];
  extern __shared__ float cen[];  // [L][3]
  const int b = blockIdx.y, tid = threadIdx.x;
  const float* c = centers + (long)b * L * cs;
  for (int i = tid; i < L * 3; i += blockDim.x) cen[i] = c[(long)(i / 3) * cs + (i % 3)];
  __syncthreads();
  // max over all pairs of sqrt(d2 + eps) = sqrt(max d2 + eps) (sqrtf is monotonic and correctly rounded: the same bits).
  // Thread t walks row i = t, t + 256, ... against every j: no integer division, the j operands are LDS broadcasts
  // (one thread per PAIR with p / L, p % L and a square root per pair was 11 us at L = 100, 34 us at L = 200)
  float m2 = 0.f;
  // rows x column quarters: thread t takes row t / 4 (+ 64 k) against the columns j = t % 4 (mod 4) -- four times the threads
  // of a row-per-thread walk at L <= 64 k, the j operands still few distinct LDS addresses per wave
  for (int i = tid >> 2; i < L; i += blockDim.x >> 2) {
    const float xi = cen[3 * i], yi = cen[3 * i + 1], zi = cen[3 * i + 2];
#pragma unroll 4
    for (int j = tid & 3; j < L; j += 4) {
      const float dx = xi - cen[3 * j], dy = yi - cen[3 * j + 1], dz = zi - cen[3 * j + 2];
      m2 = fmaxf(m2, dx * dx + dy * dy + dz * dz + eps);
    }
  }
  float mx = L > 0 ? sqrtf(fmaxf(m2, eps)) : 0.f;
  mx = wave_max(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const int i0 = blockIdx.x * 8;
  for (int p = tid; p < 8 * L; p += blockDim.x) {
    const int i = i0 + p / L, j = p % L;
    if (i >= L) break;
    const float dx = cen[3 * i] - cen[3 * j], dy = cen[3 * i + 1] - cen[3 * j + 1], dz = cen[3 * i + 2] - cen[3 * j + 2];
    const float dist = sqrtf(dx * dx + dy * dy + dz * dz + eps);
    const float d2 = sqrtf(dx * dx + dy * dy + eps);
    float* o = out + (((long)b * L + i) * L + j) * 5;
    o[0] = dist / mx; o[1] = dz / dist; o[2] = d2 / dist; o[3] = dy / d2; o[4] = dx / d2;
  }
}

// ---------------------------------------------------------------- fourier features
__global__ void fourier_kernel(const float* xyz, long xs, const float* cmin, const float* cmax, const float* G,
                               float* out, int B, int N, int half) {
  const long total = (long)B * N * half;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i % half);
    const long pn = i / half;
    const int b = (int)(pn / N);
    const float* p = xyz + pn * xs;
    float proj = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float t = (p[c] - cmin[b * 3 + c]) / (cmax[b * 3 + c] - cmin[b * 3 + c]);
      t *= 6.283185307179586f;
      proj += t * G[c * half + j];
    }
    float sn, cs;
    sincosf(proj, &sn, &cs);   // one argument reduction for both
    out[pn * 2 * half + j] = sn;
    out[pn * 2 * half + half + j] = cs;
  }
}

// two point sets of the same scenes (queries, segments) in one launch: out = [B * Na rows of set a | B * Nb rows of set b]
__global__ void fourier_pair_kernel(const float* xa, long sa, int Na, const float* xb, long sb, int Nb, const float* cmin,
                                    const float* cmax, const float* G, float* out, int B, int half) {
  // one division per THREAD (its row within the block, its column), none per element: threads (row, j) of a block cover
  // blockDim / half rows (or a column slice of one row when half > blockDim); the per-element 64-bit i / half, i % half and
  // q / N of a flat index were most of the 12 us this kernel took at config 2
  const int ra = B * Na, rows = ra + B * Nb;
  const int tpr = half < (int)blockDim.x ? half : (int)blockDim.x;   // threads per row
  const int rpb = (int)blockDim.x / tpr;                              // rows per block pass
  const int rl = (int)threadIdx.x / tpr, j0 = (int)threadIdx.x % tpr;
  for (int pn = (int)blockIdx.x * rpb + rl; pn < rows; pn += (int)gridDim.x * rpb) {
    if (rl >= rpb) break;
    const bool isa = pn < ra;
    const int q = isa ? pn : pn - ra;
    const int b = q / (isa ? Na : Nb);
    const float* p = isa ? xa + (long)q * sa : xb + (long)q * sb;
    float t[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      t[c] = (p[c] - cmin[b * 3 + c]) / (cmax[b * 3 + c] - cmin[b * 3 + c]);
      t[c] *= 6.283185307179586f;
    }
    float* o = out + (long)pn * 2 * half;
    for (int j = j0; j < half; j += tpr) {
      float proj = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) proj += t[c] * G[c * half + j];
      float sn, cs;
      sincosf(proj, &sn, &cs);   // one argument reduction for both
      o[j] = sn;
      o[half + j] = cs;
    }
  }
}

// ---------------------------------------------------------------- spatial bias
struct SbFwdGroups {
  const float* W[PQ3D_MAX_GROUPS];
  const float* bw[PQ3D_MAX_GROUPS];
  float* bias[PQ3D_MAX_GROUPS];
};
__global__ void spatial_bias_fwd_kernel(const float* pl, const SbFwdGroups gr, int B, int H, int L) {
  const float* W = gr.W[blockIdx.y];
  const float* bw = gr.bw[blockIdx.y];
  float* bias = gr.bias[blockIdx.y];
  const long LL = (long)L * L, total = (long)B * LL;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / LL, ij = i % LL;
    float f[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) f[c] = pl[i * 5 + c];
    for (int h = 0; h < H; ++h) {
      float v = bw[h];
#pragma unroll
      for (int c = 0; c < 5; ++c) v += W[h * 5 + c] * f[c];
      bias[(b * H + h) * LL + ij] = logf(fmaxf(fmaxf(v, 0.f), 1e-6f));
    }
  }
}

// Single pass: each thread reads its (b,i,j) feature vector ONCE and updates all heads' 6 partial sums in registers
// (H <= 16), then one wave reduction + LDS tree per block and 6*H atomics per block.  (The first version looped
// over heads outside the element loop and re-read pl H times: 29 us for 80k elements.)
// (measured: capping the grid at 64 blocks to cut the 6*H contended atomics per block made it slower, 20 -> 28 us: one
// element per thread is what hides the load latency; the layers of a step are batched into one launch instead)
constexpr long SB_BLOCKS = 512;
struct SbGroups {
  const float* W[PQ3D_MAX_GROUPS];
  const float* bw[PQ3D_MAX_GROUPS];
  const float* dbias[PQ3D_MAX_GROUPS];
  float* dW[PQ3D_MAX_GROUPS];
  float* dbw[PQ3D_MAX_GROUPS];
};
template <int HMAX>
__global__ __launch_bounds__(256) void spatial_bias_bwd_kernel(const float* pl, const SbGroups gr, int B, int H, int L) {
  const float* W = gr.W[blockIdx.y];
  const float* bw = gr.bw[blockIdx.y];
  const float* dbias = gr.dbias[blockIdx.y];
  float* dW = gr.dW[blockIdx.y];
  float* dbw = gr.dbw[blockIdx.y];
  __shared__ float red[4][HMAX * 6];
  const long LL = (long)L * L, total = (long)B * LL;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc[HMAX][6];
  float w[HMAX][6];
#pragma unroll
  for (int h = 0; h < HMAX; ++h)
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      acc[h][c] = 0.f;
      w[h][c] = h < H ? (c < 5 ? W[h * 5 + c] : bw[h]) : 0.f;
    }
  // SBU elements per thread, every load of all of them requested before the first is used: a quarter of the blocks (and of the
  // 6 H contended atomics per block) at the same number of bytes in flight (one element per thread: 19.6 us at config 2)
  constexpr int SBU = 4;
  const long st = (long)gridDim.x * blockDim.x;
  for (long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x; i0 < total; i0 += st * SBU) {
    float f[SBU][5], db_[SBU][HMAX];
    bool ok[SBU];
#pragma unroll
    for (int u = 0; u < SBU; ++u) {
      const long i = i0 + u * st;
      ok[u] = i < total;
      const long ic = ok[u] ? i : 0;
      const long b = ic / LL, ij = ic % LL;
#pragma unroll
      for (int c = 0; c < 5; ++c) f[u][c] = pl[ic * 5 + c];
#pragma unroll
      for (int h = 0; h < HMAX; ++h) db_[u][h] = h < H ? dbias[(b * H + h) * LL + ij] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < SBU; ++u) {
      if (!ok[u]) continue;
#pragma unroll
      for (int h = 0; h < HMAX; ++h) {
        if (h < H) {
          float v = w[h][5];
#pragma unroll
          for (int c = 0; c < 5; ++c) v += w[h][c] * f[u][c];
          const float g = v > 1e-6f ? db_[u][h] / v : 0.f;
#pragma unroll
          for (int c = 0; c < 5; ++c) acc[h][c] += g * f[u][c];
          acc[h][5] += g;
        }
      }
    }
  }
#pragma unroll
  for (int h = 0; h < HMAX; ++h)
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const float sres = wave_sum(acc[h][c]);
      if (lane == 0) red[wave][h * 6 + c] = sres;
    }
  __syncthreads();
  if (threadIdx.x < H * 6) {
    const int t = threadIdx.x, h = t / 6, c = t % 6;
    const float sres = red[0][t] + red[1][t] + red[2][t] + red[3][t];
    if (c < 5) unsafeAtomicAdd(&dW[h * 5 + c], sres);
    else unsafeAtomicAdd(&dbw[h], sres);
  }
}

// ---------------------------------------------------------------- gate mix
__global__ void gate_fwd_kernel(const float* q, const float* u, const float* g, float* y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float s = 1.f / (1.f + expf(-g[i]));
    y[i] = (1.f - s) * q[i] + s * u[i];
  }
}
__global__ void gate_bwd_kernel(const float* q, const float* u, const float* g, const float* dy, float* dq, float* du,
                                float* dg, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float s = 1.f / (1.f + expf(-g[i]));
    dq[i] = (1.f - s) * dy[i];
    du[i] = s * dy[i];
    dg[i] = dy[i] * (u[i] - q[i]) * s * (1.f - s);
  }
}

// (segment pooling: segment.hip)

inline unsigned grid1d(long total, int block = 256, long cap = 4096) {
  long g = (total + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (unsigned)g;
}
inline int memset_async(void* p, size_t bytes, hipStream_t s) {   // fp32 buffers only (bytes % 4 == 0)
  ZeroList z;
  z.add(p, (long)(bytes / 4));
  return pq3d_zero_launch(z, s);
}

__global__ __launch_bounds__(256) void zero_kernel(const ZeroList z) {
  float* p = z.ptr[blockIdx.y];
  const long n = z.count[blockIdx.y];
  long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  const long stride = (long)gridDim.x * 1024;
  if ((((uintptr_t)p) & 15) == 0) {
    for (; i + 3 < n; i += stride) *(float4*)(p + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    for (; i < n; i += stride)
      for (long j = i; j < min(i + 4, n); ++j) p[j] = 0.f;
  } else {
    for (; i < n; i += stride)
      for (long j = i; j < min(i + 4, n); ++j) p[j] = 0.f;
  }
}

}  // namespace

int pq3d_zero_launch(const ZeroList& z, hipStream_t s) {
  if (z.n == 0) return 0;
  long mx = 0;
  for (int i = 0; i < z.n; ++i) mx = z.count[i] > mx ? z.count[i] : mx;
  hipLaunchKernelGGL(zero_kernel, dim3(grid1d((mx + 3) / 4, 256, 1024), z.n), dim3(256), 0, s, z);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { pq3d_set_error(hipGetErrorString(e)); return (int)e; }
  return 0;
}


// ---- small glue ops that used to be at::native launches inside the captured step (~17 of them) ----------
namespace {
struct NotPtrs { const uint8_t* src[PQ3D_MAX_GROUPS]; uint8_t* dst[PQ3D_MAX_GROUPS]; long n[PQ3D_MAX_GROUPS]; };
// dst_g[i] = !src_g[i] for up to PQ3D_MAX_GROUPS byte masks of different lengths in one launch ('True = valid' pad masks of
// data_dict -> PyTorch's 'True = ignore', query3d_unified.py:113,139,143,148,155)
__global__ void mask_not_kernel(const NotPtrs p) {
  const uint8_t* s = p.src[blockIdx.y];
  uint8_t* d = p.dst[blockIdx.y];
  const long n = p.n[blockIdx.y];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) d[i] = s[i] ? 0 : 1;
}

struct SumPtrs { const float* src[PQ3D_MAX_GROUPS]; };
__global__ void sum_n_kernel(const SumPtrs p, int n, float* out, long cnt) {   // out = sum_g src_g (fixed order)
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < cnt; i += (long)gridDim.x * blockDim.x * 4) {
    float4 a = *(const float4*)(p.src[0] + i);
    for (int g = 1; g < n; ++g) {
      const float4 b = *(const float4*)(p.src[g] + i);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    *(float4*)(out + i) = a;
  }
}

// two sums of different lengths in one launch (blockIdx.y selects the job)
struct SumPair { SumPtrs p[2]; int n[2]; float* out[2]; long cnt[2]; };
__global__ void sum_pair_kernel(const SumPair q) {
  const int y = blockIdx.y;
  const float* const* src = q.p[y].src;
  const int n = q.n[y];
  float* out = q.out[y];
  const long cnt = q.cnt[y];
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < cnt; i += (long)gridDim.x * blockDim.x * 4) {
    float4 a = *(const float4*)(src[0] + i);
    for (int g = 1; g < n; ++g) {
      const float4 b = *(const float4*)(src[g] + i);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    *(float4*)(out + i) = a;
  }
}

// mean of all elements, one launch: every block reduces a slice (float4 loads, all in flight), publishes its partial sum,
// and the block that arrives LAST (device-scope ticket) adds the partials with a fixed tree -> deterministic.  ws[0] is the
// ticket counter (zero before the first use; the last block resets it), ws[1..] the partials.
__global__ __launch_bounds__(256) void mean_all_kernel(const float* x, long n, float* out, float* ws) {
  __shared__ float red[4];
  __shared__ int last;
  const long n4 = n >> 2;
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = ((const float4*)x)[i];
    s += (v.x + v.y) + (v.z + v.w);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) s += x[(n4 << 2) + threadIdx.x];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_store(&ws[1 + blockIdx.x], (red[0] + red[1]) + (red[2] + red[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add((unsigned*)ws, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = (t == gridDim.x - 1);
  }
  __syncthreads();
  if (!last) return;
  // the last block: one partial per thread (gridDim.x <= 256 = blockDim.x), all loads in flight at once, fixed tree
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  float p = threadIdx.x < gridDim.x ? __hip_atomic_load(&ws[1 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
  p = wave_sum(p);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = p;
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)n;
    __hip_atomic_store((unsigned*)ws, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__global__ void fill_scaled_kernel(float* dst, long n, const float* scalar, float c) {   // dst[i] = scalar[0] * c
  const float v = scalar[0] * c;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = v;
}


// ---- sum over up to 32 tensors of mean(f_g(x_g)) in ONE launch, and its gradient in one more: the synthetic loss of the
// mask configurations (sum over prediction layers of mean(clamp(mask_logits, -50)) + mean(class logits with
// the filtered -inf columns dropped)) cost ~14 framework launches per prediction layer (isfinite / where / clamp / mean /
// add forward, fills and masks backward).  f: 0 identity, 1 max(x, c) (clamp(min=c)), 2 x if finite else 0.
// grid = (MEAN_MANY_BLOCKS, tensors); deterministic: fixed per-block slices, partials combined by the last arriver in
// (tensor, block) order.  ws[0] = ticket (zero before first use; reset by the last block), ws[1..] partials.
#define MEAN_MANY_BLOCKS 128   /* per tensor (32 with one load in flight per thread streamed config 4's 65 MB at under 1 TB/s) */
struct MeanMany { int n; const float* x[PQ3D_MAX_GROUPS]; float* dx[PQ3D_MAX_GROUPS]; long count[PQ3D_MAX_GROUPS]; int mode[PQ3D_MAX_GROUPS]; float cmin[PQ3D_MAX_GROUPS]; };
PQ_DEV float mean_many_f(float v, int mode, float c) { return mode == 1 ? fmaxf(v, c) : (mode == 2 ? (isfinite(v) ? v : 0.f) : v); }
__global__ __launch_bounds__(256) void mean_many_kernel(const MeanMany m, float* out, float* ws) {
  __shared__ float red[4];
  __shared__ int last;
  const int g = blockIdx.y;
  const float* x = m.x[g];
  const long n = m.count[g], n4 = n >> 2;
  const int mode = m.mode[g];
  const float c = m.cmin[g];
  float s = 0.f;
  if ((((uintptr_t)x) & 15) == 0) {
    const long st = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * st < n4; i += 4 * st) {   // four 16-byte loads in flight per thread; fixed order of additions
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = ((const float4*)x)[i + u * st];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        s += (mean_many_f(v[u].x, mode, c) + mean_many_f(v[u].y, mode, c)) + (mean_many_f(v[u].z, mode, c) + mean_many_f(v[u].w, mode, c));
    }
    for (; i < n4; i += st) {
      const float4 v = ((const float4*)x)[i];
      s += (mean_many_f(v.x, mode, c) + mean_many_f(v.y, mode, c)) + (mean_many_f(v.z, mode, c) + mean_many_f(v.w, mode, c));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) s += mean_many_f(x[(n4 << 2) + threadIdx.x], mode, c);
  } else {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += mean_many_f(x[i], mode, c);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  // partial sums only; a second, one-block launch combines them (mean_many_combine_kernel).  An in-kernel last-arriver
  // combine (release fence + ticket per block) was measured at ~10 us per 13 MB tensor whatever the block count (1.3 TB/s:
  // the agent-scope release of every block writes the XCD's L2 back) against one more ~5 us launch here.
  if (threadIdx.x == 0) ws[1 + 2 * PQ3D_MAX_GROUPS + g * MEAN_MANY_BLOCKS + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  (void)out; (void)last;
}
// one block: wave w sums the partials of tensors w, w + 4, ... (all requested at once, fixed tree), thread 0 adds the
// tensors in index order -> deterministic
__global__ __launch_bounds__(256) void mean_many_combine_kernel(const MeanMany m, float* out, const float* ws, int nblocks) {
  __shared__ float tsum[PQ3D_MAX_GROUPS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float* part = ws + 1 + 2 * PQ3D_MAX_GROUPS;
  for (int t = wv; t < m.n; t += 4) {
    float p = 0.f;
    for (int b = lane; b < nblocks; b += 64) p += part[t * MEAN_MANY_BLOCKS + b];
    p = wave_sum(p);
    if (lane == 0) tsum[t] = p / (float)m.count[t];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float q = 0.f;
    for (int t = 0; t < m.n; ++t) q += tsum[t];
    out[0] = q;
  }
}
__global__ __launch_bounds__(256) void mean_many_bwd_kernel(const MeanMany m, const float* gout) {
  const int g = blockIdx.y;
  const float* x = m.x[g];
  float* dx = m.dx[g];
  const long n = m.count[g];
  const int mode = m.mode[g];
  const float c = m.cmin[g], sc = gout[0] / (float)n;
  auto f = [&](float v) { return mode == 1 ? (v >= c ? sc : 0.f) : (mode == 2 ? (isfinite(v) ? sc : 0.f) : sc); };
  long i0 = 0;
  if (((((uintptr_t)x) | ((uintptr_t)dx)) & 15) == 0) {   // 16-byte pieces, two in flight per thread
    const long n4 = n >> 2, st = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += 2 * st) {
      const float4 a = ((const float4*)x)[i];
      const bool two = i + st < n4;
      const float4 b = two ? ((const float4*)x)[i + st] : a;
      ((float4*)dx)[i] = make_float4(f(a.x), f(a.y), f(a.z), f(a.w));
      if (two) ((float4*)dx)[i + st] = make_float4(f(b.x), f(b.y), f(b.z), f(b.w));
    }
    i0 = n4 << 2;
  }
  for (long i = i0 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dx[i] = f(x[i]);
}

struct CastTPtrs { const float* src[PQ3D_MAX_GROUPS]; bf16_t* out[PQ3D_MAX_GROUPS]; bf16_t* outT[PQ3D_MAX_GROUPS]; };
// src_g [rows, cols] fp32 -> out_g [rows, cols] bf16 AND outT_g = per (cols x cols) row block transposed:
// outT[t][k][n] = src[t * cols + n][k]  (rows = T * cols).  32 x 32 tiles through LDS: both writes are coalesced.
__global__ __launch_bounds__(256) void cast_transpose_kernel(const CastTPtrs p, int rows, int cols) {
  __shared__ float tile[32][33];
  const float* src = p.src[blockIdx.z];
  bf16_t* out = p.out[blockIdx.z];
  bf16_t* outT = p.outT[blockIdx.z];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    const int r = r0 + j, c = c0 + tx;
    const float v = (r < rows && c < cols) ? src[(long)r * cols + c] : 0.f;
    tile[j][tx] = v;
    if (r < rows && c < cols) out[(long)r * cols + c] = f2bf(v);
  }
  __syncthreads();
  const int t = r0 / cols, n0 = r0 % cols;   // row block t, rows n0.. of it (cols % 32 == 0: a tile never straddles blocks)
  for (int j = ty; j < 32; j += 8) {
    const int k = c0 + j, n = n0 + tx;       // outT[t][k][n] = tile[n - n0][k - c0]
    if (k < cols && r0 + tx < rows) outT[((long)t * cols + k) * cols + n] = f2bf(tile[tx][j]);
  }
}
}  // namespace

extern "C" int pq3d_mask_not(const uint8_t* const* src, uint8_t* const* dst, const int64_t* counts, int32_t groups, void* stream) {
  PQ_DEVICE_GUARD(stream, nullptr);
  PQ_CHECK_ARG(src && dst && counts && groups >= 1 && groups <= PQ3D_MAX_GROUPS, "pq3d_mask_not: bad args");
  NotPtrs p;
  long mx = 0;
  for (int g = 0; g < groups; ++g) {
    PQ_CHECK_ARG(src[g] && dst[g] && counts[g] >= 0, "pq3d_mask_not: null pointer / negative count");
    p.src[g] = src[g]; p.dst[g] = dst[g]; p.n[g] = (long)counts[g];
    mx = counts[g] > mx ? (long)counts[g] : mx;
  }
  if (mx == 0) return 0;
  hipLaunchKernelGGL(mask_not_kernel, dim3(grid1d(mx, 256, 256), groups), dim3(256), 0, (hipStream_t)stream, p);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_zero_many(float* const* bufs, const int64_t* counts, int32_t n, void* stream) {
  PQ_DEVICE_GUARD(stream, nullptr);
  PQ_CHECK_ARG(bufs && counts && n >= 0, "pq3d_zero_many: bad args");
  ZeroList z;
  for (int i = 0; i < n; ++i) {
    PQ_CHECK_ARG(counts[i] == 0 || bufs[i], "pq3d_zero_many: null buffer");
    z.add(bufs[i], (long)counts[i]);
    if (z.full() || i + 1 == n) {
      if (int e = pq3d_zero_launch(z, (hipStream_t)stream)) return e;
      z.n = 0;
    }
  }
  return 0;
}

namespace {
struct CopyList { int n = 0; const float* src[PQ_ZERO_MAX]; float* dst[PQ_ZERO_MAX]; long count[PQ_ZERO_MAX]; };
// dst_g[0 .. count_g) = src_g[..] for up to 64 (pointer, pointer, length) triples in one launch: the gradient pack of the
// data-parallel reducer (parameter gradients -> their slices of the flat bucket), one short copy per parameter
__global__ __launch_bounds__(256) void copy_many_kernel(const CopyList c) {
  const float* s = c.src[blockIdx.y];
  float* d = c.dst[blockIdx.y];
  const long n = c.count[blockIdx.y];
  const bool vec = ((((uintptr_t)s) | ((uintptr_t)d)) & 15) == 0;
  long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  const long stride = (long)gridDim.x * 1024;
  for (; i < n; i += stride) {
    if (vec && i + 3 < n) *(float4*)(d + i) = *(const float4*)(s + i);
    else for (long j = i; j < min(i + 4, n); ++j) d[j] = s[j];
  }
}
}  // namespace

extern "C" int pq3d_copy_many(const float* const* src, float* const* dst, const int64_t* counts, int32_t n, void* stream) {
  PQ_DEVICE_GUARD(stream, nullptr);
  PQ_CHECK_ARG(src && dst && counts && n >= 0, "pq3d_copy_many: bad args");
  CopyList c;
  long mx = 0;
  for (int i = 0; i < n; ++i) {
    PQ_CHECK_ARG(counts[i] == 0 || (src[i] && dst[i]), "pq3d_copy_many: null buffer");
    if (counts[i] > 0) { c.src[c.n] = src[i]; c.dst[c.n] = dst[i]; c.count[c.n] = (long)counts[i]; mx = counts[i] > mx ? (long)counts[i] : mx; ++c.n; }
    if (c.n == PQ_ZERO_MAX || (i + 1 == n && c.n > 0)) {
      hipLaunchKernelGGL(copy_many_kernel, dim3(grid1d((mx + 3) / 4, 256, 256), c.n), dim3(256), 0, (hipStream_t)stream, c);
      PQ_LAUNCH_CHECK();
      c.n = 0; mx = 0;
    }
  }
  return 0;
}

extern "C" int pq3d_sum_n(const float* const* src, int32_t n, float* out, int64_t count, void* stream) {
  PQ_DEVICE_GUARD(stream, nullptr);
  PQ_CHECK_ARG(src && out && n >= 1 && n <= PQ3D_MAX_GROUPS && count >= 0 && (count % 4) == 0, "pq3d_sum_n: bad args (count % 4 == 0)");
  SumPtrs p;
  for (int g = 0; g < n; ++g) {
    PQ_CHECK_ARG(src[g] && ((((uintptr_t)src[g]) & 15) == 0), "pq3d_sum_n: inputs must be 16-byte aligned");
    p.src[g] = src[g];
  }
  PQ_CHECK_ARG((((uintptr_t)out) & 15) == 0, "pq3d_sum_n: out must be 16-byte aligned");
  if (count == 0) return 0;
  hipLaunchKernelGGL(sum_n_kernel, dim3(grid1d(count / 4, 256, 1024)), dim3(256), 0, (hipStream_t)stream, p, n, out, (long)count);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_sum_pair(const float* const* src_a, int32_t na, float* out_a, int64_t count_a, const float* const* src_b,
                             int32_t nb, float* out_b, int64_t count_b, void* stream) {
  PQ_DEVICE_GUARD(stream, nullptr);
  PQ_CHECK_ARG(src_a && src_b && out_a && out_b && na >= 1 && nb >= 1 && na <= PQ3D_MAX_GROUPS && nb <= PQ3D_MAX_GROUPS &&
                   count_a >= 0 && count_b >= 0 && (count_a % 4) == 0 && (count_b % 4) == 0,
               "pq3d_sum_pair: bad args (counts % 4 == 0)");
  SumPair q;
  for (int g = 0; g < na; ++g) {
    PQ_CHECK_ARG(src_a[g] && ((((uintptr_t)src_a[g]) & 15) == 0), "pq3d_sum_pair: inputs must be 16-byte aligned");
    q.p[0].src[g] = src_a[g];
  }
  for (int g = 0; g < nb; ++g) {
    PQ_CHECK_ARG(src_b[g] && ((((uintptr_t)src_b[g]) & 15) == 0), "pq3d_sum_pair: inputs must be 16-byte aligned");
    q.p[1].src[g] = src_b[g];
  }
  PQ_CHECK_ARG(((((uintptr_t)out_a) | ((uintptr_t)out_b)) & 15) == 0, "pq3d_sum_pair: outputs must be 16-byte aligned");
  q.n[0] = na; q.n[1] = nb; q.out[0] = out_a; q.out[1] = out_b; q.cnt[0] = count_a; q.cnt[1] = count_b;
  const long big = count_a > count_b ? count_a : count_b;
  if (big == 0) return 0;
  hipLaunchKernelGGL(sum_pair_kernel, dim3(grid1d(big / 4, 256, 1024), 2), dim3(256), 0, (hipStream_t)stream, q);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_mean_all(const float* x, int64_t n, float* out, float* ws, void* stream) {
  PQ_DEVICE_GUARD(stream, x);
  PQ_CHECK_ARG(x && out && ws && n >= 1, "pq3d_mean_all: bad args");
  PQ_CHECK_ARG((((uintptr_t)x) & 15) == 0, "pq3d_mean_all: x must be 16-byte aligned");
  long nb = (n / 4 + 255) / 256;
  if (nb > PQ3D_MEAN_MAX_BLOCKS) nb = PQ3D_MEAN_MAX_BLOCKS;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(mean_all_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, (long)n, out, ws);
  PQ_LAUNCH_CHECK();
  return 0;
}


static int mean_many_fill(MeanMany& m, const float* const* x, float* const* dx, const int64_t* counts, const int32_t* modes,
                          const float* clamp_min, int32_t n) {
  m.n = n;
  for (int g = 0; g < n; ++g) {
    if (!x[g] || counts[g] < 1 || modes[g] < 0 || modes[g] > 2 || (dx && !dx[g])) return 1;
    m.x[g] = x[g]; m.dx[g] = dx ? dx[g] : nullptr; m.count[g] = (long)counts[g]; m.mode[g] = modes[g];
    m.cmin[g] = clamp_min ? clamp_min[g] : 0.f;
  }
  return 0;
}
extern "C" int pq3d_mean_many(const float* const* x, const int64_t* counts, const int32_t* modes, const float* clamp_min,
                              int32_t n, float* out, float* ws, void* stream) {
  PQ_DEVICE_GUARD(stream, out);
  PQ_CHECK_ARG(x && counts && modes && out && ws && n >= 1 && n <= PQ3D_MAX_GROUPS, "pq3d_mean_many: bad args");
  MeanMany m;
  PQ_CHECK_ARG(!mean_many_fill(m, x, nullptr, counts, modes, clamp_min, n), "pq3d_mean_many: null tensor / empty / bad mode");
  hipLaunchKernelGGL(mean_many_kernel, dim3(MEAN_MANY_BLOCKS, n), dim3(256), 0, (hipStream_t)stream, m, out, ws);
  hipLaunchKernelGGL(mean_many_combine_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, m, out, (const float*)ws, MEAN_MANY_BLOCKS);
  PQ_LAUNCH_CHECK();
  return 0;
}
extern "C" int pq3d_mean_many_bwd(const float* const* x, float* const* dx, const int64_t* counts, const int32_t* modes,
                                  const float* clamp_min, int32_t n, const float* gout, void* stream) {
  PQ_DEVICE_GUARD(stream, gout);
  PQ_CHECK_ARG(x && dx && counts && modes && gout && n >= 1 && n <= PQ3D_MAX_GROUPS, "pq3d_mean_many_bwd: bad args");
  MeanMany m;
  PQ_CHECK_ARG(!mean_many_fill(m, x, dx, counts, modes, clamp_min, n), "pq3d_mean_many_bwd: null tensor / empty / bad mode");
  hipLaunchKernelGGL(mean_many_bwd_kernel, dim3(128, n), dim3(256), 0, (hipStream_t)stream, m, gout);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_fill_scaled(float* dst, int64_t n, const float* scalar, float c, void* stream) {
  PQ_DEVICE_GUARD(stream, dst);
  PQ_CHECK_ARG(dst && scalar && n >= 0, "pq3d_fill_scaled: bad args");
  if (n == 0) return 0;
  hipLaunchKernelGGL(fill_scaled_kernel, dim3(grid1d(n, 256, 1024)), dim3(256), 0, (hipStream_t)stream, dst, (long)n, scalar, c);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_cast_transpose(const float* const* src, void* const* out, void* const* outT, int32_t groups, int32_t rows,
                                   int32_t cols, void* stream) {
  PQ_DEVICE_GUARD(stream, nullptr);
  PQ_CHECK_ARG(src && out && outT && groups >= 1 && groups <= PQ3D_MAX_GROUPS, "pq3d_cast_transpose: bad args");
  PQ_CHECK_ARG(rows > 0 && cols > 0 && cols % 32 == 0 && rows % cols == 0, "pq3d_cast_transpose: cols % 32 == 0, rows % cols == 0");
  CastTPtrs p;
  for (int g = 0; g < groups; ++g) {
    PQ_CHECK_ARG(src[g] && out[g] && outT[g], "pq3d_cast_transpose: null pointer");
    p.src[g] = src[g]; p.out[g] = (bf16_t*)out[g]; p.outT[g] = (bf16_t*)outT[g];
  }
  hipLaunchKernelGGL(cast_transpose_kernel, dim3(cols / 32, rows / 32, groups), dim3(256), 0, (hipStream_t)stream, p, rows, cols);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_colsum(const void* x, int32_t dt, int64_t R, int64_t N, int64_t ld, float* out, void* stream) {
  PQ_DEVICE_GUARD(stream, x);
  PQ_CHECK_ARG(x && out && R >= 0 && N >= 1 && ld >= N, "pq3d_colsum: bad args");
  const void* xs[1] = {x};
  float* outs[1] = {out};
  return pq3d_colsum_grouped(xs, outs, 1, dt, R, N, ld, 0, stream);
}

extern "C" int pq3d_colsum_grouped(const void* const* x, float* const* out, int32_t groups, int32_t dt, int64_t R,
                                   int64_t N, int64_t ld, int32_t accumulate, void* stream) {
  PQ_DEVICE_GUARD(stream, nullptr);
  PQ_CHECK_ARG(x && out && groups >= 1 && groups <= PQ3D_MAX_GROUPS && R >= 0 && N >= 1 && ld >= N,
               "pq3d_colsum_grouped: bad args");
  ColsumPtrs cp;
  for (int g = 0; g < groups; ++g) {
    PQ_CHECK_ARG(x[g] && out[g], "pq3d_colsum_grouped: null pointer");
    cp.x[g] = x[g]; cp.out[g] = out[g];
  }
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)((N + 63) / 64), groups);
  if (accumulate == 1 && R >= 2048) {   // (accumulate == 2: the deterministic single-slice form at every row count) long columns into an accumulating (pre-zeroed / running) output: row slices, ~2 workgroups per CU
    const long blocks = (long)grid.x * grid.y;
    long nz = 512 / (blocks > 0 ? blocks : 1);
    if (nz > R / 512) nz = R / 512;
    if (nz > 1) grid.z = (unsigned)nz;
  }
  if (dt == PQ3D_F32) hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(1024), 0, s, cp, (long)R, (long)N, (long)ld, accumulate);
  else hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(1024), 0, s, cp, (long)R, (long)N, (long)ld, accumulate);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_scale_rows(const void* x, int32_t dtx, void* y, int32_t dty, int64_t R, int64_t C,
                               const float* scale, const uint8_t* zero_flag, const uint8_t* keep_mask, void* stream) {
  PQ_DEVICE_GUARD(stream, x);
  PQ_CHECK_ARG(x && y && R >= 0 && C >= 1, "pq3d_scale_rows: bad args");
  if (R == 0) return 0;
  hipLaunchKernelGGL(scale_rows_kernel, dim3(grid1d(R * C)), dim3(256), 0, (hipStream_t)stream, x, dtx, y, dty, (long)R,
                     (long)C, scale, zero_flag, keep_mask);
  PQ_LAUNCH_CHECK();
  return 0;
}

namespace {
struct ScaleRowsPtrs { const void* x[PQ3D_MAX_GROUPS]; void* y[PQ3D_MAX_GROUPS]; };
// scale_rows_kernel's fast path for several same-shape tensors that share the row scales / flags (blockIdx.y = tensor): the
// mask-logit gradients of all prediction layers of a pass
__global__ void scale_rows_grouped_kernel(const ScaleRowsPtrs p, int dty, long R, long C, const float* scale, const uint8_t* zero_flag,
                                          const uint8_t* keep_mask) {
  const float* x = (const float*)p.x[blockIdx.y];
  void* y = p.y[blockIdx.y];
  const long total = R * C;
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < total; i += (long)gridDim.x * blockDim.x * 8) {
    const long r = i / C;
    const bool keep = (!zero_flag || !zero_flag[r]) && (!keep_mask || keep_mask[r]);
    const float sc = keep ? (scale ? scale[r] : 1.f) : 0.f;
    const float4 a = *(const float4*)(x + i), b = *(const float4*)(x + i + 4);
    const float v[8] = {keep ? a.x * sc : 0.f, keep ? a.y * sc : 0.f, keep ? a.z * sc : 0.f, keep ? a.w * sc : 0.f,
                        keep ? b.x * sc : 0.f, keep ? b.y * sc : 0.f, keep ? b.z * sc : 0.f, keep ? b.w * sc : 0.f};
    if (dty == PQ3D_BF16) *(u32x4*)((bf16_t*)y + i) = pack_frag<bf16_t>(v);
    else {
      *(float4*)((float*)y + i) = make_float4(v[0], v[1], v[2], v[3]);
      *(float4*)((float*)y + i + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
  }
}
}  // namespace

extern "C" int pq3d_scale_rows_grouped(const float* const* x, void* const* y, int32_t groups, int32_t dty, int64_t R, int64_t C,
                                       const float* scale, const uint8_t* zero_flag, const uint8_t* keep_mask, void* stream) {
  PQ_DEVICE_GUARD(stream, nullptr);
  PQ_CHECK_ARG(x && y && groups >= 1 && groups <= PQ3D_MAX_GROUPS && R >= 0 && C >= 8 && (C & 7) == 0 &&
               (dty == PQ3D_F32 || dty == PQ3D_BF16), "pq3d_scale_rows_grouped: 1..32 fp32 tensors, C % 8 == 0");
  if (R == 0) return 0;
  ScaleRowsPtrs p;
  for (int g = 0; g < groups; ++g) {
    PQ_CHECK_ARG(x[g] && y[g] && ((((uintptr_t)x[g]) | ((uintptr_t)y[g])) & 15) == 0, "pq3d_scale_rows_grouped: tensors must be 16-byte aligned");
    p.x[g] = x[g]; p.y[g] = y[g];
  }
  hipLaunchKernelGGL(scale_rows_grouped_kernel, dim3(grid1d(R * C / 8, 256, 1024), groups), dim3(256), 0, (hipStream_t)stream, p, dty,
                     (long)R, (long)C, scale, zero_flag, keep_mask);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_add_cast(const float* const* a, const float* const* b, void* const* out, int32_t groups,
                             int32_t dt_out, int64_t n, void* stream) {
  PQ_DEVICE_GUARD(stream, nullptr);
  PQ_CHECK_ARG(a && out && groups >= 1 && groups <= PQ3D_MAX_GROUPS && n >= 0, "pq3d_add_cast: bad args");
  PQ_CHECK_ARG((n % 8) == 0, "pq3d_add_cast: n must be a multiple of 8");
  AddCastPtrs p;
  for (int g = 0; g < groups; ++g) {
    PQ_CHECK_ARG(a[g] && out[g], "pq3d_add_cast: null pointer");
    p.a[g] = a[g]; p.b[g] = b ? b[g] : nullptr; p.out[g] = out[g];
  }
  if (n == 0) return 0;
  hipLaunchKernelGGL(add_cast_kernel, dim3(grid1d(n / 8, 256, 1024), groups), dim3(256), 0, (hipStream_t)stream, p,
                     dt_out, (long)n);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_split_planes(const float* const* a, const float* const* b, void* const* hi, void* const* lo,
                                 const int64_t* counts, int32_t groups, void* stream) {
  PQ_DEVICE_GUARD(stream, nullptr);
  PQ_CHECK_ARG(a && lo && counts && groups >= 1 && groups <= PQ3D_MAX_GROUPS, "pq3d_split_planes: bad args");
  SplitPtrs p;
  long nmax = 0;
  for (int g = 0; g < groups; ++g) {
    PQ_CHECK_ARG(a[g] && lo[g] && counts[g] >= 0 && (counts[g] % 8) == 0, "pq3d_split_planes: null pointer / count not a multiple of 8");
    PQ_CHECK_ARG(((((uintptr_t)a[g]) | ((uintptr_t)lo[g]) | ((uintptr_t)(hi ? hi[g] : nullptr)) | ((uintptr_t)(b ? b[g] : nullptr))) & 15) == 0,
                 "pq3d_split_planes: operands must be 16-byte aligned");
    p.a[g] = a[g]; p.b[g] = b ? b[g] : nullptr; p.hi[g] = hi ? hi[g] : nullptr; p.lo[g] = lo[g]; p.n[g] = (long)counts[g];
    nmax = counts[g] > nmax ? (long)counts[g] : nmax;
  }
  if (nmax == 0) return 0;
  hipLaunchKernelGGL(split_planes_kernel, dim3(grid1d(nmax / 8, 256, 1024), groups), dim3(256), 0, (hipStream_t)stream, p);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_bias_add_rows(const float* x, const float* bias, float* out, int64_t R, int64_t N, void* stream) {
  PQ_DEVICE_GUARD(stream, x);
  PQ_CHECK_ARG(x && bias && out && R >= 0 && N >= 4 && (N % 4) == 0, "pq3d_bias_add_rows: bad args (N % 4 == 0)");
  if (R == 0) return 0;
  hipLaunchKernelGGL(bias_add_rows_kernel, dim3(grid1d(R * N / 4)), dim3(256), 0, (hipStream_t)stream, x, bias, out,
                     (long)R, (long)N);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_act_bwd(const void* dy, int32_t dt_dy, const void* saved, int32_t dt_saved, void* dpre,
                            int32_t dt_dpre, int32_t act, int64_t n, void* stream) {
  PQ_DEVICE_GUARD(stream, dy);
  PQ_CHECK_ARG(dy && saved && dpre && n >= 0, "pq3d_act_bwd: bad args");
  if (n == 0) return 0;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, dy, dt_dy, saved, dt_saved,
                     dpre, dt_dpre, act, (long)n);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_fill_cols(const float* x, float* y, int64_t R, int64_t C, const int32_t* cols, int32_t ncols,
                              float value, void* stream) {
  PQ_DEVICE_GUARD(stream, x);
  PQ_CHECK_ARG(x && y && R >= 0 && C >= 1 && ncols >= 0 && (ncols == 0 || cols), "pq3d_fill_cols: bad args");
  if (R == 0) return 0;
  hipLaunchKernelGGL(fill_cols_kernel, dim3(grid1d(R * C)), dim3(256), 0, (hipStream_t)stream, x, y, (long)R, (long)C,
                     cols, ncols, value);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_mask_inv_den(const uint8_t* const* masks, int32_t M, int64_t n, float* inv_den, void* stream) {
  PQ_DEVICE_GUARD(stream, nullptr);
  PQ_CHECK_ARG(masks && M >= 1 && M <= PQ3D_MAX_GROUPS && inv_den && n >= 0, "pq3d_mask_inv_den: bad args");
  if (n == 0) return 0;
  MaskPtrs mp;
  for (int m = 0; m < PQ3D_MAX_GROUPS; ++m) mp.p[m] = m < M ? masks[m] : nullptr;
  hipLaunchKernelGGL(mask_inv_den_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mp, M,
                     (long)n, inv_den);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_pairwise_locs(const float* centers, int64_t center_stride, float* out, int32_t B, int32_t L,
                                  float eps, void* stream) {
  PQ_DEVICE_GUARD(stream, centers);
  PQ_CHECK_ARG(centers && out && B >= 0 && L >= 0 && center_stride >= 3, "pq3d_pairwise_locs: bad args");
  PQ_CHECK_ARG((size_t)L * 3 * sizeof(float) <= 48 * 1024, "pq3d_pairwise_locs: L too large");
  if (B == 0 || L == 0) return 0;
  hipLaunchKernelGGL(pairwise_locs_kernel, dim3((L + 7) / 8, B), dim3(256), sizeof(float) * 3 * L, (hipStream_t)stream,
                     centers, (long)center_stride, out, L, eps);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_fourier(const float* xyz, int64_t xyz_stride, const float* cmin, const float* cmax,
                            const float* gauss_B, float* out, int32_t B, int32_t N, int32_t half, void* stream) {
  PQ_DEVICE_GUARD(stream, xyz);
  PQ_CHECK_ARG(xyz && cmin && cmax && gauss_B && out && xyz_stride >= 3 && half >= 1, "pq3d_fourier: bad args");
  if (B == 0 || N == 0) return 0;
  hipLaunchKernelGGL(fourier_kernel, dim3(grid1d((long)B * N * half)), dim3(256), 0, (hipStream_t)stream, xyz,
                     (long)xyz_stride, cmin, cmax, gauss_B, out, B, N, half);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_fourier_pair(const float* xyz_a, int64_t stride_a, int32_t Na, const float* xyz_b, int64_t stride_b,
                                 int32_t Nb, const float* cmin, const float* cmax, const float* gauss_B, float* out,
                                 int32_t B, int32_t half, void* stream) {
  PQ_DEVICE_GUARD(stream, xyz_a);
  PQ_CHECK_ARG(xyz_a && xyz_b && cmin && cmax && gauss_B && out && stride_a >= 3 && stride_b >= 3 && half >= 1 && Na >= 0 &&
                   Nb >= 0, "pq3d_fourier_pair: bad args");
  if (B == 0 || Na + Nb == 0) return 0;
  PQ_CHECK_ARG((long)B * (Na + Nb) < (1L << 30), "pq3d_fourier_pair: too many rows");
  const int rpb_ = half < 256 ? 256 / half : 1;
  hipLaunchKernelGGL(fourier_pair_kernel, dim3(grid1d(((long)B * (Na + Nb) + rpb_ - 1) / rpb_, 1, 65535)), dim3(256), 0, (hipStream_t)stream, xyz_a,
                     (long)stride_a, Na, xyz_b, (long)stride_b, Nb, cmin, cmax, gauss_B, out, B, half);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_spatial_bias_fwd_grouped(const float* pl, const float* const* W, const float* const* bw,
                                             float* const* bias, int32_t groups, int32_t B, int32_t H, int32_t L,
                                             void* stream) {
  PQ_DEVICE_GUARD(stream, pl);
  PQ_CHECK_ARG(pl && W && bw && bias && H >= 1 && groups >= 1 && groups <= PQ3D_MAX_GROUPS,
               "pq3d_spatial_bias_fwd_grouped: bad args");
  SbFwdGroups gr;
  for (int g = 0; g < groups; ++g) {
    PQ_CHECK_ARG(W[g] && bw[g] && bias[g], "pq3d_spatial_bias_fwd_grouped: null pointer");
    gr.W[g] = W[g]; gr.bw[g] = bw[g]; gr.bias[g] = bias[g];
  }
  if (B == 0 || L == 0) return 0;
  hipLaunchKernelGGL(spatial_bias_fwd_kernel, dim3(grid1d((long)B * L * L), groups), dim3(256), 0, (hipStream_t)stream, pl,
                     gr, B, H, L);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_spatial_bias_fwd(const float* pl, const float* W, const float* bw, float* bias, int32_t B,
                                     int32_t H, int32_t L, void* stream) {
  PQ_DEVICE_GUARD(stream, pl);
  return pq3d_spatial_bias_fwd_grouped(pl, &W, &bw, &bias, 1, B, H, L, stream);
}

extern "C" int pq3d_spatial_bias_bwd(const float* pl, const float* W, const float* bw, const float* dbias, float* dW,
                                     float* dbw, int32_t B, int32_t H, int32_t L, void* stream) {
  PQ_DEVICE_GUARD(stream, pl);
  PQ_CHECK_ARG(pl && W && bw && dbias && dW && dbw && H >= 1, "pq3d_spatial_bias_bwd: bad args");
  hipStream_t s = (hipStream_t)stream;
  if (int e = memset_async(dW, sizeof(float) * H * 5, s)) return e;
  if (int e = memset_async(dbw, sizeof(float) * H, s)) return e;
  return pq3d_spatial_bias_bwd_acc(pl, W, bw, dbias, dW, dbw, B, H, L, stream);
}

extern "C" int pq3d_spatial_bias_bwd_grouped(const float* pl, const float* const* W, const float* const* bw,
                                             const float* const* dbias, float* const* dW, float* const* dbw,
                                             int32_t groups, int32_t B, int32_t H, int32_t L, void* stream) {
  PQ_DEVICE_GUARD(stream, pl);
  PQ_CHECK_ARG(pl && W && bw && dbias && dW && dbw && H >= 1 && groups >= 1 && groups <= PQ3D_MAX_GROUPS,
               "pq3d_spatial_bias_bwd_grouped: bad args");
  SbGroups gr;
  for (int g = 0; g < groups; ++g) {
    PQ_CHECK_ARG(W[g] && bw[g] && dbias[g] && dW[g] && dbw[g], "pq3d_spatial_bias_bwd_grouped: null pointer");
    gr.W[g] = W[g]; gr.bw[g] = bw[g]; gr.dbias[g] = dbias[g]; gr.dW[g] = dW[g]; gr.dbw[g] = dbw[g];
  }
  hipStream_t s = (hipStream_t)stream;
  if (B == 0 || L == 0) return 0;
  PQ_CHECK_ARG(H <= 16, "pq3d_spatial_bias_bwd: at most 16 heads");
  if (H <= 8)
    hipLaunchKernelGGL(spatial_bias_bwd_kernel<8>, dim3(grid1d(((long)B * L * L + 3) / 4, 256, SB_BLOCKS), groups), dim3(256), 0, s, pl,
                       gr, B, H, L);
  else
    hipLaunchKernelGGL(spatial_bias_bwd_kernel<16>, dim3(grid1d(((long)B * L * L + 3) / 4, 256, SB_BLOCKS), groups), dim3(256), 0, s, pl,
                       gr, B, H, L);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_spatial_bias_bwd_acc(const float* pl, const float* W, const float* bw, const float* dbias,
                                         float* dW, float* dbw, int32_t B, int32_t H, int32_t L, void* stream) {
  PQ_DEVICE_GUARD(stream, pl);
  return pq3d_spatial_bias_bwd_grouped(pl, &W, &bw, &dbias, &dW, &dbw, 1, B, H, L, stream);
}

extern "C" int pq3d_gate_mix_fwd(const float* q, const float* u, const float* g, float* y, int64_t n, void* stream) {
  PQ_DEVICE_GUARD(stream, q);
  PQ_CHECK_ARG(q && u && g && y && n >= 0, "pq3d_gate_mix_fwd: bad args");
  if (n == 0) return 0;
  hipLaunchKernelGGL(gate_fwd_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, q, u, g, y, (long)n);
  PQ_LAUNCH_CHECK();
  return 0;
}
extern "C" int pq3d_gate_mix_bwd(const float* q, const float* u, const float* g, const float* dy, float* dq, float* du,
                                 float* dg, int64_t n, void* stream) {
  PQ_DEVICE_GUARD(stream, q);
  PQ_CHECK_ARG(q && u && g && dy && dq && du && dg && n >= 0, "pq3d_gate_mix_bwd: bad args");
  if (n == 0) return 0;
  hipLaunchKernelGGL(gate_bwd_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, q, u, g, dy, dq, du, dg,
                     (long)n);
  PQ_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------ dropout
namespace {
__global__ void dropout_mask_kernel(uint8_t* keep, long rows, long cols, const pq3d_dropout dr) {
  const DropState s = drop_init(dr, 0, cols);
  const long n = rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    keep[i] = drop_keep(s, (uint32_t)(i / cols), (uint32_t)(i % cols)) ? 1 : 0;
}
__global__ void dropout_apply_kernel(const void* x, int dtx, void* y, int dty, long rows, long cols, const pq3d_dropout dr,
                                     const float alpha) {
  DropState s = drop_init(dr, 0, cols);
  s.scale *= alpha;
  const long half = (cols + 1) >> 1, n = rows * half;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / half, j = i % half;
    const uint32_t w = drop_word(s, (uint32_t)r, (uint32_t)j);
    const long e = r * cols + 2 * j;
    store_elem(y, dty, e, drop_keep_lo(s, w) ? load_elem(x, dtx, e) * s.scale : 0.f);
    if (2 * j + 1 < cols) store_elem(y, dty, e + 1, drop_keep_hi(s, w) ? load_elem(x, dtx, e + 1) * s.scale : 0.f);
  }
}
}  // namespace

extern "C" int pq3d_dropout_mask(uint8_t* keep, int64_t rows, int64_t cols, const pq3d_dropout* dr, void* stream) {
  PQ_DEVICE_GUARD(stream, keep);
  PQ_CHECK_ARG(keep && dr && dr->seed && rows >= 0 && cols >= 1 && dr->p >= 0.f, "pq3d_dropout_mask: bad args");
  PQ_CHECK_DROP(*dr, rows, cols, "pq3d_dropout_mask");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid1d(rows * cols, 256, 8192)), dim3(256), 0, (hipStream_t)stream, keep,
                     (long)rows, (long)cols, *dr);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_dropout_apply(const void* x, int32_t dt_x, void* y, int32_t dt_y, int64_t rows, int64_t cols,
                                  const pq3d_dropout* dr, void* stream) {
  return pq3d_dropout_apply_scaled(x, dt_x, y, dt_y, rows, cols, dr, 1.f, stream);
}
extern "C" int pq3d_dropout_apply_scaled(const void* x, int32_t dt_x, void* y, int32_t dt_y, int64_t rows, int64_t cols,
                                         const pq3d_dropout* dr, float alpha, void* stream) {
  PQ_DEVICE_GUARD(stream, x);
  PQ_CHECK_ARG(x && y && dr && dr->seed && rows >= 0 && cols >= 1 && dr->p > 0.f, "pq3d_dropout_apply: bad args");
  PQ_CHECK_DROP(*dr, rows, cols, "pq3d_dropout_apply");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(dropout_apply_kernel, dim3(grid1d(rows * ((cols + 1) / 2), 256, 8192)), dim3(256), 0,
                     (hipStream_t)stream, x, dt_x, y, dt_y, (long)rows, (long)cols, *dr, alpha);
  PQ_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------ ragged -> padded
// pad_sequence / pad_sequence_2d of the reference's collate (data/data_utils.py:337-382) for an on-device ragged batch:
// samples packed back to back, per-sample extents in small device arrays.  Pure copies, typed by element size.
namespace {
template <typename E>
__global__ __launch_bounds__(256) void pad_sequence_kernel(const E* __restrict__ src, const int64_t* __restrict__ offsets,
                                                           E* __restrict__ out, uint8_t* __restrict__ mask, long L, long D,
                                                           E pad) {
  const int b = blockIdx.y;
  const long n = offsets[b + 1] - offsets[b], base = offsets[b] * D;
  const long total = L * D;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long row = i / D;
    out[(long)b * total + i] = row < n ? src[base + i] : pad;
    if (mask && i % D == 0) mask[(long)b * L + row] = row >= n ? 1 : 0;   // True as masked (data_utils.py:352)
  }
}
template <typename E>
__global__ __launch_bounds__(256) void pad_sequence_2d_kernel(const E* __restrict__ src, const int64_t* __restrict__ offsets,
                                                              const int32_t* __restrict__ hs, const int32_t* __restrict__ ws,
                                                              E* __restrict__ out, uint8_t* __restrict__ mask, long H, long W,
                                                              long D, E pad) {
  const int b = blockIdx.y;
  const long h = hs[b], w = ws[b], base = offsets[b];
  const long total = H * W * D;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long c = i % D, col = (i / D) % W, row = i / (D * W);
    const bool in = row < h && col < w;
    out[(long)b * total + i] = in ? src[base + (row * w + col) * D + c] : pad;
    if (mask && c == 0) mask[((long)b * H + row) * W + col] = in ? 0 : 1;
  }
}
template <typename E> E pad_as(const void* p) { E v; memcpy(&v, p, sizeof(E)); return v; }
}  // namespace

extern "C" int pq3d_pad_sequence(const void* src, const int64_t* offsets, void* out, uint8_t* mask, int32_t B, int64_t L,
                                 int64_t D, int32_t elem_size, const void* pad_value, void* stream) {
  PQ_DEVICE_GUARD(stream, src);
  PQ_CHECK_ARG(offsets && out && pad_value && B >= 0 && L >= 0 && D >= 1, "pq3d_pad_sequence: bad args");
  PQ_CHECK_ARG(elem_size == 1 || elem_size == 2 || elem_size == 4 || elem_size == 8, "pq3d_pad_sequence: element size");
  if (B == 0 || L == 0) return 0;
  PQ_CHECK_ARG(src != nullptr, "pq3d_pad_sequence: null src");
  dim3 grid(grid1d(L * D, 256, 2048), B);
  hipStream_t s = (hipStream_t)stream;
#define PQ_PAD1(E) hipLaunchKernelGGL(pad_sequence_kernel<E>, grid, dim3(256), 0, s, (const E*)src, offsets, (E*)out, mask, \
                                      (long)L, (long)D, pad_as<E>(pad_value))
  if (elem_size == 1) PQ_PAD1(uint8_t); else if (elem_size == 2) PQ_PAD1(uint16_t);
  else if (elem_size == 4) PQ_PAD1(uint32_t); else PQ_PAD1(uint64_t);
#undef PQ_PAD1
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_pad_sequence_2d(const void* src, const int64_t* offsets, const int32_t* heights, const int32_t* widths,
                                    void* out, uint8_t* mask, int32_t B, int64_t H, int64_t W, int64_t D, int32_t elem_size,
                                    const void* pad_value, void* stream) {
  PQ_DEVICE_GUARD(stream, src);
  PQ_CHECK_ARG(offsets && heights && widths && out && pad_value && B >= 0 && H >= 0 && W >= 0 && D >= 1,
               "pq3d_pad_sequence_2d: bad args");
  PQ_CHECK_ARG(elem_size == 1 || elem_size == 2 || elem_size == 4 || elem_size == 8, "pq3d_pad_sequence_2d: element size");
  if (B == 0 || H == 0 || W == 0) return 0;
  PQ_CHECK_ARG(src != nullptr, "pq3d_pad_sequence_2d: null src");
  dim3 grid(grid1d(H * W * D, 256, 2048), B);
  hipStream_t s = (hipStream_t)stream;
#define PQ_PAD2(E) hipLaunchKernelGGL(pad_sequence_2d_kernel<E>, grid, dim3(256), 0, s, (const E*)src, offsets, heights, widths, \
                                      (E*)out, mask, (long)H, (long)W, (long)D, pad_as<E>(pad_value))
  if (elem_size == 1) PQ_PAD2(uint8_t); else if (elem_size == 2) PQ_PAD2(uint16_t);
  else if (elem_size == 4) PQ_PAD2(uint32_t); else PQ_PAD2(uint64_t);
#undef PQ_PAD2
  PQ_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------ embedding rows
// out[r,:] = table[ids[r],:] (nn.Embedding of the T5 decoder input tokens) and its scatter-add gradient.
namespace {
__global__ void embedding_fwd_kernel(const float* __restrict__ table, const int64_t* __restrict__ ids, float* __restrict__ out,
                                     long R, int d) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < R * d; i += (long)gridDim.x * blockDim.x)
    out[i] = table[ids[i / d] * d + i % d];
}
__global__ void embedding_bwd_kernel(const float* __restrict__ dout, const int64_t* __restrict__ ids, float* __restrict__ dtable,
                                     long R, int d) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < R * d; i += (long)gridDim.x * blockDim.x)
    unsafeAtomicAdd(&dtable[ids[i / d] * d + i % d], dout[i]);
}
}  // namespace
extern "C" int pq3d_embedding_fwd(const float* table, const int64_t* ids, float* out, int64_t R, int32_t d, void* stream) {
  PQ_DEVICE_GUARD(stream, table);
  PQ_CHECK_ARG(table && ids && out && R >= 0 && d >= 1, "pq3d_embedding_fwd: bad args");
  if (R == 0) return 0;
  hipLaunchKernelGGL(embedding_fwd_kernel, dim3(grid1d(R * d)), dim3(256), 0, (hipStream_t)stream, table, ids, out, (long)R, d);
  PQ_LAUNCH_CHECK();
  return 0;
}
extern "C" int pq3d_embedding_bwd_acc(const float* dout, const int64_t* ids, float* dtable, int64_t R, int32_t d,
                                      void* stream) {
  PQ_DEVICE_GUARD(stream, dout);
  PQ_CHECK_ARG(dout && ids && dtable && R >= 0 && d >= 1, "pq3d_embedding_bwd_acc: bad args");
  if (R == 0) return 0;
  hipLaunchKernelGGL(embedding_bwd_kernel, dim3(grid1d(R * d)), dim3(256), 0, (hipStream_t)stream, dout, ids, dtable, (long)R,
                     d);
  PQ_LAUNCH_CHECK();
  return 0;
}
