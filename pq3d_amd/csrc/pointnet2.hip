// PointNet++ point-set operators: the only native code of the reference
// (modules/third_party/pointnet2/_ext_src/src/*.cu), re-designed for gfx950 rather than translated:
//   * furthest point sampling keeps every point's coordinates AND running min-distance in registers for the whole
//     m-step loop (the reference re-reads the cloud and a global temp array every step) and finds the arg-max with wave
//     DPP reductions + one 16-entry LDS exchange (the reference walks a 10-level shared-memory tree);
//   * ball query is one wave per centre scanning 64 points per step with a ballot + prefix-popcount to keep the
//     reference's "first nsample hits in index order" semantics (the reference scans sequentially in one thread);
//   * three_nn stages the known points through LDS tiles shared by the block; gathers / interpolation are coalesced
//     along the point axis; backward passes scatter with fp32 atomics like the reference.
// All fp32 / int32, batch-first tensors exactly as the reference's Python wrappers pass them
// (modules/third_party/pointnet2/pointnet2_utils.py).
#include "common.h"

namespace {

constexpr int FPS_THREADS = 1024;
constexpr int FPS_MAXPT = 8;   // points per thread held in registers -> n <= 8192

// arg-max of (value, index) with ties to the SMALLER index (the reference's tie order depends on its block size; ties
// only arise for duplicated points)
PQ_DEV void argmax_pair(float& v, int& i, float ov, int oi) {
  if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}

// NT threads x PT points per thread.  Small clouds use few waves (cheap barriers, 4-entry cross-wave exchange): the m-step
// loop is a pure latency chain, so fewer, fatter threads win (1024 points: 4 waves x 4 points = 0.9 us/step vs 2.3 us/step
// with 16 waves x 1 point).
template <int NT, int PT>
__global__ __launch_bounds__(NT) void fps_kernel(const float* __restrict__ xyz, int32_t* __restrict__ idxs, int n, int m) {
  constexpr int FPS_THREADS = NT, NWV = NT / 64;
  __shared__ float red_v[2][NWV];
  __shared__ int red_i[2][NWV];
  __shared__ float cur[2][3];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* p = xyz + (long)b * n * 3;
  int32_t* out = idxs + (long)b * m;
  float px[PT], py[PT], pz[PT], dist[PT];
  bool live[PT];
#pragma unroll
  for (int j = 0; j < PT; ++j) {
    const int k = tid + j * FPS_THREADS;
    const bool in = k < n;
    px[j] = in ? p[k * 3 + 0] : 0.f;
    py[j] = in ? p[k * 3 + 1] : 0.f;
    pz[j] = in ? p[k * 3 + 2] : 0.f;
    dist[j] = 1e10f;                                                        // sampling.cpp: temp = full(1e10)
    live[j] = in && (px[j] * px[j] + py[j] * py[j] + pz[j] * pz[j]) > 1e-3f;   // the reference skips |p|^2 <= 1e-3
  }
  if (tid == 0) { out[0] = 0; cur[0][0] = p[0]; cur[0][1] = p[1]; cur[0][2] = p[2]; }
  __syncthreads();
  for (int s = 1; s < m; ++s) {
    const int pb = (s - 1) & 1;
    const float x1 = cur[pb][0], y1 = cur[pb][1], z1 = cur[pb][2];
    float best = -1.f;
    int besti = 0;
#pragma unroll
    for (int j = 0; j < PT; ++j) {
      if (live[j]) {
        const float dx = px[j] - x1, dy = py[j] - y1, dz = pz[j] - z1;
        const float d = dx * dx + dy * dy + dz * dz;
        dist[j] = fminf(d, dist[j]);
        if (dist[j] > best) { best = dist[j]; besti = tid + j * FPS_THREADS; }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) argmax_pair(best, besti, __shfl_xor(best, o, 64), __shfl_xor(besti, o, 64));
    if (lane == 0) { red_v[pb][wave] = best; red_i[pb][wave] = besti; }
    __syncthreads();
    float v = red_v[pb][lane & (NWV - 1)];
    int i = red_i[pb][lane & (NWV - 1)];
#pragma unroll
    for (int o = NWV / 2; o > 0; o >>= 1) argmax_pair(v, i, __shfl_xor(v, o, 64), __shfl_xor(i, o, 64));
    // every thread now knows the winner; its owner publishes the coordinates for the next step
#pragma unroll
    for (int j = 0; j < PT; ++j)
      if (tid + j * FPS_THREADS == i) { cur[pb ^ 1][0] = px[j]; cur[pb ^ 1][1] = py[j]; cur[pb ^ 1][2] = pz[j]; }
    if (tid == 0) out[s] = i;
    __syncthreads();
  }
}

// one wave per centre; 64 candidate points per step, hits appended in index order
__global__ __launch_bounds__(256) void ball_query_kernel(const float* __restrict__ new_xyz, const float* __restrict__ xyz,
                                                         int32_t* __restrict__ idx, int n, int m, float radius2,
                                                         int nsample) {
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= m) return;
  const float* c = new_xyz + ((long)b * m + j) * 3;
  const float cx = c[0], cy = c[1], cz = c[2];
  const float* p = xyz + (long)b * n * 3;
  int32_t* o = idx + ((long)b * m + j) * nsample;
  int cnt = 0;
  for (int k0 = 0; k0 < n && cnt < nsample; k0 += 64) {
    const int k = k0 + lane;
    bool hit = false;
    if (k < n) {
      const float dx = cx - p[k * 3 + 0], dy = cy - p[k * 3 + 1], dz = cz - p[k * 3 + 2];
      hit = dx * dx + dy * dy + dz * dz < radius2;
    }
    const unsigned long long bal = __ballot(hit);
    if (bal == 0) continue;
    if (cnt == 0) {   // first hit of this centre: the reference pre-fills every slot with it
      const int first = k0 + __ffsll((long long)bal) - 1;
      for (int l = lane; l < nsample; l += 64) o[l] = first;
    }
    const int pos = cnt + __popcll(bal & ((1ull << lane) - 1ull));
    if (hit && pos < nsample) o[pos] = k;
    cnt += __popcll(bal);
  }
  if (cnt == 0)   // no point inside the ball: the reference leaves the (zero-initialised) output untouched
    for (int l = lane; l < nsample; l += 64) o[l] = 0;
}

// out[b,c,i] = points[b,c,idx[b,i]]  (gather_points: i over m sampled points; group_points: i over npoint*nsample)
// each thread reads its index ONCE and walks the channels (blockIdx.y = group of CPB channels): writes are coalesced along
// the point axis, the index tensor is read C / CPB times instead of C times
constexpr int CPB = 16;
__global__ void gather_kernel(const float* __restrict__ points, const int32_t* __restrict__ idx, float* __restrict__ out,
                              int C, int n, long L) {
  const int b = blockIdx.z, c0 = blockIdx.y * CPB, c1 = min(c0 + CPB, C);
  const int32_t* ix = idx + (long)b * L;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < L; i += (long)gridDim.x * blockDim.x) {
    const int k = ix[i];
    for (int c = c0; c < c1; ++c) out[((long)b * C + c) * L + i] = points[((long)b * C + c) * n + k];
  }
}
__global__ void gather_grad_kernel(const float* __restrict__ gout, const int32_t* __restrict__ idx, float* __restrict__ gpts,
                                   int C, int n, long L) {
  const int b = blockIdx.z, c = blockIdx.y;
  const float* src = gout + ((long)b * C + c) * L;
  float* dst = gpts + ((long)b * C + c) * n;
  const int32_t* ix = idx + (long)b * L;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < L; i += (long)gridDim.x * blockDim.x)
    unsafeAtomicAdd(&dst[ix[i]], src[i]);
}

// three nearest known points of every unknown point (squared distances, strict '<' ordering as the reference)
__global__ __launch_bounds__(256) void three_nn_kernel(const float* __restrict__ unknown, const float* __restrict__ known,
                                                       float* __restrict__ dist2, int32_t* __restrict__ idx, int n, int m) {
  __shared__ float kx[256], ky[256], kz[256];
  const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  const float* u = unknown + ((long)b * n + min(j, n - 1)) * 3;
  const float ux = u[0], uy = u[1], uz = u[2];
  const float* kp = known + (long)b * m * 3;
  float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;   // the reference starts from 1e40 (double): same effect
  int i1 = 0, i2 = 0, i3 = 0;
  for (int k0 = 0; k0 < m; k0 += 256) {
    const int kk = k0 + threadIdx.x;
    if (kk < m) { kx[threadIdx.x] = kp[kk * 3 + 0]; ky[threadIdx.x] = kp[kk * 3 + 1]; kz[threadIdx.x] = kp[kk * 3 + 2]; }
    __syncthreads();
    const int lim = min(256, m - k0);
    for (int t = 0; t < lim; ++t) {
      const float dx = ux - kx[t], dy = uy - ky[t], dz = uz - kz[t];
      const float d = dx * dx + dy * dy + dz * dz;
      const int k = k0 + t;
      if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = k; }
      else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = k; }
      else if (d < b3) { b3 = d; i3 = k; }
    }
    __syncthreads();
  }
  if (j < n) {
    float* dd = dist2 + ((long)b * n + j) * 3;
    int32_t* ii = idx + ((long)b * n + j) * 3;
    // fewer than 3 known points: the reference's 1e40 sentinel becomes +inf when stored as fp32
    dd[0] = b1; dd[1] = b2; dd[2] = b3;
    ii[0] = i1; ii[1] = i2; ii[2] = i3;
  }
}

__global__ void three_interpolate_kernel(const float* __restrict__ points, const int32_t* __restrict__ idx,
                                         const float* __restrict__ weight, float* __restrict__ out, int C, int m, int n) {
  const int b = blockIdx.z, c = blockIdx.y;
  const float* src = points + ((long)b * C + c) * m;
  float* dst = out + ((long)b * C + c) * n;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int32_t* ix = idx + ((long)b * n + i) * 3;
    const float* w = weight + ((long)b * n + i) * 3;
    dst[i] = src[ix[0]] * w[0] + src[ix[1]] * w[1] + src[ix[2]] * w[2];
  }
}
__global__ void three_interpolate_grad_kernel(const float* __restrict__ gout, const int32_t* __restrict__ idx,
                                              const float* __restrict__ weight, float* __restrict__ gpts, int C, int m,
                                              int n) {
  const int b = blockIdx.z, c = blockIdx.y;
  const float* src = gout + ((long)b * C + c) * n;
  float* dst = gpts + ((long)b * C + c) * m;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int32_t* ix = idx + ((long)b * n + i) * 3;
    const float* w = weight + ((long)b * n + i) * 3;
    const float g = src[i];
    unsafeAtomicAdd(&dst[ix[0]], g * w[0]);
    unsafeAtomicAdd(&dst[ix[1]], g * w[1]);
    unsafeAtomicAdd(&dst[ix[2]], g * w[2]);
  }
}

unsigned blocks_for(long n, int threads, long cap = 1024) {
  long g = (n + threads - 1) / threads;
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int pq3d_furthest_point_sampling(const float* xyz, int32_t* idx, int32_t B, int32_t N, int32_t M, void* stream) {
  PQ_DEVICE_GUARD(stream, xyz);
  PQ_CHECK_ARG(xyz && idx && B >= 0 && N >= 1 && M >= 0, "pq3d_furthest_point_sampling: bad args");
  PQ_CHECK_ARG(N <= FPS_THREADS * FPS_MAXPT, "pq3d_furthest_point_sampling: at most 8192 points per cloud");
  if (B == 0 || M == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (N <= 256) hipLaunchKernelGGL((fps_kernel<64, 4>), dim3(B), dim3(64), 0, s, xyz, idx, N, M);
  else if (N <= 1024) hipLaunchKernelGGL((fps_kernel<256, 4>), dim3(B), dim3(256), 0, s, xyz, idx, N, M);
  else if (N <= 2048) hipLaunchKernelGGL((fps_kernel<256, 8>), dim3(B), dim3(256), 0, s, xyz, idx, N, M);
  else if (N <= 4096) hipLaunchKernelGGL((fps_kernel<512, 8>), dim3(B), dim3(512), 0, s, xyz, idx, N, M);
  else hipLaunchKernelGGL((fps_kernel<1024, 8>), dim3(B), dim3(1024), 0, s, xyz, idx, N, M);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_ball_query(const float* new_xyz, const float* xyz, int32_t* idx, int32_t B, int32_t N, int32_t M,
                               float radius, int32_t nsample, void* stream) {
  PQ_DEVICE_GUARD(stream, new_xyz);
  PQ_CHECK_ARG(new_xyz && xyz && idx && B >= 0 && N >= 1 && M >= 0 && nsample >= 1, "pq3d_ball_query: bad args");
  if (B == 0 || M == 0) return 0;
  hipLaunchKernelGGL(ball_query_kernel, dim3((M + 3) / 4, B), dim3(256), 0, (hipStream_t)stream, new_xyz, xyz, idx, N, M,
                     radius * radius, nsample);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_gather_points(const float* points, const int32_t* idx, float* out, int32_t B, int32_t C, int32_t N,
                                  int64_t L, void* stream) {
  PQ_DEVICE_GUARD(stream, points);
  PQ_CHECK_ARG(points && idx && out && B >= 0 && C >= 1 && N >= 1 && L >= 0, "pq3d_gather_points: bad args");
  if (B == 0 || L == 0) return 0;
  hipLaunchKernelGGL(gather_kernel, dim3(blocks_for(L, 256, 256), (C + CPB - 1) / CPB, B), dim3(256), 0, (hipStream_t)stream,
                     points, idx, out, C, N, (long)L);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_gather_points_grad(const float* grad_out, const int32_t* idx, float* grad_points, int32_t B, int32_t C,
                                       int32_t N, int64_t L, void* stream) {
  PQ_DEVICE_GUARD(stream, grad_out);
  PQ_CHECK_ARG(grad_out && idx && grad_points && B >= 0 && C >= 1 && N >= 1 && L >= 0, "pq3d_gather_points_grad: bad args");
  if (B == 0) return 0;
  ZeroList z;
  z.add(grad_points, (long)B * C * N);
  if (int e = pq3d_zero_launch(z, (hipStream_t)stream)) return e;
  if (L == 0) return 0;
  hipLaunchKernelGGL(gather_grad_kernel, dim3(blocks_for(L, 256, 64), C, B), dim3(256), 0, (hipStream_t)stream, grad_out, idx,
                     grad_points, C, N, (long)L);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_three_nn(const float* unknown, const float* known, float* dist2, int32_t* idx, int32_t B, int32_t N,
                             int32_t M, void* stream) {
  PQ_DEVICE_GUARD(stream, unknown);
  PQ_CHECK_ARG(unknown && known && dist2 && idx && B >= 0 && N >= 0 && M >= 1, "pq3d_three_nn: bad args");
  if (B == 0 || N == 0) return 0;
  hipLaunchKernelGGL(three_nn_kernel, dim3((N + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, unknown, known, dist2, idx,
                     N, M);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_three_interpolate(const float* points, const int32_t* idx, const float* weight, float* out, int32_t B,
                                      int32_t C, int32_t M, int32_t N, void* stream) {
  PQ_DEVICE_GUARD(stream, points);
  PQ_CHECK_ARG(points && idx && weight && out && B >= 0 && C >= 1 && M >= 1 && N >= 0, "pq3d_three_interpolate: bad args");
  if (B == 0 || N == 0) return 0;
  hipLaunchKernelGGL(three_interpolate_kernel, dim3(blocks_for(N, 256, 64), C, B), dim3(256), 0, (hipStream_t)stream, points,
                     idx, weight, out, C, M, N);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_three_interpolate_grad(const float* grad_out, const int32_t* idx, const float* weight, float* grad_points,
                                           int32_t B, int32_t C, int32_t M, int32_t N, void* stream) {
  PQ_DEVICE_GUARD(stream, grad_out);
  PQ_CHECK_ARG(grad_out && idx && weight && grad_points && B >= 0 && C >= 1 && M >= 1 && N >= 0,
               "pq3d_three_interpolate_grad: bad args");
  if (B == 0) return 0;
  ZeroList z;
  z.add(grad_points, (long)B * C * M);
  if (int e = pq3d_zero_launch(z, (hipStream_t)stream)) return e;
  if (N == 0) return 0;
  hipLaunchKernelGGL(three_interpolate_grad_kernel, dim3(blocks_for(N, 256, 64), C, B), dim3(256), 0, (hipStream_t)stream,
                     grad_out, idx, weight, grad_points, C, M, N);
  PQ_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------ set-abstraction rows
// The frozen PointNet++ backbone (modules/layers/pointnet.py:22-63) runs its SharedMLPs as row GEMMs: one row per
// (cloud, centre, sample), channels last.  group_rows builds those rows straight from the ball-query indices (centred
// xyz first, then the point features, zero-padded to Kp so the GEMM's K is a multiple of 8); group_maxpool is the
// max over the samples of a group, producing the next stage's [cloud, point, channel] feature rows.
__global__ void __launch_bounds__(256) group_rows_kernel(const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                                          const void* __restrict__ feats, int dt_f, long feat_stride,
                                                          const int32_t* __restrict__ idx, void* __restrict__ out, int dt_o,
                                                          int N, int C, int np, int ns, int Kp, long total) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long row = e / Kp;
    const int c = (int)(e - row * Kp);
    const long g = row / ns;            // (cloud, centre)
    const int b = (int)(g / np);
    const int j = idx ? idx[row] : (int)(row - g * ns);   // GroupAll: sample s is point s
    float v = 0.f;
    if (c < 3) {
      v = xyz[((long)b * N + j) * 3 + c];
      if (new_xyz) v -= new_xyz[g * 3 + c];
    } else if (c < 3 + C) {
      v = load_elem(feats, dt_f, ((long)b * N + j) * feat_stride + (c - 3));
    }
    store_elem(out, dt_o, e, v);
  }
}

__global__ void __launch_bounds__(256) group_maxpool_kernel(const void* __restrict__ rows, void* __restrict__ out, int dt,
                                                             long G, int ns, int C) {
  const long total = G * C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long g = e / C;
    const int c = (int)(e - g * C);
    float m = -INFINITY;
    for (int s = 0; s < ns; ++s) m = fmaxf(m, load_elem(rows, dt, (g * ns + s) * C + c));
    store_elem(out, dt, e, m);
  }
}

extern "C" int pq3d_group_rows(const float* xyz, const float* new_xyz, const void* feats, int32_t dt_f, int64_t feat_stride,
                               const int32_t* idx, void* out, int32_t dt_o, int32_t B, int32_t N, int32_t C, int32_t np,
                               int32_t ns, int32_t Kp, void* stream) {
  PQ_DEVICE_GUARD(stream, xyz);
  PQ_CHECK_ARG(xyz && out && B >= 0 && N >= 1 && C >= 0 && np >= 1 && ns >= 1 && Kp >= 3 + C, "pq3d_group_rows: bad args");
  PQ_CHECK_ARG(C == 0 || (feats && feat_stride >= C), "pq3d_group_rows: features missing");
  PQ_CHECK_ARG(idx || ns == N, "pq3d_group_rows: without indices every point is a sample (ns == N)");
  PQ_CHECK_ARG((dt_f == PQ3D_F32 || dt_f == PQ3D_BF16) && (dt_o == PQ3D_F32 || dt_o == PQ3D_BF16), "pq3d_group_rows: bad dtype");
  const long total = (long)B * np * ns * Kp;
  if (total == 0) return 0;
  hipLaunchKernelGGL(group_rows_kernel, dim3(blocks_for(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream, xyz, new_xyz, feats,
                     dt_f, (long)feat_stride, idx, out, dt_o, N, C, np, ns, Kp, total);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_group_maxpool(const void* rows, void* out, int32_t dt, int64_t G, int32_t ns, int32_t C, void* stream) {
  PQ_DEVICE_GUARD(stream, rows);
  PQ_CHECK_ARG(rows && out && G >= 0 && ns >= 1 && C >= 1 && (dt == PQ3D_F32 || dt == PQ3D_BF16), "pq3d_group_maxpool: bad args");
  if (G == 0) return 0;
  hipLaunchKernelGGL(group_maxpool_kernel, dim3(blocks_for((long)G * C, 256, 4096)), dim3(256), 0, (hipStream_t)stream, rows, out,
                     dt, (long)G, ns, C);
  PQ_LAUNCH_CHECK();
  return 0;
}
