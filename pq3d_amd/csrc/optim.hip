// Optimizer side of the training step around the decoder (trainer/query3d_trainer.py:18-28): gradient-norm clipping
// (accelerator.clip_grad_norm_, trainer/build.py:144-145) + torch.optim.AdamW (optim/utils.py, betas from
// configs/instseg_sceneverse.yaml:72-76) + the LambdaLR schedules of optim/scheduler.py, on ONE flat fp32 parameter /
// gradient buffer.  Everything step-dependent (step count, learning rate, bias corrections, clip coefficient) lives
// in device memory and is advanced by a kernel, so a whole training step can sit in one HIP graph and never syncs the
// host.  Pure HBM streaming: 16 B read + 12 B written per parameter.
#include "common.h"

namespace {

constexpr int NPART = 1024;   // partial sums of the gradient norm (deterministic two-pass reduction)

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ partials) {
  __shared__ float red[4];
  float s = 0.f;
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = ((const float4*)g)[i];
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float t = g[(n4 << 2) + threadIdx.x]; s += t * t; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// one wave: finish the norm, advance the step counter, evaluate schedule / bias corrections / clip coefficient
__global__ void train_scalars_kernel(const pq3d_adamw_hp hp, long long* step, const float* partials, int npart,
                                     float* sc) {
  const int lane = threadIdx.x;
  double s = 0.0;
  for (int i = lane; i < npart; i += 64) s += (double)partials[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane != 0) return;
  const long long done = *step;        // optimizer / scheduler steps already taken
  const long long t = done + 1;
  *step = t;
  // LambdaLR: the factor in effect for optimizer step t is lambda(t - 1)   (optim/scheduler.py:5-17)
  double fac = 1.0;
  const double st = (double)done * (double)(hp.sched_stride > 0 ? hp.sched_stride : 1), wu = (double)hp.warmup_steps, tot = (double)hp.total_steps;
  if (hp.sched != PQ3D_SCHED_CONSTANT) {
    if (st <= wu && hp.warmup_steps > 0) fac = st / wu;
    else if (hp.sched == PQ3D_SCHED_WARMUP_COSINE) fac = fmax(0.5 * (1.0 + cos((st - wu) / (tot - wu) * 3.14159265358979323846)), 1e-5);
    else fac = pow((double)hp.sched_gamma, st / (tot - wu));   // warmup_exp
  }
  const double lr = (double)hp.lr * fac;
  const double bc1 = 1.0 - pow((double)hp.beta1, (double)t), bc2 = 1.0 - pow((double)hp.beta2, (double)t);
  const double norm = sqrt(s);
  double coef = 1.0;
  if (hp.max_grad_norm > 0.f) coef = fmin(1.0, (double)hp.max_grad_norm / (norm + 1e-6));   // clip_grad_norm_
  sc[0] = (float)lr;
  sc[1] = (float)(lr / bc1);          // step_size
  sc[2] = (float)(1.0 / sqrt(bc2));   // 1 / bias_correction2_sqrt
  sc[3] = (float)coef;
  sc[4] = (float)norm;                // total gradient norm before clipping (what the trainer logs)
  sc[5] = 0.f;
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long n, const pq3d_opt_segments segs,
                                                    const float* __restrict__ sc, float beta1, float beta2, float eps) {
  const float lr = sc[0], step_size = sc[1], inv_bc2s = sc[2], coef = sc[3];
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4 + (n & 3 ? 1 : 0); i += (long)gridDim.x * 256) {
    const long e0 = i << 2;
    const int cnt = (int)min(4L, n - e0);
    float pv[4], gv[4], mv[4], vv[4];
    if (cnt == 4) {
      const float4 a = ((const float4*)p)[i], b = ((const float4*)g)[i], c = ((const float4*)m)[i], d = ((const float4*)v)[i];
      pv[0] = a.x; pv[1] = a.y; pv[2] = a.z; pv[3] = a.w; gv[0] = b.x; gv[1] = b.y; gv[2] = b.z; gv[3] = b.w;
      mv[0] = c.x; mv[1] = c.y; mv[2] = c.z; mv[3] = c.w; vv[0] = d.x; vv[1] = d.y; vv[2] = d.z; vv[3] = d.w;
    } else {
      for (int j = 0; j < 4; ++j) {
        const bool ok = j < cnt;
        pv[j] = ok ? p[e0 + j] : 0.f; gv[j] = ok ? g[e0 + j] : 0.f; mv[j] = ok ? m[e0 + j] : 0.f; vv[j] = ok ? v[e0 + j] : 0.f;
      }
    }
    // parameter group of the first element; a group boundary inside these 4 elements is handled per element
    int sg = 0;
    while (sg + 1 < segs.n && e0 >= segs.end[sg]) ++sg;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      while (sg + 1 < segs.n && e0 + j >= segs.end[sg]) ++sg;
      const float lm = segs.lr_mul[sg], wd = segs.weight_decay[sg];
      if (lm < 0.f) continue;   // skip segment: a parameter without a gradient this step (torch.optim.AdamW: untouched)
      const float gr = gv[j] * coef;
      const float pp = pv[j] * (1.f - lr * lm * wd);              // param.mul_(1 - lr * weight_decay)
      mv[j] = mv[j] + (gr - mv[j]) * (1.f - beta1);               // exp_avg.lerp_(grad, 1 - beta1)
      vv[j] = vv[j] * beta2 + (1.f - beta2) * gr * gr;
      const float denom = sqrtf(vv[j]) * inv_bc2s + eps;
      pv[j] = pp - step_size * lm * (mv[j] / denom);
    }
    if (cnt == 4) {
      ((float4*)p)[i] = make_float4(pv[0], pv[1], pv[2], pv[3]);
      ((float4*)m)[i] = make_float4(mv[0], mv[1], mv[2], mv[3]);
      ((float4*)v)[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
    } else {
      for (int j = 0; j < cnt; ++j) { p[e0 + j] = pv[j]; m[e0 + j] = mv[j]; v[e0 + j] = vv[j]; }
    }
  }
}

}  // namespace

extern "C" int pq3d_sumsq_partials(const float* g, int64_t n, float* partials, void* stream) {
  PQ_DEVICE_GUARD(stream, g);
  PQ_CHECK_ARG(g && partials && n >= 0, "pq3d_sumsq_partials: bad args");
  PQ_CHECK_ARG((((uintptr_t)g) & 15) == 0, "pq3d_sumsq_partials: g must be 16-byte aligned");
  hipLaunchKernelGGL(sumsq_kernel, dim3(NPART), dim3(256), 0, (hipStream_t)stream, g, (long)n, partials);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_train_scalars(const pq3d_adamw_hp* hp, int64_t* step, const float* partials, float* scalars,
                                  void* stream) {
  PQ_DEVICE_GUARD(stream, step);
  PQ_CHECK_ARG(hp && step && partials && scalars, "pq3d_train_scalars: null argument");
  PQ_CHECK_ARG(hp->sched >= PQ3D_SCHED_CONSTANT && hp->sched <= PQ3D_SCHED_WARMUP_EXP, "pq3d_train_scalars: bad schedule");
  PQ_CHECK_ARG(hp->sched == PQ3D_SCHED_CONSTANT || hp->total_steps > hp->warmup_steps,
               "pq3d_train_scalars: total_steps must exceed warmup_steps");
  hipLaunchKernelGGL(train_scalars_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, *hp, (long long*)step, partials,
                     NPART, scalars);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_adamw(float* p, const float* g, float* m, float* v, int64_t n, const pq3d_adamw_hp* hp,
                          const pq3d_opt_segments* segs, const float* scalars, void* stream) {
  PQ_DEVICE_GUARD(stream, p);
  PQ_CHECK_ARG(p && g && m && v && hp && segs && scalars && n >= 0, "pq3d_adamw: bad args");
  PQ_CHECK_ARG(segs->n >= 1 && segs->n <= PQ3D_MAX_OPT_SEGMENTS && segs->end[segs->n - 1] >= n,
               "pq3d_adamw: segment table must cover [0, n)");
  PQ_CHECK_ARG(((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0,
               "pq3d_adamw: buffers must be 16-byte aligned");
  if (n == 0) return 0;
  long nb = (n / 4 + 255) / 256;
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n, *segs,
                     scalars, hp->beta1, hp->beta2, hp->eps);
  PQ_LAUNCH_CHECK();
  return 0;
}
