// Instance-segmentation set criterion around the path's outputs: the Hungarian cost matrix of
// HungarianMatcher.memory_efficient_forward (modules/third_party/mask3d/matcher.py:104-184) and the matched losses of
// SetCriterion (criterion.py:136-206), re-derived so that ONE grouped fp32 MFMA GEMM per prediction layer carries all
// the segment-dimension work:
//     pos - neg = softplus(-x) - softplus(x) = -x   =>
//     cost_mask[q,t] = ( sum_s softplus(x_sq) - (T X)[t,q] ) / S            (batch_sigmoid_ce_loss, matcher.py:37-60)
//     cost_dice[q,t] = 1 - (2 (T sigma(X))[t,q] + 1) / (sum_s sigma(x_sq) + sum_s T_ts + 1)      (matcher.py:12-28)
// with X = mask logits [S, Nq] of one scene (segments first, as the mask head writes them), T = target masks [Nt, S].
// With num_points = -1 (all points, the shipped config) the matched losses sigmoid_ce_loss / dice_loss
// (criterion.py:27-70) are exactly the matched ENTRIES of those two matrices, so the forward losses are gathers and
// only the gradient needs a second pass over the logits.  The assignment itself (scipy LSA in the reference) stays on
// the host.  Kernels here are HBM-bound streaming / reductions; the contraction is pq3d_gemm (ct = F32).
#include "common.h"

namespace {

constexpr int PREP_ROWS = 256;   // segments per block of the prep kernel
constexpr int GRAD_ROWS = 16;    // segments per block of the gradient kernel (LDS tile 2 x 16 x (Nq+1) floats)

PQ_DEV float softplus_f(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }
PQ_DEV float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// sig[b,s,q] = sigma(x) (0 for s >= len_b); partial column sums over this block's rows of softplus(x) and sigma(x)
__global__ __launch_bounds__(256) void mask_cost_prep_kernel(const pq3d_mask_prep_desc d) {
  __shared__ float red[2][4][64];
  const int Ns = d.Ns, Nq = d.Nq, nsplit = d.nsplit;
  const int layer = blockIdx.z / d.B, bb = blockIdx.z % d.B;
  const float* __restrict__ X = d.X[layer];
  float* __restrict__ sig = d.sig + (long)layer * d.B * Ns * Nq;
  float* __restrict__ sp_part = d.sp_part + (long)layer * d.B * nsplit * Nq;
  float* __restrict__ sg_part = d.sg_part + (long)layer * d.B * nsplit * Nq;
  const int b = bb, split = blockIdx.y, q = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
  const int len = min(d.seg_len[b], Ns);
  const int s0 = split * PREP_ROWS, s1 = min(s0 + PREP_ROWS, Ns);
  float sp = 0.f, sg = 0.f;
  if (q < Nq) {
    for (int s = s0 + rg; s < s1; s += 4) {
      const long i = ((long)b * Ns + s) * Nq + q;
      const float x = X[i];
      const bool in = s < len;
      const float g = in ? sigmoid_f(x) : 0.f;
      sig[i] = g;
      sg += g;
      sp += in ? softplus_f(x) : 0.f;
    }
  }
  red[0][rg][threadIdx.x & 63] = sp;
  red[1][rg][threadIdx.x & 63] = sg;
  __syncthreads();
  if (threadIdx.x < 64 && q < Nq) {
    const long o = ((long)b * nsplit + split) * Nq + q;
    sp_part[o] = red[0][0][threadIdx.x] + red[0][1][threadIdx.x] + red[0][2][threadIdx.x] + red[0][3][threadIdx.x];
    sg_part[o] = red[1][0][threadIdx.x] + red[1][1][threadIdx.x] + red[1][2][threadIdx.x] + red[1][3][threadIdx.x];
  }
}

// one wave per (scene, query): class-probability row statistics, then the Nt costs of that query
__global__ __launch_bounds__(64) void match_cost_kernel(const pq3d_match_cost_desc dd) {
  // per-layer views of the stacked buffers
  struct {
    int B, Nq, Nt, Ns, C, nsplit;
    float w_class, w_mask, w_dice;
    long ignore_label;
    const float *TX, *TS, *sp_part, *sg_part, *t_sum, *cls_logits;
    const int32_t *seg_len, *n_inst;
    const int64_t* labels;
    float *cost, *cost_mask, *cost_dice;
  } d;
  const int layer = blockIdx.z;
  d.B = dd.B; d.Nq = dd.Nq; d.Nt = dd.Nt; d.Ns = dd.Ns; d.C = dd.C; d.nsplit = dd.nsplit;
  d.w_class = dd.w_class; d.w_mask = dd.w_mask; d.w_dice = dd.w_dice; d.ignore_label = dd.ignore_label;
  const long txs = (long)dd.B * dd.Nt * dd.Nq, prt = (long)dd.B * dd.nsplit * dd.Nq, cst = (long)dd.B * dd.Nq * dd.Nt;
  d.TX = dd.TXS + (long)layer * 2 * txs; d.TS = d.TX + txs;
  d.sp_part = dd.sp_part + (long)layer * prt; d.sg_part = dd.sg_part + (long)layer * prt;
  d.t_sum = dd.t_sum; d.seg_len = dd.seg_len; d.n_inst = dd.n_inst; d.labels = dd.labels;
  d.cls_logits = dd.cls_logits[layer];
  d.cost = dd.cost + (long)layer * 3 * cst; d.cost_mask = d.cost + cst; d.cost_dice = d.cost + 2 * cst;
  const int q = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  const float* lg = d.cls_logits + ((long)b * d.Nq + q) * d.C;
  float mx = -INFINITY;
  for (int c = lane; c < d.C; c += 64) mx = fmaxf(mx, lg[c]);
  mx = wave_max(mx);
  float se = 0.f;
  for (int c = lane; c < d.C; c += 64) se += expf(lg[c] - mx);
  se = wave_sum(se);
  float sp = 0.f, sg = 0.f;   // finish the deterministic two-stage column sums
  for (int i = 0; i < d.nsplit; ++i) {
    sp += d.sp_part[((long)b * d.nsplit + i) * d.Nq + q];
    sg += d.sg_part[((long)b * d.nsplit + i) * d.Nq + q];
  }
  const int nt = d.n_inst[b];
  const float inv_s = 1.f / (float)min(d.seg_len[b], d.Ns);
  for (int t = lane; t < d.Nt; t += 64) {
    const long o = ((long)b * d.Nq + q) * d.Nt + t;
    if (t >= nt) { d.cost[o] = 0.f; d.cost_mask[o] = 0.f; d.cost_dice[o] = 0.f; continue; }
    const long gi = ((long)b * d.Nt + t) * d.Nq + q;
    const float cm = (sp - d.TX[gi]) * inv_s;
    const float cd = 1.f - (2.f * d.TS[gi] + 1.f) / (sg + d.t_sum[(long)b * d.Nt + t] + 1.f);
    const long lab = d.labels[(long)b * d.Nt + t];
    const float cc = lab == d.ignore_label ? -1.f : -expf(lg[lab] - mx) / se;   // -softmax prob; "perfect match" if ignored
    d.cost_mask[o] = cm;
    d.cost_dice[o] = cd;
    d.cost[o] = d.w_mask * cm + d.w_class * cc + d.w_dice * cd;
  }
}

// dX[b,s,q_j] = gm_b * (sigma - T)/S_b + gd_b * sigma(1-sigma) * (-(2 T D - (2 TS + 1)) / D^2),  D = sig_sum + t_sum + 1;
// every other entry of dX is zero.  Block = (16-segment tile, scene, layer): X / dX tiles go through LDS so that both the
// [S, Nq]-major logits and the [Nt, S]-major targets are read and written coalesced.
__global__ __launch_bounds__(256) void matched_mask_grad_kernel(const pq3d_mask_grad_desc dd) {
  extern __shared__ float lds[];   // [GRAD_ROWS][Nq + 1] sigma tile, then the same for the gradient tile
  struct {
    int B, Ns, Nq, Nt, Nm;
    const float *sig, *T, *TS, *sig_sum, *t_sum, *g_mask, *g_dice;
    const int32_t *seg_len, *q_idx, *t_idx, *n_match;
    float* dX;
  } d;
  const int layer = blockIdx.z;
  d.B = dd.B; d.Ns = dd.Ns; d.Nq = dd.Nq; d.Nt = dd.Nt; d.Nm = dd.Nm;
  d.sig = dd.sig + (long)layer * dd.B * dd.Ns * dd.Nq;
  d.T = dd.T; d.t_sum = dd.t_sum; d.seg_len = dd.seg_len;
  d.TS = dd.TXS + ((long)layer * 2 + 1) * dd.B * dd.Nt * dd.Nq;
  d.sig_sum = dd.sig_sum + (long)layer * dd.B * dd.Nq;
  d.q_idx = dd.q_idx + (long)layer * dd.B * dd.Nm; d.t_idx = dd.t_idx + (long)layer * dd.B * dd.Nm;
  d.n_match = dd.n_match + (long)layer * dd.B;
  d.g_mask = dd.g + (long)layer * 2 * dd.B; d.g_dice = d.g_mask + dd.B;
  d.dX = dd.dX[layer];
  const int b = blockIdx.y, s0 = blockIdx.x * GRAD_ROWS, tid = threadIdx.x;
  const int ld = d.Nq + 1;
  float* sg_t = lds;
  float* g_t = lds + GRAD_ROWS * ld;
  const int len = min(d.seg_len[b], d.Ns);
  for (int i = tid; i < GRAD_ROWS * d.Nq; i += 256) {
    const int r = i / d.Nq, c = i % d.Nq;
    const int s = s0 + r;
    sg_t[r * ld + c] = s < d.Ns ? d.sig[((long)b * d.Ns + s) * d.Nq + c] : 0.f;
    g_t[r * ld + c] = 0.f;
  }
  __syncthreads();
  const int nm = d.n_match[b];
  const float gm = d.g_mask[b] / (float)len, gd = d.g_dice[b];
  const int sl = tid % GRAD_ROWS, s = s0 + sl;
  if (s < len) {
    for (int j = tid / GRAD_ROWS; j < nm; j += 256 / GRAD_ROWS) {
      const int q = d.q_idx[(long)b * d.Nm + j], t = d.t_idx[(long)b * d.Nm + j];
      const float sg = sg_t[sl * ld + q];
      const float tv = d.T[((long)b * d.Nt + t) * d.Ns + s];
      const float ssum = d.sig_sum[(long)b * d.Nq + q], D = ssum + d.t_sum[(long)b * d.Nt + t] + 1.f;
      const float num = 2.f * d.TS[((long)b * d.Nt + t) * d.Nq + q] + 1.f;
      g_t[sl * ld + q] = gm * (sg - tv) + gd * sg * (1.f - sg) * (-(2.f * tv * D - num) / (D * D));
    }
  }
  __syncthreads();
  for (int i = tid; i < GRAD_ROWS * d.Nq; i += 256) {
    const int r = i / d.Nq, c = i % d.Nq;
    if (s0 + r < d.Ns) d.dX[((long)b * d.Ns + s0 + r) * d.Nq + c] = g_t[r * ld + c];
  }
}

// ---- padded (no matching) mask losses: batch_mask_loss / batch_dice_loss (optim/loss/instseg_loss.py:54-85) -------
// X [B, S, N] mask logits (segments first), T / P [B, N, S] targets / padding mask.  Per (scene, instance):
//   sums[0] = sum_s bce(x, t) p,  sums[1] = sum_s p,  sums[2] = sum_s sigma(x) t p,  sums[3] = sum_s (sigma(x) + t) p
// Block = (64-segment tile, scene): the X tile goes through LDS so both the [S, N]-major logits and the [N, S]-major
// targets are read coalesced; per-tile partial sums go to part[B, tiles, N, 4] (deterministic two-stage reduction).
__global__ __launch_bounds__(256) void padded_mask_sums_kernel(const float* __restrict__ X, const float* __restrict__ T,
                                                               const uint8_t* __restrict__ P, float* __restrict__ part,
                                                               int S, int N, int ntiles) {
  extern __shared__ float lds[];   // [64][N + 1]
  const int b = blockIdx.y, tile = blockIdx.x, s0 = tile * 64, tid = threadIdx.x, ld = N + 1;
  for (int i = tid; i < 64 * N; i += 256) {
    const int r = i / N, c = i % N;
    lds[r * ld + c] = s0 + r < S ? X[((long)b * S + s0 + r) * N + c] : 0.f;
  }
  __syncthreads();
  const int sl = tid & 63, s = s0 + sl;
  for (int n = tid >> 6; n < N; n += 4) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (s < S) {
      const long ti = ((long)b * N + n) * S + s;
      const float p = P[ti] ? 1.f : 0.f, t = T[ti], x = lds[sl * ld + n];
      const float sg = sigmoid_f(x);
      v[0] = (softplus_f(x) - x * t) * p;   // BCE with logits = softplus(x) - x t
      v[1] = p;
      v[2] = sg * t * p;
      v[3] = (sg + t) * p;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = wave_sum(v[j]);
    if (sl == 0) *(float4*)&part[(((long)b * ntiles + tile) * N + n) * 4] = make_float4(v[0], v[1], v[2], v[3]);
  }
}
// dX[b,s,n] = p * ( gm[b,n] (sigma - t) + gd[b,n] sigma (1 - sigma) (-(2 t U - I2) / U^2) ),  U = sums[3] + 1e-6,
// I2 = 2 sums[2] + 1e-6 (dice_score = I2 / U)
__global__ __launch_bounds__(256) void padded_mask_grad_kernel(const float* __restrict__ X, const float* __restrict__ T,
                                                               const uint8_t* __restrict__ P, const float* __restrict__ sums,
                                                               const float* __restrict__ gm, const float* __restrict__ gd,
                                                               float* __restrict__ dX, int S, int N) {
  extern __shared__ float lds[];   // [GRAD_ROWS][N + 1]
  const int b = blockIdx.y, s0 = blockIdx.x * GRAD_ROWS, tid = threadIdx.x, ld = N + 1;
  for (int i = tid; i < GRAD_ROWS * N; i += 256) {
    const int r = i / N, c = i % N;
    lds[r * ld + c] = s0 + r < S ? X[((long)b * S + s0 + r) * N + c] : 0.f;
  }
  __syncthreads();
  const int sl = tid % GRAD_ROWS, s = s0 + sl;
  if (s < S) {
    for (int n = tid / GRAD_ROWS; n < N; n += 256 / GRAD_ROWS) {
      const long ti = ((long)b * N + n) * S + s;
      float g = 0.f;
      if (P[ti]) {
        const float t = T[ti], sg = sigmoid_f(lds[sl * ld + n]);
        const float* sm = sums + ((long)b * N + n) * 4;
        const float U = sm[3] + 1e-6f, I2 = 2.f * sm[2] + 1e-6f;
        g = gm[(long)b * N + n] * (sg - t) + gd[(long)b * N + n] * sg * (1.f - sg) * (-(2.f * t * U - I2) / (U * U));
      }
      lds[sl * ld + n] = g;   // each (row, n) is read and rewritten by the same thread
    }
  }
  __syncthreads();
  for (int i = tid; i < GRAD_ROWS * N; i += 256) {
    const int r = i / N, c = i % N;
    if (s0 + r < S) dX[((long)b * S + s0 + r) * N + c] = lds[r * ld + c];
  }
}

// cross entropy over rows (F.cross_entropy(..., ignore_index)): one wave per row
__global__ __launch_bounds__(256) void ce_fwd_kernel(const pq3d_ce_desc d) {
  const long R = d.R, ignore = d.ignore_index;
  const int C = d.C, layer = blockIdx.y;
  const float* __restrict__ logits = d.logits[layer];
  const int64_t* __restrict__ target = d.target + (long)layer * R;
  float* __restrict__ row_loss = d.row_loss + (long)layer * R;
  float* __restrict__ lse = d.lse + (long)layer * R;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const int lane = threadIdx.x & 63;
  const float* x = logits + row * C;
  float mx = -INFINITY;
  for (int c = lane; c < C; c += 64) mx = fmaxf(mx, x[c]);
  mx = wave_max(mx);
  float se = 0.f;
  for (int c = lane; c < C; c += 64) se += expf(x[c] - mx);
  se = wave_sum(se);
  if (lane == 0) {
    const float l = mx + logf(se);
    lse[row] = l;
    const long t = target[row];
    // torch's nll_loss device-asserts 0 <= t < C; here an out-of-range label poisons the loss (NaN) instead of reading
    // out of bounds -- loud at the first .item()/isfinite check, never a silent wrong number
    row_loss[row] = t == ignore ? 0.f : ((unsigned long)t < (unsigned long)C ? l - x[t] : NAN);
  }
}
__global__ __launch_bounds__(256) void ce_bwd_kernel(const pq3d_ce_desc d) {
  const long R = d.R, ignore = d.ignore_index;
  const int C = d.C, layer = blockIdx.y;
  const float* __restrict__ logits = d.logits[layer];
  const int64_t* __restrict__ target = d.target + (long)layer * R;
  const float* __restrict__ lse = d.lse + (long)layer * R;
  float* __restrict__ dlogits = d.dlogits[layer];
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const int lane = threadIdx.x & 63;
  const long t = target[row];
  const float sc = t == ignore ? 0.f : d.scale[layer] * (d.scale_mul ? d.scale_mul[layer] : 1.f);
  const float l = lse[row];
  for (int c = lane; c < C; c += 64) {
    const float x = logits[row * C + c];
    dlogits[row * C + c] = sc == 0.f ? 0.f : sc * (expf(x - l) - (c == t ? 1.f : 0.f));
  }
}


// ---- wide rows (the caption head's 32128-way LM head, generation_head.py:24-27 + generation_loss): ONE WORKGROUP per row,
// 16-byte loads, a single pass with an online (max, sum) per thread.  A 512 x 32128 fp32 logit matrix is 66 MB: the forward
// reads it once, the backward reads it once and writes the gradient once (streaming stores).
PQ_DEV void ce_online(float& m, float& s, float v) {
  const float mn = fmaxf(m, v);
  s = s * __expf(m - mn) + __expf(v - mn);
  m = mn;
}
__global__ __launch_bounds__(256) void ce_fwd_wide_kernel(const pq3d_ce_desc d) {
  __shared__ float sm[4], ss[4];
  const long R = d.R, ignore = d.ignore_index, row = blockIdx.x;
  const int C = d.C, layer = blockIdx.y, tid = threadIdx.x;
  const float* __restrict__ x = d.logits[layer] + row * C;
  float m = -INFINITY, s = 0.f;
  const int head = (int)(((16 - ((uintptr_t)x & 15)) & 15) >> 2);          // floats before the first 16-byte boundary
  const int h = head < C ? head : C, nv = (C - h) >> 2;
  if (tid < h) ce_online(m, s, x[tid]);
  const f32x4* x4 = (const f32x4*)(x + h);
  for (int i = tid; i < nv; i += 256) {
    const f32x4 v = __builtin_nontemporal_load(x4 + i);
    const float mv = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), mn = fmaxf(m, mv);
    s = s * __expf(m - mn) + ((__expf(v[0] - mn) + __expf(v[1] - mn)) + (__expf(v[2] - mn) + __expf(v[3] - mn)));
    m = mn;
  }
  for (int c = h + 4 * nv + tid; c < C; c += 256) ce_online(m, s, x[c]);
  const float wm = wave_max(m);
  s = wave_sum(m == -INFINITY ? 0.f : s * __expf(m - wm));
  if ((tid & 63) == 0) { sm[tid >> 6] = wm; ss[tid >> 6] = s; }
  __syncthreads();
  if (tid == 0) {
    const float M = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    float S = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) S += sm[w] == -INFINITY ? 0.f : ss[w] * __expf(sm[w] - M);
    const float l = M + logf(S);
    d.lse[(long)layer * R + row] = l;
    const long t = d.target[(long)layer * R + row];
    d.row_loss[(long)layer * R + row] = t == ignore ? 0.f : ((unsigned long)t < (unsigned long)C ? l - x[t] : NAN);
  }
}
__global__ __launch_bounds__(256) void ce_bwd_wide_kernel(const pq3d_ce_desc d) {
  const long R = d.R, ignore = d.ignore_index, row = blockIdx.x;
  const int C = d.C, layer = blockIdx.y, tid = threadIdx.x;
  const float* __restrict__ x = d.logits[layer] + row * C;
  float* __restrict__ dl = d.dlogits[layer] + row * C;
  const long t = d.target[(long)layer * R + row];
  const float sc = t == ignore ? 0.f : d.scale[layer] * (d.scale_mul ? d.scale_mul[layer] : 1.f);
  const float l = d.lse[(long)layer * R + row];
  const int head = (int)(((16 - ((uintptr_t)x & 15)) & 15) >> 2);
  const int h = head < C ? head : C, nv = (C - h) >> 2;
  const bool al = (((uintptr_t)(dl + h)) & 15) == 0;
  auto one = [&](int c) { dl[c] = sc == 0.f ? 0.f : sc * (__expf(x[c] - l) - (c == t ? 1.f : 0.f)); };
  if (tid < h) one(tid);
  if (al) {
    const f32x4* x4 = (const f32x4*)(x + h);
    f32x4* d4 = (f32x4*)(dl + h);
    for (int i = tid; i < nv; i += 256) {
      const f32x4 v = __builtin_nontemporal_load(x4 + i);
      f32x4 g;
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] = sc == 0.f ? 0.f : sc * (__expf(v[j] - l) - ((long)(h + 4 * i + j) == t ? 1.f : 0.f));
      __builtin_nontemporal_store(g, d4 + i);
    }
  } else {
    for (int c = h + tid; c < h + 4 * nv; c += 256) one(c);
  }
  for (int c = h + 4 * nv + tid; c < C; c += 256) one(c);
}
// mean over the kept rows: loss[layer] = sum(row_loss) / max(kept, 1) ... F.cross_entropy's 'mean' divides by the kept count
// (0 kept rows -> nan there; here 0 / 0 -> nan as well), inv_count[layer] = 1 / kept for the backward.  One block per layer,
// fixed summation order.
__global__ __launch_bounds__(256) void ce_mean_kernel(const pq3d_ce_desc d, float* __restrict__ loss, float* __restrict__ inv_count,
                                                      const float* __restrict__ addend) {
  __shared__ float ps[4], pc[4];
  const long R = d.R, ignore = d.ignore_index;
  const int layer = blockIdx.x, tid = threadIdx.x;
  float s = 0.f, c = 0.f;
  for (long r = tid; r < R; r += 256) {
    s += d.row_loss[(long)layer * R + r];
    c += d.target[(long)layer * R + r] == ignore ? 0.f : 1.f;
  }
  s = wave_sum(s); c = wave_sum(c);
  if ((tid & 63) == 0) { ps[tid >> 6] = s; pc[tid >> 6] = c; }
  __syncthreads();
  if (tid == 0) {
    const float S = (ps[0] + ps[1]) + (ps[2] + ps[3]), Cn = (pc[0] + pc[1]) + (pc[2] + pc[3]);
    loss[layer] = S / Cn + (addend ? addend[layer] : 0.f);
    inv_count[layer] = 1.f / Cn;
  }
}

}  // namespace

static int check_layers(int layers, const char* who) {
  if (layers < 1 || layers > PQ3D_MAX_GROUPS) { pq3d_set_error(who); return PQ3D_ERR_ARG; }
  return 0;
}

extern "C" int pq3d_mask_cost_prep(const pq3d_mask_prep_desc* dp, void* stream) {
  PQ_DEVICE_GUARD(stream, nullptr);
  PQ_CHECK_ARG(dp != nullptr, "pq3d_mask_cost_prep: null descriptor");
  const pq3d_mask_prep_desc d = *dp;
  if (int e = check_layers(d.layers, "pq3d_mask_cost_prep: layers must be in [1, PQ3D_MAX_GROUPS]")) return e;
  PQ_CHECK_ARG(d.seg_len && d.sig && d.sp_part && d.sg_part && d.B >= 0 && d.Ns >= 1 && d.Nq >= 1 &&
               d.nsplit == (d.Ns + PREP_ROWS - 1) / PREP_ROWS, "pq3d_mask_cost_prep: bad args");
  for (int l = 0; l < d.layers; ++l) PQ_CHECK_ARG(d.X[l] != nullptr, "pq3d_mask_cost_prep: null X");
  if (d.B == 0) return 0;
  hipLaunchKernelGGL(mask_cost_prep_kernel, dim3((d.Nq + 63) / 64, d.nsplit, d.layers * d.B), dim3(256), 0,
                     (hipStream_t)stream, d);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int32_t pq3d_mask_cost_nsplit(int32_t Ns) { return (Ns + PREP_ROWS - 1) / PREP_ROWS; }

extern "C" int pq3d_match_cost(const pq3d_match_cost_desc* dp, void* stream) {
  PQ_DEVICE_GUARD(stream, nullptr);
  PQ_CHECK_ARG(dp != nullptr, "pq3d_match_cost: null descriptor");
  const pq3d_match_cost_desc d = *dp;
  if (int e = check_layers(d.layers, "pq3d_match_cost: layers must be in [1, PQ3D_MAX_GROUPS]")) return e;
  PQ_CHECK_ARG(d.TXS && d.sp_part && d.sg_part && d.t_sum && d.seg_len && d.n_inst && d.labels && d.cost,
               "pq3d_match_cost: null pointer");
  for (int l = 0; l < d.layers; ++l) PQ_CHECK_ARG(d.cls_logits[l] != nullptr, "pq3d_match_cost: null cls_logits");
  PQ_CHECK_ARG(d.B >= 0 && d.Nq >= 1 && d.Nt >= 1 && d.C >= 1 && d.Ns >= 1 && d.nsplit == (d.Ns + PREP_ROWS - 1) / PREP_ROWS,
               "pq3d_match_cost: bad sizes");
  if (d.B == 0) return 0;
  hipLaunchKernelGGL(match_cost_kernel, dim3(d.Nq, d.B, d.layers), dim3(64), 0, (hipStream_t)stream, d);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_matched_mask_grad(const pq3d_mask_grad_desc* dp, void* stream) {
  PQ_DEVICE_GUARD(stream, nullptr);
  PQ_CHECK_ARG(dp != nullptr, "pq3d_matched_mask_grad: null descriptor");
  const pq3d_mask_grad_desc d = *dp;
  if (int e = check_layers(d.layers, "pq3d_matched_mask_grad: layers must be in [1, PQ3D_MAX_GROUPS]")) return e;
  PQ_CHECK_ARG(d.sig && d.T && d.TXS && d.sig_sum && d.t_sum && d.seg_len && d.q_idx && d.t_idx && d.n_match && d.g,
               "pq3d_matched_mask_grad: null pointer");
  for (int l = 0; l < d.layers; ++l) PQ_CHECK_ARG(d.dX[l] != nullptr, "pq3d_matched_mask_grad: null dX");
  PQ_CHECK_ARG(d.B >= 0 && d.Ns >= 1 && d.Nq >= 1 && d.Nt >= 1 && d.Nm >= 1, "pq3d_matched_mask_grad: bad sizes");
  const size_t lds = (size_t)2 * GRAD_ROWS * (d.Nq + 1) * sizeof(float);
  PQ_CHECK_ARG(lds <= 160 * 1024, "pq3d_matched_mask_grad: Nq too large for the LDS tile");
  if (d.B == 0) return 0;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)matched_mask_grad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { pq3d_set_error(hipGetErrorString(e)); return (int)e; }
  }
  hipLaunchKernelGGL(matched_mask_grad_kernel, dim3((d.Ns + GRAD_ROWS - 1) / GRAD_ROWS, d.B, d.layers), dim3(256), lds, (hipStream_t)stream, d);
  PQ_LAUNCH_CHECK();
  return 0;
}

static int check_ce(const pq3d_ce_desc& d, bool bwd) {
  if (int e = check_layers(d.layers, "pq3d_cross_entropy: layers must be in [1, PQ3D_MAX_GROUPS]")) return e;
  PQ_CHECK_ARG(d.target && d.lse && d.R >= 0 && d.C >= 1, "pq3d_cross_entropy: bad args");
  PQ_CHECK_ARG(bwd ? d.scale != nullptr : d.row_loss != nullptr, "pq3d_cross_entropy: null row_loss / scale");
  for (int l = 0; l < d.layers; ++l)
    PQ_CHECK_ARG(d.logits[l] && (!bwd || d.dlogits[l]), "pq3d_cross_entropy: null logits / dlogits");
  return 0;
}
extern "C" int pq3d_cross_entropy_fwd(const pq3d_ce_desc* dp, void* stream) {
  PQ_DEVICE_GUARD(stream, nullptr);
  PQ_CHECK_ARG(dp != nullptr, "pq3d_cross_entropy_fwd: null descriptor");
  const pq3d_ce_desc d = *dp;
  if (int e = check_ce(d, false)) return e;
  if (d.R == 0) return 0;
  if (d.C >= 1024) hipLaunchKernelGGL(ce_fwd_wide_kernel, dim3((unsigned)d.R, d.layers), dim3(256), 0, (hipStream_t)stream, d);
  else hipLaunchKernelGGL(ce_fwd_kernel, dim3((unsigned)((d.R + 3) / 4), d.layers), dim3(256), 0, (hipStream_t)stream, d);
  PQ_LAUNCH_CHECK();
  return 0;
}
extern "C" int pq3d_cross_entropy_mean(const pq3d_ce_desc* dp, float* loss, float* inv_count, const float* addend, void* stream) {
  PQ_DEVICE_GUARD(stream, loss);
  PQ_CHECK_ARG(dp != nullptr && loss && inv_count, "pq3d_cross_entropy_mean: null argument");
  const pq3d_ce_desc d = *dp;
  if (int e = check_layers(d.layers, "pq3d_cross_entropy_mean: layers must be in [1, PQ3D_MAX_GROUPS]")) return e;
  PQ_CHECK_ARG(d.target && d.row_loss && d.R >= 0, "pq3d_cross_entropy_mean: bad args");
  hipLaunchKernelGGL(ce_mean_kernel, dim3(d.layers), dim3(256), 0, (hipStream_t)stream, d, loss, inv_count, addend);
  PQ_LAUNCH_CHECK();
  return 0;
}
extern "C" int pq3d_cross_entropy_bwd(const pq3d_ce_desc* dp, void* stream) {
  PQ_DEVICE_GUARD(stream, nullptr);
  PQ_CHECK_ARG(dp != nullptr, "pq3d_cross_entropy_bwd: null descriptor");
  const pq3d_ce_desc d = *dp;
  if (int e = check_ce(d, true)) return e;
  if (d.R == 0) return 0;
  if (d.C >= 1024) hipLaunchKernelGGL(ce_bwd_wide_kernel, dim3((unsigned)d.R, d.layers), dim3(256), 0, (hipStream_t)stream, d);
  else hipLaunchKernelGGL(ce_bwd_kernel, dim3((unsigned)((d.R + 3) / 4), d.layers), dim3(256), 0, (hipStream_t)stream, d);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_padded_mask_sums(const float* X, const float* T, const uint8_t* P, float* part, int32_t B, int32_t S,
                                     int32_t N, void* stream) {
  PQ_DEVICE_GUARD(stream, X);
  PQ_CHECK_ARG(X && T && P && part && B >= 0 && S >= 1 && N >= 1, "pq3d_padded_mask_sums: bad args");
  const size_t lds = (size_t)64 * (N + 1) * sizeof(float);
  PQ_CHECK_ARG(lds <= 160 * 1024, "pq3d_padded_mask_sums: N too large for the LDS tile");
  if (B == 0) return 0;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)padded_mask_sums_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { pq3d_set_error(hipGetErrorString(e)); return (int)e; }
  }
  const int ntiles = (S + 63) / 64;
  hipLaunchKernelGGL(padded_mask_sums_kernel, dim3(ntiles, B), dim3(256), lds, (hipStream_t)stream, X, T, P, part, S, N, ntiles);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_padded_mask_grad(const float* X, const float* T, const uint8_t* P, const float* sums, const float* gm,
                                     const float* gd, float* dX, int32_t B, int32_t S, int32_t N, void* stream) {
  PQ_DEVICE_GUARD(stream, X);
  PQ_CHECK_ARG(X && T && P && sums && gm && gd && dX && B >= 0 && S >= 1 && N >= 1, "pq3d_padded_mask_grad: bad args");
  const size_t lds = (size_t)GRAD_ROWS * (N + 1) * sizeof(float);
  PQ_CHECK_ARG(lds <= 64 * 1024, "pq3d_padded_mask_grad: N too large for the LDS tile");
  if (B == 0) return 0;
  hipLaunchKernelGGL(padded_mask_grad_kernel, dim3((S + GRAD_ROWS - 1) / GRAD_ROWS, B), dim3(256), lds, (hipStream_t)stream,
                     X, T, P, sums, gm, gd, dX, S, N);
  PQ_LAUNCH_CHECK();
  return 0;
}
