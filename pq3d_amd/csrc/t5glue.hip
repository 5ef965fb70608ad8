// Glue of the caption head's teacher-forced T5 decoder (pq3d_amd/t5.py; reference: modules/heads/generation_head.py:20-30
// driving HF T5ForConditionalGeneration): what the HF model does with framework elementwise ops around its layers, as ONE
// launch forward and ONE backward.  At config 5 these were ~20 at::native launches of ~5 us each per step.
//   pq3d_t5_prep      decoder_input_ids = shift_right(labels) (T5's _shift_right: [start, labels[:-1]], -100 -> pad),
//                     self-attention bias [B, H, T, T] = relative_attention_bias[bucket(q, k)] with -inf above the diagonal
//                     (position bias + causal mask, shared by all layers), key-padding bytes of the encoder tokens = !valid
//   pq3d_t5_bias_bwd  d relative_attention_bias[nb, h] (+)= sum over scenes and the causal (q, k) pairs of bucket nb of
//                     d bias[b, h, q, k]; one workgroup per head, fixed summation order
//   pq3d_embedding_drop_fwd / _bwd_acc   token embedding rows with the embedding dropout fused in (and its gradient)
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void t5_prep_kernel(const int64_t* __restrict__ labels, long start_id, long pad_id,
                                                      const float* __restrict__ rel, const int64_t* __restrict__ buckets,
                                                      const uint8_t* __restrict__ enc_valid, int64_t* __restrict__ ids,
                                                      float* __restrict__ bias, uint8_t* __restrict__ kpm, int B, int T, int H,
                                                      long N) {
  const long nb = (long)B * H * T * T, ni = (long)B * T, nk = kpm ? (long)B * N : 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nb + ni + nk; i += (long)gridDim.x * 256) {
    if (i < nb) {
      const int k = (int)(i % T), q = (int)((i / T) % T), h = (int)((i / ((long)T * T)) % H);
      bias[i] = k > q ? -INFINITY : rel[buckets[q * T + k] * H + h];
    } else if (i < nb + ni) {
      const long j = i - nb;
      const int t = (int)(j % T);
      long v = t == 0 ? start_id : labels[j - 1];
      ids[j] = v == -100 ? pad_id : v;
    } else {
      const long j = i - nb - ni;
      kpm[j] = enc_valid[j] ? 0 : 1;
    }
  }
}

__global__ __launch_bounds__(256) void t5_bias_bwd_kernel(const float* __restrict__ dbias, const int64_t* __restrict__ buckets,
                                                          float* __restrict__ drel, int B, int T, int H, int NB, int accumulate) {
  extern __shared__ float S[];                 // [T * T] sums over the scenes, then [T * T] bucket ids as bytes
  uint8_t* bk = (uint8_t*)(S + T * T);
  const int h = blockIdx.x, tid = threadIdx.x, TT = T * T;
  for (int p = tid; p < TT; p += 256) {
    const int q = p / T, k = p % T;
    float s = 0.f;
    if (k <= q) {
      const float* src = dbias + ((long)h * T + q) * T + k;
      const long sb = (long)H * T * T;
      for (int b0 = 0; b0 < B; b0 += 8) {       // 8 independent loads in flight (one per scene), summed in scene order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = b0 + u < B ? src[(long)(b0 + u) * sb] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
      }
    }
    S[p] = s;
    bk[p] = (uint8_t)buckets[p];
  }
  __syncthreads();
  // 8 lanes per bucket: lane j sums the pairs p = j, j + 8, ..., then a fixed 3-step tree over the 8 partial sums
  for (int nb0 = 0; nb0 < NB; nb0 += 32) {
    const int nb = nb0 + (tid >> 3), j = tid & 7;
    float a = 0.f;
    if (nb < NB)
      for (int p = j; p < TT; p += 8) a += bk[p] == nb ? S[p] : 0.f;
    a += __shfl_xor(a, 1); a += __shfl_xor(a, 2); a += __shfl_xor(a, 4);
    if (nb < NB && j == 0) {
      float* o = drel + (long)nb * H + h;
      *o = accumulate ? *o + a : a;
    }
  }
}

__global__ void embedding_drop_fwd_kernel(const float* __restrict__ table, const int64_t* __restrict__ ids, float* __restrict__ out,
                                          long R, int d, const pq3d_dropout dr) {
  const DropState s = drop_init(dr, 0, d);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < R * d; i += (long)gridDim.x * blockDim.x) {
    const long r = i / d;
    const int c = (int)(i % d);
    out[i] = drop_keep(s, (uint32_t)r, (uint32_t)c) ? table[ids[r] * d + c] * s.scale : 0.f;
  }
}
__global__ void embedding_drop_bwd_kernel(const float* __restrict__ dout, const int64_t* __restrict__ ids, float* __restrict__ dtable,
                                          long R, int d, const pq3d_dropout dr) {
  const DropState s = drop_init(dr, 0, d);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < R * d; i += (long)gridDim.x * blockDim.x) {
    const long r = i / d;
    const int c = (int)(i % d);
    if (drop_keep(s, (uint32_t)r, (uint32_t)c)) unsafeAtomicAdd(&dtable[ids[r] * d + c], dout[i] * s.scale);
  }
}
inline unsigned grid_of(long total, long cap = 4096) {
  long g = (total + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int pq3d_t5_prep(const int64_t* labels, int64_t start_id, int64_t pad_id, const float* rel, const int64_t* buckets,
                            const uint8_t* enc_valid, int64_t* ids, float* bias, uint8_t* kpm, int32_t B, int32_t T, int32_t H,
                            int64_t N, void* stream) {
  PQ_DEVICE_GUARD(stream, bias);
  PQ_CHECK_ARG(labels && rel && buckets && ids && bias && B >= 0 && T >= 1 && H >= 1 && N >= 0 && (!kpm || enc_valid),
               "pq3d_t5_prep: bad args");
  if (B == 0) return 0;
  const long total = (long)B * H * T * T + (long)B * T + (kpm ? (long)B * N : 0);
  hipLaunchKernelGGL(t5_prep_kernel, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, labels, (long)start_id, (long)pad_id,
                     rel, buckets, enc_valid, ids, bias, kpm, B, T, H, (long)N);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_t5_bias_bwd(const float* dbias, const int64_t* buckets, float* drel, int32_t B, int32_t T, int32_t H,
                                int32_t NB, int32_t accumulate, void* stream) {
  PQ_DEVICE_GUARD(stream, dbias);
  PQ_CHECK_ARG(dbias && buckets && drel && B >= 0 && T >= 1 && H >= 1 && NB >= 1 && NB <= 256, "pq3d_t5_bias_bwd: bad args");
  const size_t lds = (size_t)T * T * 5 + 16;
  PQ_CHECK_ARG(lds <= 64 * 1024, "pq3d_t5_bias_bwd: T * T * 5 bytes of LDS must fit 64 KB (T <= 114)");
  hipLaunchKernelGGL(t5_bias_bwd_kernel, dim3(H), dim3(256), lds, (hipStream_t)stream, dbias, buckets, drel, B, T, H, NB, accumulate);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_embedding_drop_fwd(const float* table, const int64_t* ids, float* out, int64_t R, int32_t d,
                                       const pq3d_dropout* dr, void* stream) {
  PQ_DEVICE_GUARD(stream, table);
  PQ_CHECK_ARG(table && ids && out && dr && dr->seed && R >= 0 && d >= 1, "pq3d_embedding_drop_fwd: bad args");
  PQ_CHECK_DROP(*dr, R, d, "pq3d_embedding_drop_fwd");
  if (R == 0) return 0;
  hipLaunchKernelGGL(embedding_drop_fwd_kernel, dim3(grid_of(R * d)), dim3(256), 0, (hipStream_t)stream, table, ids, out, (long)R, d, *dr);
  PQ_LAUNCH_CHECK();
  return 0;
}
extern "C" int pq3d_embedding_drop_bwd_acc(const float* dout, const int64_t* ids, float* dtable, int64_t R, int32_t d,
                                           const pq3d_dropout* dr, void* stream) {
  PQ_DEVICE_GUARD(stream, dout);
  PQ_CHECK_ARG(dout && ids && dtable && dr && dr->seed && R >= 0 && d >= 1, "pq3d_embedding_drop_bwd_acc: bad args");
  PQ_CHECK_DROP(*dr, R, d, "pq3d_embedding_drop_bwd_acc");
  if (R == 0) return 0;
  hipLaunchKernelGGL(embedding_drop_bwd_kernel, dim3(grid_of(R * d)), dim3(256), 0, (hipStream_t)stream, dout, ids, dtable, (long)R, d, *dr);
  PQ_LAUNCH_CHECK();
  return 0;
}
