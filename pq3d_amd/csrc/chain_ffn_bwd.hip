// One launch for the row-local head of a decoder layer's backward (autograd of FFNLayer and of the self-attention sublayer's
// post-norm, query_encoder.py:384-388, 224-225):
//     g2  = LN2'(x2 + z; dx)              (= d z = d(residual) without dropout;  d gamma2, d beta2 accumulated)
//     dhp = [h > 0] (g2 W2)               (bf16: the operand of linear1's weight gradient and of the next product)
//     p_k = dhp[:, quarter k] W1[quarter k, :]          (K = F in 4 deterministic partial sums)
//     g1  = LN1'(x1s + f; g2 + p_0 + p_1 + p_2 + p_3)   (= d f = d x1s-residual;  d gamma1, d beta1 accumulated)
// -- four dependent launches before (add_ln_bwd, gemm_wk, gemm_wk with split-K atomics, add_ln_bwd: 35 us per layer at
// config 2).  Same construction as chain_ffn.hip: a group of 8 workgroups on one XCD owns NRT 32-row tiles through all steps,
// rows cross between members through that XCD's L2 (flags + sc1 loads).  Single-bf16 products (every backward product of the
// path), fp32 accumulation, k ascending; the input gradient of linear1 is summed in a fixed order here (the separate launches
// add their four k slices with atomics), the LayerNorm parameter gradients go to the arena with one atomic per column and
// workgroup as before.  The weight gradients themselves stay with the end-of-pass flush (they read g2, dhp, g1 from memory).
#include <atomic>

#include "chain_common.h"
#include "gemm_common.h"

namespace {

constexpr int B2K = 64, B2N = 256, B2LD = B2N + 8;     // step 2: [64 k][256 n] slab of W2 (k = model dim, n = hidden dim)
constexpr int B3K = 128, B3N = 128, B3LD = B3N + 8;    // step 3: [128 k][128 n] slab of W1 (k = hidden dim, n = model dim)
constexpr int A3LD = B3K + 8;
constexpr int S0N = 32, S0LD = S0N + 8;                // step 0: [256 k][32 n] slab of a query-projection weight
constexpr size_t bwd_lds(int nrt) {
  const size_t s2b = (size_t)B2K * B2LD * 2, s2c = (size_t)TM * (B2N + 4) * 4;
  const size_t s2 = (size_t)nrt * TM * LDR * 2 + (s2b > s2c ? s2b : s2c);
  const size_t s3b = (size_t)B3K * B3LD * 2, s3c = (size_t)TM * (B3N + 4) * 4;
  const size_t s3 = (size_t)nrt * TM * A3LD * 2 + (s3b > s3c ? s3b : s3c);
  const size_t sl = (size_t)2 * 8 * D * 4;   // LayerNorm parameter-gradient partials [2][8 waves][256]
  const size_t s0 = (size_t)nrt * TM * LDR * 2 + (size_t)D * S0LD * 2 + (size_t)TM * 36 * 4;   // step 0
  size_t m = s2 > s3 ? s2 : s3;
  m = m > s0 ? m : s0;
  return m > sl ? m : sl;
}

// one LayerNorm backward row (norm.hip's add_ln_bwd at M = 1, no dropout): returns g = d(x + o); accumulates this lane's
// parameter-gradient partials.  dyin: upstream gradient row (4 values of this lane)
PQ_DEV void ln_bwd_row(const float (&v)[4], const float (&dyr)[4], const float (&gam)[4], float mean, float rstd, float (&g)[4],
                       float (&dg)[4], float (&db)[4]) {
  float xh[4], dz[4];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float du = 1.f * dyr[j];
    xh[j] = (v[j] - mean) * rstd;
    dg[j] += du * xh[j];
    db[j] += du;
    dz[j] = du * gam[j];
    s1 += dz[j];
    s2 += dz[j] * xh[j];
  }
  s1 = wave_sum(s1) / (float)D;
  s2 = wave_sum(s2) / (float)D;
#pragma unroll
  for (int j = 0; j < 4; ++j) g[j] = rstd * (dz[j] - s1 - xh[j] * s2);
}
template <int NRT>
__global__ __launch_bounds__(CT) void chain_ffn_bwd_kernel(const pq3d_chain_ffn_bwd_desc d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ch_smem[];
  Ctx c;
  c.Ah = (bf16_t*)ch_smem; c.Al = c.Ah; c.Bh = c.Ah; c.Bl = c.Ah; c.Ct = (float*)ch_smem;
  c.tid = threadIdx.x; c.lane = c.tid & 63; c.wave = c.tid >> 6; c.li = c.lane & 15; c.lg = c.lane >> 4;
  c.wm = (c.wave >> 2) * 16; c.wn = (c.wave & 3) * 16;
  constexpr int GR = TM * NRT;
  const int id = (int)blockIdx.x, xcd = id & 7, q = id >> 3, slot = q >> 3, j = q & 7;
  const int grp = slot * 8 + xcd, m0 = grp * GR;
  const int R = d.R, F = d.F;
  if (m0 >= R) return;
  unsigned* const group = d.flags + (long)grp * G * 16;
  unsigned* const mine = group + j * 16;
  const unsigned v0 = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned vs = v0;   // hand-off targets: strictly increasing, the same sequence in every member
  float* const lnws = d.lnws + (long)grp * G * 1024;   // [8 members][2 LayerNorms][gamma 256 | beta 256]
  const long lrow = m0 + 4 * NRT * j + c.wave;
  const bool lnw = c.wave < 4 * NRT && lrow < R;
  const long lbase = lrow * D + c.lane * 4;
  const int wr = (c.wave >> 2) * 16;

  // step 2's first weight slabs travel under step 1 (weights do not depend on the chain): [64 k][256 n] of W2, 4 chunks / thread
  RawB ring[2];
  auto issue_w2 = [&](int l, RawB& r) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ch = c.tid + i * CT;
      load8<false>(d.W2, (long)(l * B2K + (ch >> 5)) * F + j * B2N + (ch & 31) * 8, r.v[i]);
    }
  };
  issue_w2(0, ring[0]); issue_w2(1, ring[1]);

  // ---- 0. (optional) the upstream gradient itself: dx = sum_m dq_m Wq_m + dxr, the input gradient of the cross-attention query
  // projections of the layer application that ran backward just before this one (pq3d_gemm: transB, kconcat = nq, C2 = the sum
  // without the addend).  Member j owns columns [32 j, + 32); waves 0..3 = 2 row halves x 2 column blocks, one accumulator
  // per row tile over all nq x 256 reduction steps (the separate launch's order).
  if (d.nq > 0) {   // uniform
    bf16_t* const Ap = (bf16_t*)ch_smem;                 // [NRT][32][LDR]: dq_m (bf16)
    bf16_t* const Bp = Ap + NRT * TM * LDR;              // [256 k][S0LD]: Wq_m[:, 32 j .. + 32)
    float* const Ct = (float*)(Bp + D * S0LD);           // [32][36]
    const int wr0 = (c.wave >> 1) * 16, wc0 = (c.wave & 1) * 16;   // waves 0..3
    f32x4 acc[NRT];
#pragma unroll
    for (int t = 0; t < NRT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int m = 0; m < d.nq; ++m) {
      if (m > 0) __syncthreads();
      const bf16_t* A = (const bf16_t*)d.dq[m];
#pragma unroll
      for (int t = 0; t < NRT; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int ch = c.tid + i * CT;
          *(u32x4*)&Ap[t * TM * LDR + (ch >> 5) * LDR + (ch & 31) * 8] =
              *(const u32x4*)(A + (long)min(m0 + t * TM + (ch >> 5), R - 1) * D + (ch & 31) * 8);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i) {                      // 256 k x 4 chunks of 8 columns
        const int ch = c.tid + i * CT;
        float v[8];
        load8<false>(d.Wq[m], (long)(ch >> 2) * D + j * S0N + (ch & 3) * 8, v);
        *(u32x4*)&Bp[(ch >> 2) * S0LD + (ch & 3) * 8] = pack_frag<bf16_t>(v);
      }
      __syncthreads();
      if (c.wave < 4) {
#pragma unroll
        for (int ks = 0; ks < D / 32; ++ks) {
          const u32x4 bh = km_frag(Bp, S0LD, wc0, ks, c.li, c.lg);
#pragma unroll
          for (int t = 0; t < NRT; ++t)
            Mma<bf16_t>::mma(acc[t], *(const u32x4*)&Ap[t * TM * LDR + (wr0 + c.li) * LDR + ks * 32 + c.lg * 8], bh);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < NRT; ++t) {
      __syncthreads();
      if (c.wave < 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Ct[(wr0 + c.lg * 4 + r) * 36 + wc0 + c.li] = (acc[t][r] + 0.f) * 1.f;
      }
      __syncthreads();
      if (c.tid < 256) {                                 // 32 rows x 8 pieces of 4 columns
        const int orow = c.tid >> 3, col = (c.tid & 7) * 4, row = m0 + t * TM + orow;
        if (row < R) {
          float4 v = *(const float4*)&Ct[orow * 36 + col];
          const long o = (long)row * D + j * S0N + col;
          *(float4*)(d.gq + o) = v;
          const float4 a = *(const float4*)(d.dxr + o);
          v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
          *(float4*)(d.dxo + o) = v;
        }
      }
    }
    handoff(c, mine, group, ++vs, d.err);
  }
  // ---- 1. g2 = LN2'(x2 + z; dx)
  {
    float dg[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f};
    if (lnw) {
      float xv[4], ov[4], v[4], dyr[4], gam[4], g[4];
      load4<false>(d.x2, lbase, xv);
      load4<false>(d.z, lbase, ov);
      if (d.nq > 0) load4<true>(d.dxo, lbase, dyr);
      else load4<false>(d.dx, lbase, dyr);
      load4<false>(d.g2, c.lane * 4, gam);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = xv[k] + ov[k];
      ln_bwd_row(v, dyr, gam, d.mean2[lrow], d.rstd2[lrow], g, dg, db);
      *(float4*)(d.dy + lbase) = make_float4(g[0], g[1], g[2], g[3]);
    }
    ln_partials_store(c, (float*)ch_smem, dg, db, lnws + j * 1024);
  }
  handoff(c, mine, group, ++vs, d.err);
  ln_partials_reduce(c, j, lnws, 1024, 0, d.dg2, d.db2);   // LayerNorm 2's parameter gradients: 64 atomics per member
  // ---- 2. dhp = [h > 0] (g2 W2): member j owns hidden columns [256 j, + 256); wave = 16 rows x 64 columns per row tile
  {
    bf16_t* const Ap = (bf16_t*)ch_smem;                 // [NRT][32][LDR]: g2 as bf16, whole K = 256
    bf16_t* const Bp = Ap + NRT * TM * LDR;              // [64 k][B2LD]: one k slab of W2
    float* const Ct = (float*)Bp;                        // [32][B2N + 4], over the slab once it is dead
    const int wc = (c.wave & 3) * 64;
#pragma unroll
    for (int t = 0; t < NRT; ++t) {
      RawA ra;
      issue_a<true>(c, ra, d.dy, D, m0 + t * TM, R, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ch = c.tid + i * CT;
        *(u32x4*)&Ap[t * TM * LDR + (ch >> 5) * LDR + (ch & 31) * 8] = pack_frag<bf16_t>(ra.v[i]);
      }
    }
    f32x4 acc[NRT][4];
#pragma unroll
    for (int t = 0; t < NRT; ++t)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) acc[t][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      if (l > 0) __syncthreads();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ch = c.tid + i * CT;
        *(u32x4*)&Bp[(ch >> 5) * B2LD + (ch & 31) * 8] = pack_frag<bf16_t>(ring[l & 1].v[i]);
      }
      __syncthreads();
      if (l + 2 < 4) issue_w2(l + 2, ring[l & 1]);
#pragma unroll
      for (int ks = 0; ks < B2K / 32; ++ks) {
        u32x4 bh[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) bh[nb] = km_frag(Bp, B2LD, wc + nb * 16, ks, c.li, c.lg);
#pragma unroll
        for (int t = 0; t < NRT; ++t) {
          const u32x4 ah = *(const u32x4*)&Ap[t * TM * LDR + (wr + c.li) * LDR + l * B2K + ks * 32 + c.lg * 8];
#pragma unroll
          for (int nb = 0; nb < 4; ++nb) Mma<bf16_t>::mma(acc[t][nb], ah, bh[nb]);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < NRT; ++t) {
      __syncthreads();
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) Ct[(wr + c.lg * 4 + r) * (B2N + 4) + wc + nb * 16 + c.li] = (acc[t][nb][r] + 0.f) * 1.f;
      __syncthreads();
      const int orow = c.tid >> 4, row = m0 + t * TM + orow;
      if (row < R) {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const int col = qd * 64 + (c.tid & 15) * 4;
          const float4 v = *(const float4*)&Ct[orow * (B2N + 4) + col];
          const float4 a = *(const float4*)(d.h + (long)row * F + j * B2N + col);
          *(u32x2*)((bf16_t*)d.dhp + (long)row * F + j * B2N + col) =
              (u32x2){pack_bf2(a.x > 0.f ? v.x : 0.f, a.y > 0.f ? v.y : 0.f), pack_bf2(a.z > 0.f ? v.z : 0.f, a.w > 0.f ? v.w : 0.f)};
        }
      }
    }
  }
  // step 3's first weight slabs (the ring is free): [128 k][128 n] of W1, 4 chunks / thread
  const int kq = j >> 1, nh = (j & 1) * B3N, Fq = F / 4;
  auto issue_w1 = [&](int l, RawB& r) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ch = c.tid + i * CT;
      load8<false>(d.W1, (long)(kq * Fq + l * B3K + (ch >> 4)) * D + nh + (ch & 15) * 8, r.v[i]);
    }
  };
  issue_w1(0, ring[0]); issue_w1(1, ring[1]);
  handoff(c, mine, group, ++vs, d.err);
  // ---- 3. p_k = dhp[:, quarter k] W1[quarter k, :]: member j owns quarter j / 2 and model columns [128 (j & 1), + 128)
  {
    bf16_t* const Ap = (bf16_t*)ch_smem;                 // [NRT][32][A3LD]: one k slab of dhp
    bf16_t* const Bp = Ap + NRT * TM * A3LD;             // [128 k][B3LD]: one k slab of W1
    float* const Ct = (float*)Bp;
    const int wc = (c.wave & 3) * 32;
    u32x4 av[NRT];
    auto issue_h = [&](int l) {                          // [32 NRT rows][128 k] of dhp (bf16): 512 NRT chunks of 16 bytes
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)d.dhp, 0, 0x7ffffff0, 0x00020000);
#pragma unroll
      for (int t = 0; t < NRT; ++t)
        av[t] = __builtin_amdgcn_raw_buffer_load_b128(
            rs, (int)(((long)min(m0 + t * TM + (c.tid >> 4), R - 1) * F + kq * Fq + l * B3K + (c.tid & 15) * 8) * 2), 0, 16);
    };
    issue_h(0);
    f32x4 acc[NRT][2];
#pragma unroll
    for (int t = 0; t < NRT; ++t) { acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      if (l > 0) __syncthreads();
#pragma unroll
      for (int t = 0; t < NRT; ++t) *(u32x4*)&Ap[t * TM * A3LD + (c.tid >> 4) * A3LD + (c.tid & 15) * 8] = av[t];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ch = c.tid + i * CT;
        *(u32x4*)&Bp[(ch >> 4) * B3LD + (ch & 15) * 8] = pack_frag<bf16_t>(ring[l & 1].v[i]);
      }
      __syncthreads();
      if (l + 1 < 4) issue_h(l + 1);
      if (l + 2 < 4) issue_w1(l + 2, ring[l & 1]);
#pragma unroll
      for (int ks = 0; ks < B3K / 32; ++ks) {
        u32x4 bh[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) bh[nb] = km_frag(Bp, B3LD, wc + nb * 16, ks, c.li, c.lg);
#pragma unroll
        for (int t = 0; t < NRT; ++t) {
          const u32x4 ah = *(const u32x4*)&Ap[t * TM * A3LD + (wr + c.li) * A3LD + ks * 32 + c.lg * 8];
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) Mma<bf16_t>::mma(acc[t][nb], ah, bh[nb]);
        }
      }
    }
    float* const pk = d.part + (long)kq * R * D;
#pragma unroll
    for (int t = 0; t < NRT; ++t) {
      __syncthreads();
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) Ct[(wr + c.lg * 4 + r) * (B3N + 4) + wc + nb * 16 + c.li] = acc[t][nb][r];
      __syncthreads();
      const int orow = c.tid >> 4, row = m0 + t * TM + orow;
#pragma unroll
      for (int qd = 0; qd < 2; ++qd) {
        const int col = qd * 64 + (c.tid & 15) * 4;
        if (row < R) *(float4*)(pk + (long)row * D + nh + col) = *(const float4*)&Ct[orow * (B3N + 4) + col];
      }
    }
  }
  handoff(c, mine, group, ++vs, d.err);
  // ---- 4. g1 = LN1'(x1s + f; g2 + p_0 + p_1 + p_2 + p_3)
  {
    float dg[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f};
    if (lnw) {
      float xv[4], ov[4], v[4], dyr[4], gam[4], g[4];
      load4<false>(d.x1s, lbase, xv);
      load4<false>(d.f, lbase, ov);
      load4<true>(d.dy, lbase, dyr);
      for (int p = 0; p < 4; ++p) {
        float t[4];
        load4<true>(d.part + (long)p * R * D, lbase, t);
#pragma unroll
        for (int k = 0; k < 4; ++k) dyr[k] += t[k];
      }
      load4<false>(d.g1, c.lane * 4, gam);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = xv[k] + ov[k];
      ln_bwd_row(v, dyr, gam, d.mean1[lrow], d.rstd1[lrow], g, dg, db);
      *(float4*)(d.df + lbase) = make_float4(g[0], g[1], g[2], g[3]);
    }
    ln_partials_store(c, (float*)ch_smem, dg, db, lnws + j * 1024 + 512);
  }
  handoff(c, mine, group, ++vs, d.err);
  ln_partials_reduce(c, j, lnws, 1024, 512, d.dg1, d.db1);
}

}  // namespace

extern "C" int pq3d_chain_ffn_bwd(const pq3d_chain_ffn_bwd_desc* dp, void* stream) {
  PQ_DEVICE_GUARD(stream, dp ? dp->x2 : nullptr);
  PQ_CHECK_ARG(dp != nullptr, "pq3d_chain_ffn_bwd: null descriptor");
  const pq3d_chain_ffn_bwd_desc d = *dp;
  PQ_CHECK_ARG(d.R >= 1 && d.d == D && d.F == 2048, "pq3d_chain_ffn_bwd: d = 256, F = 2048");
  const int row_tiles = (d.R + TM - 1) / TM;
  const int nrt = chain_nrt(row_tiles);
  const int groups = (row_tiles + nrt - 1) / nrt, slots = (groups + 7) / 8;
  PQ_CHECK_ARG(slots * G <= 32, "pq3d_chain_ffn_bwd: more than 2048 rows (the groups would not all be resident)");
  const void* ps[] = {d.nq > 0 ? (const void*)d.dxo : (const void*)d.dx, d.x2, d.z, d.g2, d.mean2, d.rstd2, d.dg2, d.db2, d.dy, d.W2, d.h, d.dhp, d.W1, d.part, d.x1s, d.f, d.g1, d.mean1,
                      d.rstd1, d.dg1, d.db1, d.df, d.flags, d.lnws};
  for (const void* p : ps) PQ_CHECK_ARG(p != nullptr, "pq3d_chain_ffn_bwd: null pointer");
  const void* al[] = {d.nq > 0 ? (const void*)d.dxo : (const void*)d.dx, d.x2, d.z, d.g2, d.dy, d.W2, d.h, d.dhp, d.W1, d.part, d.x1s, d.f, d.g1, d.df};
  for (const void* p : al) PQ_CHECK_ARG((((uintptr_t)p) & 15) == 0, "pq3d_chain_ffn_bwd: operands must be 16-byte aligned");
  PQ_CHECK_ARG((long)d.R * d.F * 4 < 0x7ffffff0L, "pq3d_chain_ffn_bwd: hidden activations too large");
  PQ_CHECK_ARG(d.nq >= 0 && d.nq <= 3, "pq3d_chain_ffn_bwd: 0..3 query-projection terms");
  if (d.nq > 0) {
    PQ_CHECK_ARG(d.dxr && d.gq && d.dxo && ((((uintptr_t)d.dxr) | ((uintptr_t)d.gq) | ((uintptr_t)d.dxo)) & 15) == 0,
                 "pq3d_chain_ffn_bwd: step 0 needs dxr, gq, dxo (16-byte aligned)");
    for (int m = 0; m < d.nq; ++m)
      PQ_CHECK_ARG(d.dq[m] && d.Wq[m] && ((((uintptr_t)d.dq[m]) | ((uintptr_t)d.Wq[m])) & 15) == 0, "pq3d_chain_ffn_bwd: dq / Wq (non-null, aligned)");
  }
  static std::atomic<unsigned> done1{0}, done2{0};
  const dim3 grid((unsigned)(8 * G * slots));
  if (nrt == 1) {
    if (int e = pq3d_enable_big_lds(chain_ffn_bwd_kernel<1>, (int)bwd_lds(1), done1)) { pq3d_set_error(hipGetErrorString((hipError_t)e)); return e; }
    hipLaunchKernelGGL(chain_ffn_bwd_kernel<1>, grid, dim3(CT), bwd_lds(1), (hipStream_t)stream, d);
  } else {
    if (int e = pq3d_enable_big_lds(chain_ffn_bwd_kernel<2>, (int)bwd_lds(2), done2)) { pq3d_set_error(hipGetErrorString((hipError_t)e)); return e; }
    hipLaunchKernelGGL(chain_ffn_bwd_kernel<2>, grid, dim3(CT), bwd_lds(2), (hipStream_t)stream, d);
  }
  PQ_LAUNCH_CHECK();
  return 0;
}
