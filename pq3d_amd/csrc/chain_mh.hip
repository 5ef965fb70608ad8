// One launch for the row-local part of a MaskHeadSegLevel call (reference: mask_head.py:24-44 -- the class MLP of
// get_mlp_head, utils.py:17-26: Linear + ReLU + LayerNorm + Dropout + Linear; and the query side of every MaskPredictionLayer,
// mask_head.py:57-60):
//     h1   = relu(x W0^T + b0)                  qm_m = x Wq_m^T + bq_m   (m < Mm <= 3)
//     h2   = LN(h1)
//     cls  = h2 W4^T + b4, focus columns filled with -inf (the reference's logits[..., cols] = -inf)
// -- five dependent launches before (gemm_wk, add_ln_fwd, gemm_wk, fill_cols, gemm_wk: 29 us per call at config 4, which runs
// the head after every decoder block).  Same construction as chain_ffn.hip / chain_ca.hip: a group of 8 workgroups on one XCD
// owns NRT 32-row tiles through all steps, rows cross between members through that XCD's L2 (flags + sc1 loads).  Arithmetic is
// the separate kernels' own (split-bf16 products, k ascending, bias on the accumulator): tests/test_gpu_chain.py, bit for bit.
// The query -> segment mask-logit product itself (all segments of a scene per query) is not row-local and stays pq3d_gemm's.
#include <atomic>

#include "chain_common.h"
#include "gemm_common.h"

namespace {

constexpr int CN = 32, CCL = CN + 4;   // class-logit step: a member owns 32 classes
template <int NRT> constexpr size_t mh_lds() {
  const size_t s1 = proj_lds<NRT>(true);
  const size_t s3 = (size_t)NRT * 2 * TM * LDR * 2 + (size_t)2 * CN * LDR * 2 + (size_t)TM * CCL * 4;
  return s1 > s3 ? s1 : s3;
}
static_assert(mh_lds<2>() <= 160 * 1024, "LDS");

template <int NRT>
__global__ __launch_bounds__(CT) void chain_mh_fwd_kernel(const pq3d_chain_mh_desc d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ch_smem[];
  Ctx c;
  c.Ah = (bf16_t*)ch_smem;
  c.Al = c.Ah + TM * LDR;
  c.Bh = c.Al + TM * LDR;
  c.Bl = c.Bh + TN * LDR;
  c.Ct = (float*)(c.Bl + TN * LDR);
  c.tid = threadIdx.x; c.lane = c.tid & 63; c.wave = c.tid >> 6; c.li = c.lane & 15; c.lg = c.lane >> 4;
  c.wm = (c.wave >> 2) * 16; c.wn = (c.wave & 3) * 16;
  constexpr int GR = TM * NRT;
  const int id = (int)blockIdx.x, xcd = id & 7, q = id >> 3, slot = q >> 3, j = q & 7;
  const int grp = slot * 8 + xcd, m0 = grp * GR;
  const int R = d.R, Mm = d.Mm, Cn = d.C;
  if (m0 >= R) return;
  unsigned* const group = d.flags + (long)grp * G * 16;
  unsigned* const mine = group + j * 16;
  const unsigned v0 = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  // step 3's weight slab travels under steps 1 and 2: classes [32 j, + 32) x 256 k (rows beyond the last class: clamped, unused)
  const int n3 = j * CN;
  const bool on3 = n3 < Cn;
  RawA w4;
  if (on3) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ch = c.tid + i * CT;
      load8<false>(d.W4, (long)min(n3 + (ch >> 5), Cn - 1) * D + (ch & 31) * 8, w4.v[i]);
    }
  }
  // ---- 1. h1 = relu(x W0^T + b0), qm_m = x Wq_m^T + bq_m: member j < 2 (1 + Mm) owns product j / 2, half of its columns
  {
    const void* A[4] = {d.x, d.x, d.x, d.x};
    const float* W[4] = {d.W0, d.Wq[0], d.Wq[1], d.Wq[2]};
    const float* bias[4] = {d.b0, d.bq[0], d.bq[1], d.bq[2]};
    float* out[4] = {d.h1, d.qm[0], d.qm[1], d.qm[2]};
    RawB w1[2];
    proj_3x256<NRT, true, false, float>(c, ch_smem, j, 1 + Mm, m0, R, A, nullptr, W, bias, out, w1, false, 1);
  }
  handoff(c, mine, group, v0 + 1, d.err);
  // ---- 2. h2 = LN(h1) (no residual: add_ln_fwd's x = NULL): 32 NRT rows over 8 members x 4 NRT waves
  {
    const long row = m0 + 4 * NRT * j + c.wave;
    if (c.wave < 4 * NRT && row < R) {
      const long base = row * D + c.lane * 4;
      float ov[4], v[4], gm[4], bt[4], y[4];
      load4<true>(d.h1, base, ov);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = 0.f + ov[k];
      const RowStats st = row_stats4(v, d.eps);
      load4<false>(d.gamma, c.lane * 4, gm);
      load4<false>(d.beta, c.lane * 4, bt);
#pragma unroll
      for (int k = 0; k < 4; ++k) { y[k] = 0.f; y[k] += 1.f * ((v[k] - st.mean) * st.rstd * gm[k] + bt[k]); }
      if (c.lane == 0) { d.mean[row] = st.mean; d.rstd[row] = st.rstd; }
      *(float4*)(d.h2 + base) = make_float4(y[0], y[1], y[2], y[3]);
    }
  }
  handoff(c, mine, group, v0 + 2, d.err);
  // ---- 3. cls = h2 W4^T + b4 (+ the column fill): member j owns classes [32 j, + 32); waves 0..3 = 2 row halves x 2 class blocks
  if (on3) {   // uniform
    bf16_t* const Ap = (bf16_t*)ch_smem;                  // [NRT][2 planes][32][LDR]: h2, whole K
    bf16_t* const Bp = Ap + NRT * 2 * TM * LDR;           // [2 planes][32][LDR]: this member's classes
    float* const Ct = (float*)(Bp + 2 * CN * LDR);        // [32][CCL]
#pragma unroll
    for (int t = 0; t < NRT; ++t) {
      RawA ra;
      issue_a<true>(c, ra, d.h2, D, m0 + t * TM, R, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ch = c.tid + i * CT, o = (ch >> 5) * LDR + (ch & 31) * 8;
        u32x4 hi, lo;
        split_hi_lo(ra.v[i], hi, lo);
        *(u32x4*)&Ap[(t * 2) * TM * LDR + o] = hi;
        *(u32x4*)&Ap[(t * 2 + 1) * TM * LDR + o] = lo;
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ch = c.tid + i * CT, o = (ch >> 5) * LDR + (ch & 31) * 8;
      u32x4 hi, lo;
      split_hi_lo(w4.v[i], hi, lo);
      *(u32x4*)&Bp[o] = hi;
      *(u32x4*)&Bp[CN * LDR + o] = lo;
    }
    __syncthreads();
    const int wr0 = (c.wave >> 1) * 16, wc0 = (c.wave & 1) * 16;   // waves 0..3
    f32x4 acc[NRT];
#pragma unroll
    for (int t = 0; t < NRT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (c.wave < 4) {
#pragma unroll
      for (int ks = 0; ks < KC / 32; ++ks) {
        const int ob = (wc0 + c.li) * LDR + ks * 32 + c.lg * 8, oa = (wr0 + c.li) * LDR + ks * 32 + c.lg * 8;
        const u32x4 bh = *(const u32x4*)&Bp[ob], bl = *(const u32x4*)&Bp[CN * LDR + ob];
#pragma unroll
        for (int t = 0; t < NRT; ++t) {
          const u32x4 ah = *(const u32x4*)&Ap[(t * 2) * TM * LDR + oa], al = *(const u32x4*)&Ap[(t * 2 + 1) * TM * LDR + oa];
          Mma<bf16_t>::mma(acc[t], al, bh);
          Mma<bf16_t>::mma(acc[t], ah, bl);
          Mma<bf16_t>::mma(acc[t], ah, bh);
        }
      }
    }
    const int bc = n3 + wc0 + c.li;
    const float bcol = (c.wave < 4 && bc < Cn && d.b4) ? d.b4[bc] : 0.f;
#pragma unroll
    for (int t = 0; t < NRT; ++t) {
      if (t > 0) __syncthreads();
      if (c.wave < 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Ct[(wr0 + c.lg * 4 + r) * CCL + wc0 + c.li] = (acc[t][r] + bcol) * 1.f;
      }
      __syncthreads();
      const int orow = c.tid >> 4, row = m0 + t * TM + orow;   // 32 rows x 16 pairs of classes
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int cl = (c.tid & 15) * 2 + e, col = n3 + cl;
        if (row < R && col < Cn) {
          float v = Ct[orow * CCL + cl];
          if (d.colfill && d.colfill[col]) v = d.fill;
          d.cls[(long)row * Cn + col] = v;
        }
      }
    }
  }
  handoff(c, mine, group, v0 + 3, d.err);   // (the flags advance by the same amount in every member)
}

template <int NRT>
int launch_fwd(const pq3d_chain_mh_desc& d, int slots, hipStream_t s, std::atomic<unsigned>& done) {
  if (int e = pq3d_enable_big_lds(chain_mh_fwd_kernel<NRT>, (int)mh_lds<NRT>(), done)) { pq3d_set_error(hipGetErrorString((hipError_t)e)); return e; }
  hipLaunchKernelGGL(chain_mh_fwd_kernel<NRT>, dim3((unsigned)(8 * G * slots)), dim3(CT), mh_lds<NRT>(), s, d);
  return 0;
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Backward of the same part (autograd of the class MLP and of the query projections' input, as fused.py's mask_head() launches
// it without the chain: fill_cols, gemm_wk, add_ln_bwd, act_bwd, gemm_wk, gemm_wk -- 43 us per call at config 4):
//     dcl  = dc with the focus columns zeroed                      (operand of the queued weight gradient of W4)
//     dh2  = dcl W4                                                (single-bf16 product, K = C classes)
//     dh1  = LN'(h1; dh2)   (d gamma, d beta accumulated)          dpre = [h1 > 0] dh1  (bf16: operand of W0's weight gradient)
//     t    = dpre W0 + cur                                         out = sum_m dqm_m Wq_m + t
//     (optionally cur itself = sum_m dqc_m Wqc_m + dxr: the query-projection input gradient of the cross-attention that ran
//      backward just before -- one more launch, pq3d_gemm with kconcat + C2)
// Products as pq3d_gemm's transB form (weights [k][n] row-major, read through the transposing LDS load): member j owns 32 output
// columns, waves 0..3 = 2 row halves x 2 column blocks, one accumulator per row tile and product.
constexpr int S0N = 32, S0LD = S0N + 8;
template <int NRT> constexpr size_t mhb_lds() {
  const size_t s = (size_t)NRT * TM * LDR * 2 + (size_t)D * S0LD * 2 + (size_t)TM * 36 * 4;
  const size_t sl = (size_t)2 * 8 * D * 4;
  return s > sl ? s : sl;
}

PQ_DEV void ln_bwd_row(const float (&v)[4], const float (&dyr)[4], const float (&gam)[4], float mean, float rstd, float (&g)[4],
                       float (&dg)[4], float (&db)[4]) {   // norm.hip's add_ln_bwd at M = 1, no dropout
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
__global__ __launch_bounds__(CT) void chain_mh_bwd_kernel(const pq3d_chain_mh_bwd_desc d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ch_smem[];
  Ctx c;
  c.Ah = (bf16_t*)ch_smem; c.Al = c.Ah; c.Bh = c.Ah; c.Bl = c.Ah; c.Ct = (float*)ch_smem;
  c.tid = threadIdx.x; c.lane = c.tid & 63; c.wave = c.tid >> 6; c.li = c.lane & 15; c.lg = c.lane >> 4;
  c.wm = (c.wave >> 2) * 16; c.wn = (c.wave & 3) * 16;
  constexpr int GR = TM * NRT;
  const int id = (int)blockIdx.x, xcd = id & 7, q = id >> 3, slot = q >> 3, j = q & 7;
  const int grp = slot * 8 + xcd, m0 = grp * GR;
  const int R = d.R, Cn = d.C, Mm = d.Mm;
  if (m0 >= R) return;
  unsigned* const group = d.flags + (long)grp * G * 16;
  unsigned* const mine = group + j * 16;
  unsigned vs = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  float* const lnws = d.lnws + (long)grp * G * 512;   // [8 members][gamma 256 | beta 256]

  bf16_t* const Ap = (bf16_t*)ch_smem;                 // [NRT][32][LDR]: the left operand (bf16), whole K <= 256
  bf16_t* const Bp = Ap + NRT * TM * LDR;              // [256 k][S0LD]: columns [32 j, + 32) of a weight
  float* const Ct = (float*)(Bp + D * S0LD);           // [32][36]
  const int wr0 = (c.wave >> 1) * 16, wc0 = (c.wave & 1) * 16;   // waves 0..3
  // one weight slab [256 k][32 n] (k rows beyond kmax: zero): 1024 chunks of 8 floats, 2 per thread
  auto issue_wslab = [&](const float* W, int kmax, RawA& r) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ch = c.tid + i * CT, k = ch >> 2;
      if (k < kmax) load8<false>(W, (long)k * D + j * S0N + (ch & 3) * 8, r.v[i]);
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) r.v[i][e] = 0.f;
      }
    }
  };
  auto put_wslab = [&](const RawA& r) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ch = c.tid + i * CT;
      *(u32x4*)&Bp[(ch >> 2) * S0LD + (ch & 3) * 8] = pack_frag<bf16_t>(r.v[i]);
    }
  };
  auto mma_slab = [&](f32x4 (&acc)[NRT], int nks) {
    if (c.wave < 4) {
      for (int ks = 0; ks < nks; ++ks) {
        const u32x4 bh = km_frag(Bp, S0LD, wc0, ks, c.li, c.lg);
#pragma unroll
        for (int t = 0; t < NRT; ++t)
          Mma<bf16_t>::mma(acc[t], *(const u32x4*)&Ap[t * TM * LDR + (wr0 + c.li) * LDR + ks * 32 + c.lg * 8], bh);
      }
    }
  };
  RawA w4s, w0s;
  issue_wslab(d.W4, Cn, w4s);
  issue_wslab(d.W0, D, w0s);   // step 3's first weight travels under steps 1 and 2

  // ---- 1. dh2 = dcl W4 (dcl = dc, focus columns zeroed): K = C classes, zero-padded to a multiple of 32
  {
    const int nks = (Cn + 31) >> 5;
#pragma unroll
    for (int t = 0; t < NRT; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ch = c.tid + i * CT, lr = ch >> 5, cc = ch & 31, row = m0 + t * TM + lr, rr = min(row, R - 1);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int col = cc * 8 + e;
          float x = 0.f;
          if (col < Cn) {
            x = d.dc[(long)rr * Cn + col];
            if (d.colfill && d.colfill[col]) x = 0.f;
            if (d.dcl && (cc & 7) == j && row < R) d.dcl[(long)row * Cn + col] = x;
          }
          v[e] = x;
        }
        *(u32x4*)&Ap[t * TM * LDR + lr * LDR + cc * 8] = pack_frag<bf16_t>(v);
      }
    put_wslab(w4s);
    __syncthreads();
    f32x4 acc[NRT];
#pragma unroll
    for (int t = 0; t < NRT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    mma_slab(acc, nks);
#pragma unroll
    for (int t = 0; t < NRT; ++t) {
      if (t > 0) __syncthreads();
      if (c.wave < 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Ct[(wr0 + c.lg * 4 + r) * 36 + wc0 + c.li] = (acc[t][r] + 0.f) * 1.f;
      }
      __syncthreads();
      if (c.tid < 256) {                                 // 32 rows x 8 pieces of 4 columns
        const int orow = c.tid >> 3, col = (c.tid & 7) * 4, row = m0 + t * TM + orow;
        if (row < R) *(float4*)(d.dh2 + (long)row * D + j * S0N + col) = *(const float4*)&Ct[orow * 36 + col];
      }
    }
  }
  handoff(c, mine, group, ++vs, d.err);
  // ---- 2. dh1 = LN'(h1; dh2), dpre = [h1 > 0] dh1 (bf16): 32 NRT rows over 8 members x 4 NRT waves
  {
    const long lrow = m0 + 4 * NRT * j + c.wave;
    float dg[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f};
    if (c.wave < 4 * NRT && lrow < R) {
      const long lbase = lrow * D + c.lane * 4;
      float ov[4], v[4], dyr[4], gam[4], g[4];
      load4<false>(d.h1, lbase, ov);
      load4<true>(d.dh2, lbase, dyr);
      load4<false>(d.gamma, c.lane * 4, gam);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = 0.f + ov[k];
      ln_bwd_row(v, dyr, gam, d.mean[lrow], d.rstd[lrow], g, dg, db);
      *(u32x2*)((bf16_t*)d.dpre + lbase) = (u32x2){pack_bf2(ov[0] > 0.f ? g[0] : 0.f, ov[1] > 0.f ? g[1] : 0.f),
                                                   pack_bf2(ov[2] > 0.f ? g[2] : 0.f, ov[3] > 0.f ? g[3] : 0.f)};
    }
    ln_partials_store(c, (float*)ch_smem, dg, db, lnws + j * 512);
  }
  handoff(c, mine, group, ++vs, d.err);
  ln_partials_reduce(c, j, lnws, 512, 0, d.dgamma, d.dbeta);   // the LayerNorm's parameter gradients: 64 atomics per member
  // ---- 3. out = sum_m dqm_m Wq_m + (dpre W0 + cur)
  {
    f32x4 acc0[NRT], acc1[NRT], acc2[NRT];
#pragma unroll
    for (int t = 0; t < NRT; ++t) {
      acc0[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc2[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // The 1 + Mm + nq products run back to back on the same LDS tiles; the operands of product p + 1 are requested (registers)
    // before product p's MFMAs, so their latency hides behind them.  Order: dpre W0 (acc1), dqm_m Wq_m (acc2), then -- optional --
    // cur itself: the input gradient of the following layer application's cross-attention query projections, dqc_m Wqc_m (acc0:
    // cur = acc0 + dxr, gq = acc0; the member owns the same 32 columns of it, no hand-off needed).
    struct Opnd { RawA w; float af[NRT][2][8]; u32x4 ab[NRT][2]; };
    const int P = 1 + Mm + d.nq;
    auto issue_p = [&](int p, Opnd& o) {
      if (p > 0) issue_wslab(p <= Mm ? d.Wq[p - 1] : d.Wqc[p - 1 - Mm], D, o.w);   // (product 0's weight travels since the kernel's start)
      const bool f32 = p >= 1 && p <= Mm && d.dq_f32;
      const void* A = p == 0 ? (const void*)d.dpre : p <= Mm ? d.dq[p - 1] : d.dqc[p - 1 - Mm];
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0x7ffffff0, 0x00020000);
#pragma unroll
      for (int t = 0; t < NRT; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int ch = c.tid + i * CT;
          const long off = (long)min(m0 + t * TM + (ch >> 5), R - 1) * D + (ch & 31) * 8;
          if (f32) load8<false>((const float*)A, off, o.af[t][i]);
          else if (p == 0) o.ab[t][i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off * 2), 0, 16);   // written in this launch: sc1
          else o.ab[t][i] = *(const u32x4*)((const bf16_t*)A + off);
        }
    };
    auto put_p = [&](int p, const Opnd& o) {
      const bool f32 = p >= 1 && p <= Mm && d.dq_f32;
#pragma unroll
      for (int t = 0; t < NRT; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int ch = c.tid + i * CT;
          *(u32x4*)&Ap[t * TM * LDR + (ch >> 5) * LDR + (ch & 31) * 8] = f32 ? pack_frag<bf16_t>(o.af[t][i]) : o.ab[t][i];
        }
      put_wslab(p == 0 ? w0s : o.w);
    };
    Opnd op;
    issue_p(0, op);
    for (int p = 0; p < P; ++p) {
      if (p > 0) __syncthreads();   // the previous product's fragments are read
      put_p(p, op);
      __syncthreads();
      if (p + 1 < P) issue_p(p + 1, op);
      if (p == 0) mma_slab(acc1, D / 32);
      else if (p <= Mm) mma_slab(acc2, D / 32);
      else mma_slab(acc0, D / 32);
    }
#pragma unroll
    for (int t = 0; t < NRT; ++t) {
      const int orow = (c.tid & 255) >> 3, col = (c.tid & 7) * 4, row = m0 + t * TM + orow;
      const long o = (long)row * D + j * S0N + col;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f), cur4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (d.nq > 0) {   // uniform
        __syncthreads();
        if (c.wave < 4) {
#pragma unroll
          for (int r = 0; r < 4; ++r) Ct[(wr0 + c.lg * 4 + r) * 36 + wc0 + c.li] = (acc0[t][r] + 0.f) * 1.f;
        }
        __syncthreads();
        if (c.tid < 256 && row < R) {
          cur4 = *(const float4*)&Ct[orow * 36 + col];
          *(float4*)(d.gq + o) = cur4;
          const float4 a = *(const float4*)(d.dxr + o);
          cur4.x += a.x; cur4.y += a.y; cur4.z += a.z; cur4.w += a.w;
        }
      } else if (c.tid < 256 && row < R) cur4 = *(const float4*)(d.cur + o);
      __syncthreads();
      if (c.wave < 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Ct[(wr0 + c.lg * 4 + r) * 36 + wc0 + c.li] = (acc1[t][r] + 0.f) * 1.f;
      }
      __syncthreads();
      if (c.tid < 256 && row < R) {
        v = *(const float4*)&Ct[orow * 36 + col];
        v.x += cur4.x; v.y += cur4.y; v.z += cur4.z; v.w += cur4.w;
      }
      if (Mm > 0) {   // uniform
        __syncthreads();
        if (c.wave < 4) {
#pragma unroll
          for (int r = 0; r < 4; ++r) Ct[(wr0 + c.lg * 4 + r) * 36 + wc0 + c.li] = (acc2[t][r] + 0.f) * 1.f;
        }
        __syncthreads();
        if (c.tid < 256 && row < R) {
          const float4 w = *(const float4*)&Ct[orow * 36 + col];
          v.x = w.x + v.x; v.y = w.y + v.y; v.z = w.z + v.z; v.w = w.w + v.w;
        }
      }
      if (c.tid < 256 && row < R) *(float4*)(d.out + o) = v;
    }
  }
  handoff(c, mine, group, ++vs, d.err);   // (the flags advance by the same amount in every member)
}

template <int NRT>
int launch_bwd(const pq3d_chain_mh_bwd_desc& d, int slots, hipStream_t s, std::atomic<unsigned>& done) {
  if (int e = pq3d_enable_big_lds(chain_mh_bwd_kernel<NRT>, (int)mhb_lds<NRT>(), done)) { pq3d_set_error(hipGetErrorString((hipError_t)e)); return e; }
  hipLaunchKernelGGL(chain_mh_bwd_kernel<NRT>, dim3((unsigned)(8 * G * slots)), dim3(CT), mhb_lds<NRT>(), s, d);
  return 0;
}

}  // namespace

extern "C" int pq3d_chain_mh_fwd(const pq3d_chain_mh_desc* dp, void* stream) {
  PQ_DEVICE_GUARD(stream, dp ? dp->x : nullptr);
  PQ_CHECK_ARG(dp != nullptr, "pq3d_chain_mh_fwd: null descriptor");
  const pq3d_chain_mh_desc d = *dp;
  PQ_CHECK_ARG(d.R >= 1 && d.d == D && d.Mm >= 0 && d.Mm <= 3 && d.C >= 1 && d.C <= 256, "pq3d_chain_mh_fwd: d = 256, 0..3 memories, 1..256 classes");
  const int row_tiles = (d.R + TM - 1) / TM;
  const int nrt = chain_nrt(row_tiles);
  const int groups = (row_tiles + nrt - 1) / nrt, slots = (groups + 7) / 8;
  PQ_CHECK_ARG(slots * G <= 32, "pq3d_chain_mh_fwd: more than 2048 rows (the groups would not all be resident)");
  PQ_CHECK_ARG(d.x && d.W0 && d.b0 && d.gamma && d.beta && d.W4 && d.h1 && d.h2 && d.mean && d.rstd && d.cls && d.flags,
               "pq3d_chain_mh_fwd: null pointer");
  PQ_CHECK_ARG(((((uintptr_t)d.x) | ((uintptr_t)d.W0) | ((uintptr_t)d.gamma) | ((uintptr_t)d.beta) | ((uintptr_t)d.W4) |
                 ((uintptr_t)d.h1) | ((uintptr_t)d.h2)) & 15) == 0, "pq3d_chain_mh_fwd: operands must be 16-byte aligned");
  for (int m = 0; m < d.Mm; ++m)
    PQ_CHECK_ARG(d.Wq[m] && d.bq[m] && d.qm[m] && ((((uintptr_t)d.Wq[m]) | ((uintptr_t)d.qm[m])) & 15) == 0,
                 "pq3d_chain_mh_fwd: per-memory operands (non-null, 16-byte aligned)");
  static std::atomic<unsigned> done1{0}, done2{0};
  if (int e = nrt == 1 ? launch_fwd<1>(d, slots, (hipStream_t)stream, done1) : launch_fwd<2>(d, slots, (hipStream_t)stream, done2)) return e;
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_chain_mh_bwd(const pq3d_chain_mh_bwd_desc* dp, void* stream) {
  PQ_DEVICE_GUARD(stream, dp ? dp->dc : nullptr);
  PQ_CHECK_ARG(dp != nullptr, "pq3d_chain_mh_bwd: null descriptor");
  const pq3d_chain_mh_bwd_desc d = *dp;
  PQ_CHECK_ARG(d.R >= 1 && d.d == D && d.Mm >= 0 && d.Mm <= 3 && d.C >= 1 && d.C <= 256, "pq3d_chain_mh_bwd: d = 256, 0..3 memories, 1..256 classes");
  const int row_tiles = (d.R + TM - 1) / TM;
  const int nrt = chain_nrt(row_tiles);
  const int groups = (row_tiles + nrt - 1) / nrt, slots = (groups + 7) / 8;
  PQ_CHECK_ARG(slots * G <= 32, "pq3d_chain_mh_bwd: more than 2048 rows (the groups would not all be resident)");
  const void* ps[] = {d.dc, d.W4, d.h1, d.mean, d.rstd, d.gamma, d.dgamma, d.dbeta, d.dh2, d.dpre, d.W0, d.out, d.flags, d.lnws};
  for (const void* p : ps) PQ_CHECK_ARG(p != nullptr, "pq3d_chain_mh_bwd: null pointer");
  PQ_CHECK_ARG(d.nq >= 0 && d.nq <= 3, "pq3d_chain_mh_bwd: 0..3 query-projection terms");
  if (d.nq > 0) {
    PQ_CHECK_ARG(d.dxr && d.gq && ((((uintptr_t)d.dxr) | ((uintptr_t)d.gq)) & 15) == 0, "pq3d_chain_mh_bwd: dxr, gq (non-null, 16-byte aligned)");
    for (int m = 0; m < d.nq; ++m)
      PQ_CHECK_ARG(d.dqc[m] && d.Wqc[m] && ((((uintptr_t)d.dqc[m]) | ((uintptr_t)d.Wqc[m])) & 15) == 0, "pq3d_chain_mh_bwd: dqc / Wqc (non-null, aligned)");
  } else PQ_CHECK_ARG(d.cur && (((uintptr_t)d.cur) & 15) == 0, "pq3d_chain_mh_bwd: cur (non-null, 16-byte aligned)");
  const void* al[] = {d.W4, d.h1, d.gamma, d.dh2, d.dpre, d.W0, d.out};
  for (const void* p : al) PQ_CHECK_ARG((((uintptr_t)p) & 15) == 0, "pq3d_chain_mh_bwd: operands must be 16-byte aligned");
  PQ_CHECK_ARG(!d.colfill || d.dcl, "pq3d_chain_mh_bwd: a column fill needs the dcl output");
  for (int m = 0; m < d.Mm; ++m)
    PQ_CHECK_ARG(d.dq[m] && d.Wq[m] && ((((uintptr_t)d.dq[m]) | ((uintptr_t)d.Wq[m])) & 15) == 0, "pq3d_chain_mh_bwd: dq / Wq (non-null, aligned)");
  static std::atomic<unsigned> done1{0}, done2{0};
  if (int e = nrt == 1 ? launch_bwd<1>(d, slots, (hipStream_t)stream, done1) : launch_bwd<2>(d, slots, (hipStream_t)stream, done2)) return e;
  PQ_LAUNCH_CHECK();
  return 0;
}
