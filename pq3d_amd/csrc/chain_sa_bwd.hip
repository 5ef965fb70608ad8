// One launch for the row-local steps between a decoder layer's self-attention backward and its cross-attention backward
// (autograd of the self-attention's q / k / v projections, of the merged cross-attention post-norm and of the cross-attention
// out-projections; transformers.py:190-193, query_encoder.py:145-152, 304-305):
//     g_q = dq Wq,  g_k = dk Wk,  g_v = dv Wv + r          (r: the residual-branch gradient of the self-attention post-norm)
//     for m < M: dop_m = LN_m'(x + op_m; c_m ((g_q + g_k) + g_v)),   dxr = sum_m dop_m;   d gamma_m, d beta_m accumulated
//     do_m = dop_m Wo_m                                      (bf16: the attention backward's d O)
// -- three dependent launches before (gemm_wk, add_ln_bwd_merged, gemm_wk: 25 us per layer at config 2).  Construction and
// hand-offs as in chain_ffn.hip; arithmetic of the three kernels (single-bf16 products, k ascending; the merged LayerNorm
// backward's per-row order), so every output equals the separate launches' bit for bit at M = 3.
#include <atomic>

#include "chain_common.h"

namespace {

template <int NRT>
__global__ __launch_bounds__(CT) void chain_sa_bwd_kernel(const pq3d_chain_sa_bwd_desc d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ch_smem[];
  Ctx c;
  c.Ah = (bf16_t*)ch_smem; c.Al = c.Ah; c.Bh = c.Ah; c.Bl = c.Ah; c.Ct = (float*)ch_smem;
  c.tid = threadIdx.x; c.lane = c.tid & 63; c.wave = c.tid >> 6; c.li = c.lane & 15; c.lg = c.lane >> 4;
  c.wm = (c.wave >> 2) * 16; c.wn = (c.wave & 3) * 16;
  constexpr int GR = TM * NRT;
  const int id = (int)blockIdx.x, xcd = id & 7, q = id >> 3, slot = q >> 3, j = q & 7;
  const int grp = slot * 8 + xcd, m0 = grp * GR;
  const int R = d.R, M = d.M;
  if (m0 >= R) return;
  unsigned* const group = d.flags + (long)grp * G * 16;
  unsigned* const mine = group + j * 16;
  const unsigned v0 = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  float* const lnws = d.lnws + (long)grp * G * 1536;   // [8 members][3 branches][gamma 256 | beta 256]

  RawB w1[2], w3[2];
  tproj_issue_w(c, j, 3, d.Wl, w1);
  tproj_issue_w(c, j, M, d.Wo, w3);   // step 3's weights travel under steps 1 and 2
  // ---- 1. input gradients of the q / k / v projections (+ the residual-branch gradient on the v part)
  {
    const float* aux[3] = {nullptr, nullptr, d.aux2};
    tproj_3x256<NRT, false, float>(c, ch_smem, j, 3, m0, R, d.dqkv, aux, d.g3, w1);
  }
  handoff(c, mine, group, v0 + 1, d.err);
  // ---- 2. merged LayerNorm backward over the M branches (norm.hip's add_ln_bwd_merged, no dropout)
  {
    const long row = m0 + 4 * NRT * j + c.wave;
    const bool on = c.wave < 4 * NRT && row < R;
    const long base = row * D + c.lane * 4;
    const long nscene = d.coef ? R / d.rows_per_scene : 1;
    float dg[3][4], db[3][4];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int k = 0; k < 4; ++k) { dg[m][k] = 0.f; db[m][k] = 0.f; }
    if (on) {
      float xv[4], dyr[4], gsum[4] = {0.f, 0.f, 0.f, 0.f}, t[4];
      load4<false>(d.x, base, xv);
      load4<true>(d.g3[0], base, dyr);
      load4<true>(d.g3[1], base, t);
#pragma unroll
      for (int k = 0; k < 4; ++k) dyr[k] += t[k];
      load4<true>(d.g3[2], base, t);
#pragma unroll
      for (int k = 0; k < 4; ++k) dyr[k] += t[k];
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        if (m < M) {
          float ov[4], gam[4], xh[4], dz[4], go[4];
          load4<false>(d.op[m], base, ov);
          load4<false>(d.gamma[m], c.lane * 4, gam);
          const float mean = d.mean[(long)m * R + row], rstd = d.rstd[(long)m * R + row];
          const float w = d.coef ? d.coef[m * nscene + row / d.rows_per_scene] : 1.f / (float)M;
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float du = w * dyr[k];
            xh[k] = ((xv[k] + ov[k]) - mean) * rstd;
            dg[m][k] += du * xh[k];
            db[m][k] += du;
            dz[k] = du * gam[k];
            s1 += dz[k];
            s2 += dz[k] * xh[k];
          }
          s1 = wave_sum(s1) / (float)D;
          s2 = wave_sum(s2) / (float)D;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float g = rstd * (dz[k] - s1 - xh[k] * s2);
            gsum[k] += g;
            go[k] = g;
          }
          *(float4*)(d.dop[m] + base) = make_float4(go[0], go[1], go[2], go[3]);
        }
      }
      *(float4*)(d.dxr + base) = make_float4(gsum[0], gsum[1], gsum[2], gsum[3]);
    }
    // parameter gradients: the member's column sums of every branch into the group's scratch (reduced behind the hand-off)
#pragma unroll
    for (int m = 0; m < 3; ++m)
      if (m < M) ln_partials_store(c, (float*)ch_smem, dg[m], db[m], lnws + j * 1536 + m * 512);   // uniform
  }
  handoff(c, mine, group, v0 + 2, d.err);
  for (int m = 0; m < M; ++m) ln_partials_reduce(c, j, lnws, 1536, m * 512, d.dgamma[m], d.dbeta[m]);
  // ---- 3. d O of the cross-attention: do_m = dop_m Wo_m (bf16)
  {
    bf16_t* out[3] = {(bf16_t*)d.do_all[0], (bf16_t*)d.do_all[1], (bf16_t*)d.do_all[2]};
    const float* A[3] = {d.dop[0], d.dop[1], d.dop[2]};
    tproj_3x256<NRT, true, bf16_t>(c, ch_smem, j, M, m0, R, A, nullptr, out, w3);
  }
  handoff(c, mine, group, v0 + 3, d.err);   // (keeps the members' flag words in step)
}

}  // namespace

extern "C" int pq3d_chain_sa_bwd(const pq3d_chain_sa_bwd_desc* dp, void* stream) {
  PQ_DEVICE_GUARD(stream, dp ? dp->x : nullptr);
  PQ_CHECK_ARG(dp != nullptr, "pq3d_chain_sa_bwd: null descriptor");
  const pq3d_chain_sa_bwd_desc d = *dp;
  PQ_CHECK_ARG(d.R >= 1 && d.d == D && d.M >= 1 && d.M <= 3, "pq3d_chain_sa_bwd: d = 256, 1..3 memories");
  PQ_CHECK_ARG(!d.coef || (d.rows_per_scene >= 1 && d.R % d.rows_per_scene == 0), "pq3d_chain_sa_bwd: rows_per_scene must divide R");
  const int row_tiles = (d.R + TM - 1) / TM;
  const int nrt = chain_nrt(row_tiles);
  const int groups = (row_tiles + nrt - 1) / nrt, slots = (groups + 7) / 8;
  PQ_CHECK_ARG(slots * G <= 32, "pq3d_chain_sa_bwd: more than 2048 rows (the groups would not all be resident)");
  PQ_CHECK_ARG(d.x && d.aux2 && d.mean && d.rstd && d.dxr && d.flags && d.lnws && ((((uintptr_t)d.x) | ((uintptr_t)d.aux2) | ((uintptr_t)d.dxr)) & 15) == 0,
               "pq3d_chain_sa_bwd: null / unaligned pointer");
  for (int g = 0; g < 3; ++g)
    PQ_CHECK_ARG(d.dqkv[g] && d.Wl[g] && d.g3[g] && ((((uintptr_t)d.dqkv[g]) | ((uintptr_t)d.Wl[g]) | ((uintptr_t)d.g3[g])) & 15) == 0,
                 "pq3d_chain_sa_bwd: q / k / v operands (non-null, 16-byte aligned)");
  for (int m = 0; m < d.M; ++m)
    PQ_CHECK_ARG(d.op[m] && d.gamma[m] && d.dop[m] && d.dgamma[m] && d.dbeta[m] && d.Wo[m] && d.do_all[m] &&
                 ((((uintptr_t)d.op[m]) | ((uintptr_t)d.gamma[m]) | ((uintptr_t)d.dop[m]) | ((uintptr_t)d.Wo[m]) | ((uintptr_t)d.do_all[m])) & 15) == 0,
                 "pq3d_chain_sa_bwd: per-memory operands (non-null, 16-byte aligned)");
  static std::atomic<unsigned> done1{0}, done2{0};
  const dim3 grid((unsigned)(8 * G * slots));
  if (nrt == 1) {
    if (int e = pq3d_enable_big_lds(chain_sa_bwd_kernel<1>, (int)tproj_lds<1>(), done1)) { pq3d_set_error(hipGetErrorString((hipError_t)e)); return e; }
    hipLaunchKernelGGL(chain_sa_bwd_kernel<1>, grid, dim3(CT), tproj_lds<1>(), (hipStream_t)stream, d);
  } else {
    if (int e = pq3d_enable_big_lds(chain_sa_bwd_kernel<2>, (int)tproj_lds<2>(), done2)) { pq3d_set_error(hipGetErrorString((hipError_t)e)); return e; }
    hipLaunchKernelGGL(chain_sa_bwd_kernel<2>, grid, dim3(CT), tproj_lds<2>(), (hipStream_t)stream, d);
  }
  PQ_LAUNCH_CHECK();
  return 0;
}
