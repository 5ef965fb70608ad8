// One launch for the row-local steps between a decoder layer's cross-attention and its self-attention (reference:
// CrossAttentionLayer out_proj + post-norm merged over the M scene memories, query_encoder.py:145-152, 288-307; then the
// self-attention's q / k / v projections, :224, transformers.py:190-193):
//     op_m = o_m Wo_m^T + bo_m   (m < M, bf16 attention outputs, single-bf16 product)
//     x1   = sum_m c_m LN_m(x + op_m)                      (c_m = coef[m][scene] or 1 / M)
//     q = (x1 + qpos) Wq^T + bq,  k = (x1 + qpos) Wk^T + bk,  v = x1 Wv^T + bv     (split-bf16 products)
// -- three dependent launches before (gemm_wk, add_ln_fwd, gemm_wk: 22 us per layer at config 2).  Same construction as
// chain_ffn.hip: a group of 8 workgroups on one XCD owns NRT 32-row tiles through all three steps, rows cross between members
// through that XCD's L2 (flags + sc1 loads).  Arithmetic is the three kernels' own (tests/test_gpu_chain.py: bit for bit).
#include <atomic>

#include "chain_common.h"

namespace {



// OX3: the attention outputs arrive in fp32 and the out-projections are split-bf16 products (compute mode 'bf16x3'; the LDS of
// step 3 covers it)
template <int NRT, bool OX3>
__global__ __launch_bounds__(CT) void chain_ca_fwd_kernel(const pq3d_chain_ca_desc d) {
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
  const int R = d.R, M = d.M;
  if (m0 >= R) return;
  unsigned* const group = d.flags + (long)grp * G * 16;
  unsigned* const mine = group + j * 16;
  const unsigned v0 = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  RawB wq[2];
  // ---- 1. out-projections: member j < 2 M owns memory j / 2, columns [128 (j & 1), + 128) (single-bf16 product)
  {
    const void* A[3] = {d.o[0], d.o[1], d.o[2]};
    RawB w1[2];
    proj_issue_w(c, j, M, d.Wo, w1);
    proj_issue_w(c, j, 3, d.Wqkv, wq);   // step 3's weights travel under steps 1 and 2
    proj_3x256<NRT, OX3, false, float>(c, ch_smem, j, M, m0, R, A, nullptr, d.Wo, d.bo, d.op, w1, true);
  }
  handoff(c, mine, group, v0 + 1, d.err);
  // ---- 2. x1 = sum_m c_m LN_m(x + op_m): 32 NRT rows over 8 members x 4 NRT waves
  {
    const long row = m0 + 4 * NRT * j + c.wave;
    if (c.wave < 4 * NRT && row < R) {
      const long base = row * D + c.lane * 4;
      const long nscene = d.coef ? R / d.rows_per_scene : 1;
      float xr[4], y[4] = {0.f, 0.f, 0.f, 0.f};
      load4<false>(d.x, base, xr);
      for (int m = 0; m < M; ++m) {
        float ov[4], v[4], gm[4], bt[4];
        load4<true>(d.op[m], base, ov);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = xr[k] + ov[k];
        const RowStats st = row_stats4(v, d.eps);
        const float w = d.coef ? d.coef[m * nscene + row / d.rows_per_scene] : 1.f / (float)M;
        load4<false>(d.gamma[m], c.lane * 4, gm);
        load4<false>(d.beta[m], c.lane * 4, bt);
#pragma unroll
        for (int k = 0; k < 4; ++k) y[k] += w * ((v[k] - st.mean) * st.rstd * gm[k] + bt[k]);
        if (c.lane == 0) { d.mean[(long)m * R + row] = st.mean; d.rstd[(long)m * R + row] = st.rstd; }
      }
      *(float4*)(d.x1 + base) = make_float4(y[0], y[1], y[2], y[3]);
    }
  }
  handoff(c, mine, group, v0 + 2, d.err);
  // ---- 3. q / k / v projections (split-bf16; q and k read x1 + qpos): member j < 6 owns projection j / 2, half of its columns
  {
    const void* A[3] = {d.x1, d.x1, d.x1};
    const float* A2[3] = {d.qpos, d.qpos, nullptr};
    proj_3x256<NRT, true, true, float>(c, ch_smem, j, 3, m0, R, A, A2, d.Wqkv, d.bqkv, d.qkv, wq, true);
  }
  // (no hand-off behind the last step, but the flags still advance by the same amount in every member)
  handoff(c, mine, group, v0 + 3, d.err);
}

}  // namespace

extern "C" int pq3d_chain_ca_fwd(const pq3d_chain_ca_desc* dp, void* stream) {
  PQ_DEVICE_GUARD(stream, dp ? dp->x : nullptr);
  PQ_CHECK_ARG(dp != nullptr, "pq3d_chain_ca_fwd: null descriptor");
  const pq3d_chain_ca_desc d = *dp;
  PQ_CHECK_ARG(d.R >= 1 && d.d == D && d.M >= 1 && d.M <= 3, "pq3d_chain_ca_fwd: d = 256, 1..3 memories");
  PQ_CHECK_ARG(!d.coef || (d.rows_per_scene >= 1 && d.R % d.rows_per_scene == 0), "pq3d_chain_ca_fwd: rows_per_scene must divide R");
  const int row_tiles = (d.R + TM - 1) / TM;
  const int nrt = chain_nrt(row_tiles);
  const int groups = (row_tiles + nrt - 1) / nrt, slots = (groups + 7) / 8;
  PQ_CHECK_ARG(slots * G <= 32, "pq3d_chain_ca_fwd: more than 2048 rows (the groups would not all be resident)");
  PQ_CHECK_ARG(d.x && d.x1 && d.mean && d.rstd && d.qpos && d.flags, "pq3d_chain_ca_fwd: null pointer");
  for (int m = 0; m < d.M; ++m)
    PQ_CHECK_ARG(d.o[m] && d.Wo[m] && d.bo[m] && d.gamma[m] && d.beta[m] && d.op[m] &&
                 ((((uintptr_t)d.o[m]) | ((uintptr_t)d.Wo[m]) | ((uintptr_t)d.gamma[m]) | ((uintptr_t)d.beta[m]) | ((uintptr_t)d.op[m])) & 15) == 0,
                 "pq3d_chain_ca_fwd: per-memory operands (non-null, 16-byte aligned)");
  for (int g = 0; g < 3; ++g)
    PQ_CHECK_ARG(d.Wqkv[g] && d.bqkv[g] && d.qkv[g] && ((((uintptr_t)d.Wqkv[g]) | ((uintptr_t)d.qkv[g])) & 15) == 0,
                 "pq3d_chain_ca_fwd: q / k / v operands (non-null, 16-byte aligned)");
  PQ_CHECK_ARG(((((uintptr_t)d.x) | ((uintptr_t)d.x1) | ((uintptr_t)d.qpos)) & 15) == 0, "pq3d_chain_ca_fwd: operands must be 16-byte aligned");
  static std::atomic<unsigned> done[4] = {{0}, {0}, {0}, {0}};
  const dim3 grid((unsigned)(8 * G * slots));
  auto go = [&](auto kern, size_t lds, std::atomic<unsigned>& dn) -> int {
    if (int e = pq3d_enable_big_lds(kern, (int)lds, dn)) { pq3d_set_error(hipGetErrorString((hipError_t)e)); return e; }
    hipLaunchKernelGGL(kern, grid, dim3(CT), lds, (hipStream_t)stream, d);
    return 0;
  };
  int e = 0;
  if (nrt == 1) e = d.o_f32 ? go(chain_ca_fwd_kernel<1, true>, proj_lds<1>(true), done[0]) : go(chain_ca_fwd_kernel<1, false>, proj_lds<1>(true), done[1]);
  else e = d.o_f32 ? go(chain_ca_fwd_kernel<2, true>, proj_lds<2>(true), done[2]) : go(chain_ca_fwd_kernel<2, false>, proj_lds<2>(true), done[3]);
  if (e) return e;
  PQ_LAUNCH_CHECK();
  return 0;
}
