// One launch for the row-local tail of a decoder layer's forward (reference: QueryEncoderLayer's self-attention output
// projection + post-norm, then FFNLayer -- query_encoder.py:224-225, 384-388):
//     f  = o_s Wo^T + bo                x2 = LN1(x1s + f)
//     h  = relu(x2 W1^T + b1)           zp_k = h[:, k-th quarter] W2[:, k-th quarter]^T (+ b2 for k = 0)
//     z  = ((zp_0 + zp_1) + zp_2) + zp_3        x3 = LN2(x2 + z)
// Until round 5 these were five dependent launches (gemm_wk, add_ln_fwd, gemm_wk, gemm_wk, add_ln_fwd: 41.6 us per layer at
// config 2, 4.7 us of launch-to-launch floor each).  Every step is ROW-LOCAL: a 32-row tile never needs another tile's rows.
// So a group of 8 workgroups owns a row tile for all five steps, and the group sits on ONE XCD (workgroup id % 8 is the XCD):
// its members hand rows to each other through the XCD's L2 -- stores (write-through L1), s_waitcnt vmcnt(0), one flag word
// per member, L1-bypassing (sc1) loads on the consumer side -- at 1.9 us per hand-off (tools/probes/xcd_barrier_probe.hip:
// 0 stale reads in 1000 rounds x 256 workgroups; an agent-scope acquire fence instead of sc1 loads costs 9 us, a release
// with L2 write-back is only needed ACROSS XCDs).  No cross-XCD traffic, no atomics.
//
// Arithmetic is the five kernels' own, instruction for instruction: split-bf16 products (hi/lo planes, MFMA order lo*hi, hi*lo,
// hi*hi per 32-wide k-step, k ascending, one accumulator per output), bias added to the accumulator, ReLU, two-pass LayerNorm
// statistics with the same wave reductions, the four partial sums of linear2 added in index order -- tests/test_gpu_chain.py
// compares every output bit for bit with the five-launch path.
// Round 6, step 0 (optional, pq3d_chain_ffn_desc.sa_q): the self-attention CORE that used to be a launch of its own between
// chain_ca and this kernel (sa32::attn_sa_fwd_kernel: 64 workgroups at config 2) -- member j = head j forms o_s[tile rows,
// 32 j .. 32 j + 32) from the q / k / v the previous launch wrote, with attn_sa_body.h's own block loop (sa_fwd_block: the same
// instructions, hence the same bits); a tile that straddles scenes runs one (16-row block, scene) task per wave over the planes of
// up to two scenes at a time.  One more hand-off (o_s crosses members before the out-projection).  config 2: 58 -> 54 dispatches,
// the step inside the chain 6.5 us against the 9.0 us launch (profiles/NOTES_r06.md section 5).
// Residency: at most 32 workgroups per XCD (one per CU: 110 KB of LDS each), all resident at once; a hand-off wait that
// runs out of patience sets *err and goes on (wrong numbers, never a hung GPU).
#include <atomic>

#include "attn_common.h"
#include "chain_common.h"

namespace {

namespace sa32 {   // the split-bf16 self-attention forward's device code (attn_sa.hip's d_h = 32 instantiation), kernels left out
#define SA_DH 32
#define SA_MAXT 512
#define SA_DEVICE_ONLY
#include "attn_sa_body.h"
#undef SA_DEVICE_ONLY
#undef SA_DH
#undef SA_MAXT
}  // namespace sa32

// steps 3 / 4 (the FFN products): a member owns 256 (linear1) / 128 (linear2) output columns of its row tile, every wave
// 4 / 2 independent 16 x 16 accumulators, the weights staged in k slabs of 64 / 128 (one 16-dword RawB load each)
constexpr int K3 = 64, LD3 = K3 + 8, N3 = 256;      // linear1: [256 columns][64 k] slab, x2 staged whole ([32][264])
constexpr int K4 = 128, LD4 = K4 + 8, N4 = 128;     // linear2: [128 columns][128 k] slab + [32][128 k] slab of h
constexpr int CL3 = N3 + 4, CL4 = N4 + 4;
// LDS (bytes) of the widest step at NRT row tiles per group: x2 planes of all row tiles + one W1 slab (the fp32 output tile
// overlays the slab once its last fragments are read)
constexpr size_t chain_lds(int nrt) {
  const size_t s1 = (size_t)2 * (TM + TN) * LDR * 2 + (size_t)TM * CLD * 4;
  const size_t s3a = (size_t)2 * N3 * LD3 * 2, s3c = (size_t)TM * CL3 * 4;
  const size_t s3 = (size_t)nrt * 2 * TM * LDR * 2 + (s3a > s3c ? s3a : s3c);
  const size_t s4b = (size_t)2 * N4 * LD4 * 2, s4c = (size_t)TM * CL4 * 4;
  const size_t s4 = (size_t)nrt * 2 * TM * LD4 * 2 + (s4b > s4c ? s4b : s4c);
  const size_t s6 = nrt == 1 ? proj_lds<1>(true) : proj_lds<2>(true);
  const size_t m = s1 > s3 ? (s1 > s4 ? s1 : s4) : (s3 > s4 ? s3 : s4);
  return m > s6 ? m : s6;
}
static_assert(chain_lds(2) <= 160 * 1024, "LDS");
// step 0 (optional): Q planes of the group's rows + K / V planes and the additive key term of `sc` scenes
constexpr size_t sa_step_lds(int gr, int lpk, int sc) {
  return (size_t)2 * gr * sa32::LDH * 2 + (size_t)sc * (4 * lpk * sa32::LDH * 2 + lpk * 4) + 16;
}

// NRT: 32-row tiles per group (1: up to 1024 rows; 2: up to 2048 -- a member converts each weight slab once for both tiles)
// SA: step 0 -- the self-attention core of the layer (the launch that used to sit between chain_ca and this one) -- runs here:
// member j = head j forms o_s[rows of the tile, 32 j .. 32 j + 32) from q / k / v written by the PREVIOUS launch (plain loads); a
// tile that straddles scenes runs one (16-row block, scene) task per wave over the planes of up to two scenes at a time.
template <int NRT, bool SA>
__global__ __launch_bounds__(CT) void chain_ffn_fwd_kernel(const pq3d_chain_ffn_desc d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ch_smem[];
  Ctx c;
  c.Ah = (bf16_t*)ch_smem;
  c.Al = c.Ah + TM * LDR;
  c.Bh = c.Al + TM * LDR;
  c.Bl = c.Bh + TN * LDR;
  c.Ct = (float*)(c.Bl + TN * LDR);
  c.tid = threadIdx.x; c.lane = c.tid & 63; c.wave = c.tid >> 6; c.li = c.lane & 15; c.lg = c.lane >> 4;
  c.wm = (c.wave >> 2) * 16; c.wn = (c.wave & 3) * 16;
  // group = NRT consecutive row tiles; its 8 members share id % 8 (= the XCD)
  constexpr int GR = TM * NRT;                    // rows per group
  const int id = (int)blockIdx.x, xcd = id & 7, q = id >> 3, slot = q >> 3, j = q & 7;
  const int grp = slot * 8 + xcd, m0 = grp * GR;
  const int R = d.R, F = d.F;
  if (m0 >= R) return;
  unsigned* const group = d.flags + (long)grp * G * 16;
  unsigned* const mine = group + j * 16;
  const unsigned v0 = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // equal for the whole group at launch

  CH_TL(0);
  unsigned tgt = v0;   // hand-off counter (one more hand-off with step 0)
  if constexpr (SA) {
    using namespace sa32;
    const int Nq = d.sa_nq, LPk = (Nq + 31) & ~31;
    const int row_hi = min(m0 + GR, R) - 1, s_lo = m0 / Nq, s_hi = row_hi / Nq;
    const int SC = sa_step_lds(GR, LPk, 2) <= 160 * 1024 ? 2 : 1;   // scenes staged at a time
    bf16_t* const Qh = (bf16_t*)ch_smem;
    bf16_t* const Ql = Qh + GR * LDH;
    unsigned char* const kvbase = (unsigned char*)(Ql + GR * LDH);
    const size_t per_scene = (size_t)4 * LPk * LDH * 2 + (size_t)LPk * 4;
    // Every load of the step in flight before the first conversion: the Q rows of the group (rows past R: clamped duplicates,
    // never stored) and, per chunk of scenes, the K / V rows + key padding (attn_sa_body.h's stage_planes, all tensors at once).
    static_assert(GR * 4 <= CT, "one 16-byte Q chunk for each of the first 4 GR threads");
    float qv[8];
    const bool has_q = c.tid < GR * 4;
    if (has_q) load8<false>(d.sa_q, (long)min(m0 + (c.tid >> 2), R - 1) * D + 32 * j + (c.tid & 3) * 8, qv);
    const int nch = LPk * 4;            // 16-byte chunks of one K / V plane pair
    for (int sc0 = s_lo; sc0 <= s_hi; sc0 += SC) {
      const int nsc = min(SC, s_hi - sc0 + 1);
      float kv[2][2][2][8];             // [scene of the chunk][K | V][chunk of the thread]
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          const int cidx = c.tid + w * CT, row = cidx >> 2, x = (cidx & 3) * 8;
          if (u < nsc && cidx < nch) {
            const long off = ((long)(sc0 + u) * Nq + min(row, Nq - 1)) * D + 32 * j + x;
            load8<false>(d.sa_k, off, kv[u][0][w]);
            load8<false>(d.sa_v, off, kv[u][1][w]);
          }
        }
      if (sc0 > s_lo) __syncthreads();   // the previous chunk's planes are consumed
      else if (has_q) {
        const HL sp = split8(qv);
        *(u32x4*)&Qh[(c.tid >> 2) * LDH + (c.tid & 3) * 8] = sp.hi;
        *(u32x4*)&Ql[(c.tid >> 2) * LDH + (c.tid & 3) * 8] = sp.lo;
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u >= nsc) break;
        bf16_t* const Kh = (bf16_t*)(kvbase + u * per_scene);
        bf16_t* const Kl = Kh + LPk * LDH;
        bf16_t* const Vh = Kl + LPk * LDH;
        bf16_t* const Vl = Vh + LPk * LDH;
        float* const kb = (float*)(Vl + LPk * LDH);
        const long r0 = (long)(sc0 + u) * Nq;
        for (int k = c.tid; k < LPk; k += CT) kb[k] = (k < Nq && !(d.sa_kpm && d.sa_kpm[r0 + k])) ? 0.f : -INFINITY;
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          const int cidx = c.tid + w * CT, row = cidx >> 2, x = (cidx & 3) * 8;
          if (cidx < nch) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              float v[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = row < Nq ? kv[u][t][w][e] : 0.f;   // rows >= N_q of the planes are zero
              const HL sp = split8(v);
              bf16_t* const hi = t ? Vh : Kh;
              bf16_t* const lo = t ? Vl : Kl;
              *(u32x4*)&hi[row * LDH + x] = sp.hi;
              *(u32x4*)&lo[row * LDH + x] = sp.lo;
            }
          }
        }
      }
      __syncthreads();
      int task = 0;
      for (int blk = 0; blk < GR / 16; ++blk)
        for (int u = 0; u < nsc; ++u) {
          const int sc = sc0 + u, b0 = m0 + blk * 16;
          if (b0 > row_hi || b0 + 15 < sc * Nq || b0 >= (sc + 1) * Nq) continue;   // the block has no row of this scene
          if ((task++ & 7) != c.wave) continue;
          SaLds S;
          S.Qh = Qh; S.Ql = Ql;
          S.Kh = (bf16_t*)(kvbase + u * per_scene); S.Kl = S.Kh + LPk * LDH; S.Vh = S.Kl + LPk * LDH; S.Vl = S.Vh + LPk * LDH;
          S.kb = (float*)(S.Vl + LPk * LDH);
          const int lrow = blk * 16 + c.li, grow = m0 + lrow;
          const bool mine_ = grow <= row_hi && grow >= sc * Nq && grow < (sc + 1) * Nq;
          const int brow = mine_ ? grow - sc * Nq : Nq;
          const float* bias = d.sa_bias ? d.sa_bias + ((long)sc * 8 + j) * Nq * (long)Nq : nullptr;
          const SaFwdOut r = sa_fwd_block(S, lrow, brow, bias, Nq, Nq, LPk, d.sa_scale, c.lg);
          if (mine_) {
            const float inv = r.l > 0.f ? 1.f / r.l : 0.f;
            float* o = (float*)d.o_s + (long)grow * D + 32 * j;
#pragma unroll
            for (int t = 0; t < OT; ++t)
              *(float4*)(o + 16 * t + 4 * c.lg) = make_float4(r.ot[t][0] * inv, r.ot[t][1] * inv, r.ot[t][2] * inv, r.ot[t][3] * inv);
            if (c.lg == 0) d.sa_lse[((long)sc * 8 + j) * Nq + brow] = r.l > 0.f ? r.m + logf(r.l) : -INFINITY;
          }
        }
    }
  }
  // Weights do not depend on the chain: the eight slabs a member needs (4 k slabs of its 256 W1 rows, 4 of its 128 W2 rows)
  // are requested ahead of their use through a ring of three register sets -- the first three before anything else, so they
  // travel under the out-projection, the first LayerNorm and two hand-offs.
  const int Fq = F / 4;
  const int kq = j >> 1, nh = (j & 1) * N4;      // step 4: this member's quarter of linear2's reduction, its half of the columns
  RawB ring[3];
  auto issue_w = [&](int l, RawB& r) {            // load l of this member's weight sequence: 2048 chunks of 8 floats each
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ch = c.tid + i * CT;
      if (l < 4) load8<false>(d.W1, (long)(j * N3 + (ch >> 3)) * D + l * K3 + (ch & 7) * 8, r.v[i]);
      else load8<false>(d.W2, (long)(nh + (ch >> 4)) * F + kq * Fq + (l - 4) * K4 + (ch & 15) * 8, r.v[i]);
    }
  };
  // ---- 1. out-projection: 4 NRT (row tile, 64-column tile) units, one per member
  {
    RawA ra; RawB rbo;
    const bool on = j < 4 * NRT && m0 + (j >> 2) * TM < R;
    const int mt = m0 + (j >> 2) * TM;
    if constexpr (SA) {   // Wo and the first weight slabs travel under the hand-off that publishes o_s
      if (on) issue_b(c, rbo, d.Wo, D, (j & 3) * TN, 0);
      issue_w(0, ring[0]); issue_w(1, ring[1]); issue_w(2, ring[2]);
      handoff(c, mine, group, ++tgt, d.err);
      if (on) issue_a<true>(c, ra, d.o_s, D, mt, R, 0);
    } else {
      if (on) {
        issue_a<false>(c, ra, d.o_s, D, mt, R, 0);
        issue_b(c, rbo, d.Wo, D, (j & 3) * TN, 0);
      }
      issue_w(0, ring[0]); issue_w(1, ring[1]); issue_w(2, ring[2]);
    }
    if (on) {
      put_a(c, ra); put_b(c, rbo);
      __syncthreads();
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      mma_chunk(c, acc);
      store_tile(c, acc, d.bo, (j & 3) * TN, false, d.f, D, mt, R);
    }
  }
  CH_TL(1);
  handoff(c, mine, group, ++tgt, d.err);
  CH_TL(2);
  // ---- 2. x2 = LN1(x1s + f): 32 NRT rows over 8 members x 4 NRT waves
  const long lrow = m0 + 4 * NRT * j + c.wave;
  const bool lnw = c.wave < 4 * NRT && lrow < R;
  {
    const float* o1[1] = {d.f};
    if (lnw) ln_row<false>(c, lrow, d.x1s, o1, 1, 0, d.g1, d.be1, d.eps1, nullptr, d.x2, d.mean1, d.rstd1);
  }
  CH_TL(3);
  handoff(c, mine, group, ++tgt, d.err);
  CH_TL(4);
  const int wr = (c.wave >> 2) * 16;              // steps 3 / 4: this wave's 16 rows of every row tile
  // ---- 3. h = relu(x2 W1^T + b1): member j owns columns [256 j, 256 j + 256); wave = 16 rows x 64 columns per row tile
  {
    bf16_t* const Ah3 = (bf16_t*)ch_smem;         // [NRT][2 planes][32][LDR]: x2, whole K
    bf16_t* const Bh3 = Ah3 + NRT * 2 * TM * LDR; // [256][LD3] x 2 planes: one k slab of W1
    bf16_t* const Bl3 = Bh3 + N3 * LD3;
    float* const Ct3 = (float*)Bh3;               // [32][CL3], over the slab once it is dead
    const int wc = (c.wave & 3) * 64;
#pragma unroll
    for (int t = 0; t < NRT; ++t) {
      RawA ra;
      issue_a<true>(c, ra, d.x2, D, m0 + t * TM, R, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ch = c.tid + i * CT, o = (ch >> 5) * LDR + (ch & 31) * 8;
        u32x4 hi, lo;
        split_hi_lo(ra.v[i], hi, lo);
        *(u32x4*)&Ah3[(t * 2) * TM * LDR + o] = hi;
        *(u32x4*)&Ah3[(t * 2 + 1) * TM * LDR + o] = lo;
      }
    }
    f32x4 acc[NRT][4];
#pragma unroll
    for (int t = 0; t < NRT; ++t)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) acc[t][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      if (l > 0) __syncthreads();                 // the previous slab's fragment reads are done
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ch = c.tid + i * CT, o = (ch >> 3) * LD3 + (ch & 7) * 8;
        u32x4 hi, lo;
        split_hi_lo(ring[l % 3].v[i], hi, lo);
        *(u32x4*)&Bh3[o] = hi;
        *(u32x4*)&Bl3[o] = lo;
      }
      __syncthreads();
      issue_w(l + 3, ring[l % 3]);                // three loads ahead (runs into W2's slabs)
#pragma unroll
      for (int ks = 0; ks < K3 / 32; ++ks) {
        u32x4 bh[4], bl[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          const int ob = (wc + nb * 16 + c.li) * LD3 + ks * 32 + c.lg * 8;
          bh[nb] = *(const u32x4*)&Bh3[ob];
          bl[nb] = *(const u32x4*)&Bl3[ob];
        }
#pragma unroll
        for (int t = 0; t < NRT; ++t) {
          const int oa = (wr + c.li) * LDR + l * K3 + ks * 32 + c.lg * 8;
          const u32x4 ah = *(const u32x4*)&Ah3[(t * 2) * TM * LDR + oa], al = *(const u32x4*)&Ah3[(t * 2 + 1) * TM * LDR + oa];
#pragma unroll
          for (int nb = 0; nb < 4; ++nb) {
            Mma<bf16_t>::mma(acc[t][nb], al, bh[nb]);
            Mma<bf16_t>::mma(acc[t][nb], ah, bl[nb]);
            Mma<bf16_t>::mma(acc[t][nb], ah, bh[nb]);
          }
        }
      }
    }
    float bcol[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) bcol[nb] = d.b1[j * N3 + wc + nb * 16 + c.li];
#pragma unroll
    for (int t = 0; t < NRT; ++t) {
      __syncthreads();                            // the slab's last reads / the previous row tile's output reads are done
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) Ct3[(wr + c.lg * 4 + r) * CL3 + wc + nb * 16 + c.li] = (acc[t][nb][r] + bcol[nb]) * 1.f;
      __syncthreads();
      const int orow = c.tid >> 4, row = m0 + t * TM + orow;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int col = qd * 64 + (c.tid & 15) * 4;
        float4 v = *(const float4*)&Ct3[orow * CL3 + col];
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        if (row < R) *(float4*)(d.h + (long)row * F + j * N3 + col) = v;
      }
    }
  }
  CH_TL(5);
  handoff(c, mine, group, ++tgt, d.err);
  CH_TL(6);
  // ---- 4. zp_k = h[:, quarter k] W2[:, quarter k]^T (+ b2 at k = 0): member j owns quarter j / 2 and columns [128 (j & 1), + 128);
  // wave = 16 rows x 32 columns per row tile (2 accumulators), k slabs of 128
  {
    bf16_t* const Ah4 = (bf16_t*)ch_smem;         // [NRT][2 planes][32][LD4]: one k slab of h
    bf16_t* const Bh4 = Ah4 + NRT * 2 * TM * LD4; // [128][LD4] x 2 planes: one k slab of W2
    bf16_t* const Bl4 = Bh4 + N4 * LD4;
    float* const Ct4 = (float*)Bh4;               // [32][CL4], over the slab once it is dead
    const int wc = (c.wave & 3) * 32;
    float av[NRT][8];
    auto issue_h = [&](int l) {                   // [32 NRT rows][128 k] of h: 512 NRT chunks, NRT per thread
#pragma unroll
      for (int t = 0; t < NRT; ++t)
        load8<true>(d.h, (long)min(m0 + t * TM + (c.tid >> 4), R - 1) * F + kq * Fq + l * K4 + (c.tid & 15) * 8, av[t]);
    };
    issue_h(0);
    f32x4 acc[NRT][2];
#pragma unroll
    for (int t = 0; t < NRT; ++t) { acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    __syncthreads();                              // step 3's reads of the LDS this step overlays are done
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      if (l > 0) __syncthreads();
#pragma unroll
      for (int t = 0; t < NRT; ++t) {
        u32x4 hi, lo;
        split_hi_lo(av[t], hi, lo);
        const int o = (c.tid >> 4) * LD4 + (c.tid & 15) * 8;
        *(u32x4*)&Ah4[(t * 2) * TM * LD4 + o] = hi;
        *(u32x4*)&Ah4[(t * 2 + 1) * TM * LD4 + o] = lo;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ch = c.tid + i * CT, o = (ch >> 4) * LD4 + (ch & 15) * 8;
        u32x4 hi, lo;
        split_hi_lo(ring[(4 + l) % 3].v[i], hi, lo);
        *(u32x4*)&Bh4[o] = hi;
        *(u32x4*)&Bl4[o] = lo;
      }
      __syncthreads();
      if (l + 1 < 4) issue_h(l + 1);
      if (4 + l + 3 < 8) issue_w(4 + l + 3, ring[(4 + l) % 3]);
#pragma unroll
      for (int ks = 0; ks < K4 / 32; ++ks) {
        u32x4 bh[2], bl[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const int ob = (wc + nb * 16 + c.li) * LD4 + ks * 32 + c.lg * 8;
          bh[nb] = *(const u32x4*)&Bh4[ob];
          bl[nb] = *(const u32x4*)&Bl4[ob];
        }
#pragma unroll
        for (int t = 0; t < NRT; ++t) {
          const int oa = (wr + c.li) * LD4 + ks * 32 + c.lg * 8;
          const u32x4 ah = *(const u32x4*)&Ah4[(t * 2) * TM * LD4 + oa], al = *(const u32x4*)&Ah4[(t * 2 + 1) * TM * LD4 + oa];
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            Mma<bf16_t>::mma(acc[t][nb], al, bh[nb]);
            Mma<bf16_t>::mma(acc[t][nb], ah, bl[nb]);
            Mma<bf16_t>::mma(acc[t][nb], ah, bh[nb]);
          }
        }
      }
    }
    float bcol[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) bcol[nb] = kq == 0 ? d.b2[nh + wc + nb * 16 + c.li] : 0.f;
    float* const zq = d.zp + (long)kq * R * D;
#pragma unroll
    for (int t = 0; t < NRT; ++t) {
      __syncthreads();
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) Ct4[(wr + c.lg * 4 + r) * CL4 + wc + nb * 16 + c.li] = (acc[t][nb][r] + bcol[nb]) * 1.f;
      __syncthreads();
      const int orow = c.tid >> 4, row = m0 + t * TM + orow;
#pragma unroll
      for (int qd = 0; qd < 2; ++qd) {
        const int col = qd * 64 + (c.tid & 15) * 4;
        const float4 v = *(const float4*)&Ct4[orow * CL4 + col];
        if (row < R) *(float4*)(zq + (long)row * D + nh + col) = v;
      }
    }
  }
  // step 6's weights (the ring is free now) travel under the last hand-offs and the second LayerNorm
  RawB wq[2];
  if (d.nq > 0) proj_issue_w(c, j, d.nq, d.Wq, wq);
  CH_TL(7);
  handoff(c, mine, group, ++tgt, d.err);
  CH_TL(8);
  // ---- 5. z = sum of the partials, x3 = LN2(x2 + z)
  {
    const float* o2[1] = {d.zp};
    if (lnw) ln_row<true>(c, lrow, d.x2, o2, 4, (long)R * D, d.g2, d.be2, d.eps2, d.z, d.x3, d.mean2, d.rstd2);
  }
  CH_TL(9);
  // ---- 6. (optional) the next layer application's cross-attention query projections from x3 + qpos
  if (d.nq > 0) {   // uniform
    handoff(c, mine, group, ++tgt, d.err);
    const void* A[3] = {d.x3, d.x3, d.x3};
    const float* A2[3] = {d.qpos, d.qpos, d.qpos};
    if (d.qout_f32) {   // uniform: fp32 queries (compute mode 'bf16x3': the split-bf16 cross-attention splits them itself)
      float* out[3] = {(float*)d.qout[0], (float*)d.qout[1], (float*)d.qout[2]};
      proj_3x256<NRT, true, true, float>(c, ch_smem, j, d.nq, m0, R, A, A2, d.Wq, d.bq, out, wq, true);
    } else {
      bf16_t* out[3] = {(bf16_t*)d.qout[0], (bf16_t*)d.qout[1], (bf16_t*)d.qout[2]};
      proj_3x256<NRT, true, true, bf16_t>(c, ch_smem, j, d.nq, m0, R, A, A2, d.Wq, d.bq, out, wq, true);
    }
  }
}

}  // namespace

extern "C" int pq3d_chain_ffn_fwd(const pq3d_chain_ffn_desc* dp, void* stream) {
  PQ_DEVICE_GUARD(stream, dp ? dp->o_s : nullptr);
  PQ_CHECK_ARG(dp != nullptr, "pq3d_chain_ffn_fwd: null descriptor");
  const pq3d_chain_ffn_desc d = *dp;
  PQ_CHECK_ARG(d.R >= 1 && d.d == D && d.F == 2048, "pq3d_chain_ffn_fwd: d = 256, F = 2048");
  const int row_tiles = (d.R + TM - 1) / TM;
  const int nrt = chain_nrt(row_tiles);
  const int groups = (row_tiles + nrt - 1) / nrt, slots = (groups + 7) / 8;
  PQ_CHECK_ARG(slots * G <= 32, "pq3d_chain_ffn_fwd: more than 2048 rows (the groups would not all be resident)");
  PQ_CHECK_ARG(d.o_s && d.Wo && d.bo && d.x1s && d.g1 && d.be1 && d.f && d.x2 && d.mean1 && d.rstd1 && d.W1 && d.b1 && d.h && d.W2 && d.b2 &&
               d.zp && d.z && d.g2 && d.be2 && d.x3 && d.mean2 && d.rstd2 && d.flags, "pq3d_chain_ffn_fwd: null pointer");
  const void* al[] = {d.o_s, d.Wo, d.x1s, d.g1, d.be1, d.f, d.x2, d.W1, d.h, d.W2, d.zp, d.z, d.g2, d.be2, d.x3};
  for (const void* p : al) PQ_CHECK_ARG((((uintptr_t)p) & 15) == 0, "pq3d_chain_ffn_fwd: operands must be 16-byte aligned");
  PQ_CHECK_ARG((long)d.R * d.F * 4 < 0x7ffffff0L, "pq3d_chain_ffn_fwd: hidden activations too large");
  PQ_CHECK_ARG(d.nq >= 0 && d.nq <= 3 && (d.nq == 0 || (d.qpos && (((uintptr_t)d.qpos) & 15) == 0)), "pq3d_chain_ffn_fwd: 0..3 query projections");
  for (int m = 0; m < d.nq; ++m)
    PQ_CHECK_ARG(d.Wq[m] && d.bq[m] && d.qout[m] && ((((uintptr_t)d.Wq[m]) | ((uintptr_t)d.qout[m])) & 15) == 0,
                 "pq3d_chain_ffn_fwd: query-projection operands (non-null, 16-byte aligned)");
  const bool sa = d.sa_q != nullptr;
  size_t lds = chain_lds(nrt);
  if (sa) {
    PQ_CHECK_ARG(d.sa_k && d.sa_v && d.sa_lse && d.sa_nq >= 1 && d.sa_nq <= 240 && d.R % d.sa_nq == 0,
                 "pq3d_chain_ffn_fwd: self-attention step needs q / k / v / lse, 1 <= sa_nq <= 240 queries per scene dividing R");
    PQ_CHECK_ARG(((((uintptr_t)d.sa_q) | ((uintptr_t)d.sa_k) | ((uintptr_t)d.sa_v) | ((uintptr_t)d.sa_bias)) & 15) == 0 ,
                 "pq3d_chain_ffn_fwd: self-attention operands must be 16-byte aligned");
    const int lpk = (d.sa_nq + 31) & ~31, gr = TM * nrt;
    const size_t need = sa_step_lds(gr, lpk, 2) <= 160 * 1024 ? sa_step_lds(gr, lpk, 2) : sa_step_lds(gr, lpk, 1);
    PQ_CHECK_ARG(need <= 160 * 1024, "pq3d_chain_ffn_fwd: self-attention step does not fit the LDS");
    if (need > lds) lds = need;
  }
  static std::atomic<unsigned> done[4] = {{0}, {0}, {0}, {0}};
  const dim3 grid((unsigned)(8 * G * slots));
  auto go = [&](auto kern, std::atomic<unsigned>& dn) -> int {
    if (int e = pq3d_enable_big_lds(kern, 160 * 1024, dn)) { pq3d_set_error(hipGetErrorString((hipError_t)e)); return e; }
    hipLaunchKernelGGL(kern, grid, dim3(CT), lds, (hipStream_t)stream, d);
    return 0;
  };
  int e = 0;
  if (nrt == 1) e = sa ? go(chain_ffn_fwd_kernel<1, true>, done[0]) : go(chain_ffn_fwd_kernel<1, false>, done[1]);
  else e = sa ? go(chain_ffn_fwd_kernel<2, true>, done[2]) : go(chain_ffn_fwd_kernel<2, false>, done[3]);
  if (e) return e;
  PQ_LAUNCH_CHECK();
  return 0;
}
