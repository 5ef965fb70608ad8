// All-queries-resident single-pass attention backward for the decoder's cross-attention shape: FEW queries (N_q <= 224),
// MANY keys (N_seg in the thousands), small heads (d_h 32 / 64), bf16 operands.
//
// The two-kernel recompute backward of attention.hip forms S and dP twice (once per query chunk for dQ, once per key
// chunk for dK/dV), stages K/V (or Q/dO) tiles through LDS behind a barrier per tile, and re-reads Q, K, V, dO in both
// kernels (2.6x the algorithmic HBM traffic, MFMA busy < 10 %: profiles/rocprofv3_pmc_attention_r01_c2.txt).  Here one
// workgroup owns ALL queries of a (scene, head[, key slice]):
//   * Q and dO of the head (N_q x d_h each, 14 KB at config 2) are loaded ONCE into LDS and stay there, with
//     delta = rowsum(dO * O) and the log-sum-exp row -- the main loop has no barrier and no LDS staging at all;
//   * each wave walks 32-key groups with its K / V rows in registers (prefetched one group ahead), forms
//     S = Q K^T and dP = dO V^T ONCE per (query tile, key tile), and from the same P / dS tiles accumulates
//       dV^T += dO^T P      dK^T += Q^T dS       (registers: the wave owns its keys -> plain stores, no reduction)
//       dQ   += dS K                             (fp32 LDS accumulator shared by the 4 waves: ds_add_f32)
//     -- 5 products for 20 MFMAs per (32 queries x 32 keys), the algorithmic count;
//   * dS reaches the dQ product through a wave-private [key][query] LDS scratch tile written with one 8-byte store per
//     key tile and read back with the transposing LDS read, K^T likewise from a wave-private copy of the K tile;
//   * key groups are dealt round-robin over (slice, wave), so the padded tail of a scene is spread evenly; fully padded
//     groups cost one mask test.  Key slices (ksplit, a function of N_seg only) leave fp32 dQ partials that the existing
//     combine kernel sums.
// Same arithmetic as the two-kernel path (P from the saved log-sum-exp, dS = P (dP - delta) scale, bf16 operands, fp32
// accumulation); dQ now sums its key contributions in a different (non-deterministic) order.
#include <cstdlib>

#include "attn_common.h"

namespace {

constexpr int GK = 32;                // keys per wave iteration (two 16-key MFMA tiles)

template <int DH> struct RT {
  typedef AT<bf16_t, DH> A;
  static constexpr int LDR = A::LDR;          // row stride of the resident row-major Q / dO tiles (elements)
  static constexpr int LDQ = DH + 1;          // row stride of the fp32 dQ accumulator (floats)
  static constexpr int LDS2 = 16 + 4;         // row stride of the [key][query] dS scratch (elements)
  static constexpr int KT_ELEMS = GK * A::LDR;
  static constexpr int SC_ELEMS = 4 * GK * LDS2;   // one scratch tile per query tile of a pair, two pairs in flight
  static constexpr int MLD = GK + 8;          // row stride of the staged 3-D mask tile [32 queries][32 keys] (bytes)
};

typedef short v4i16r_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16r_t lds_v4i16r_t;
// fragment (row r0 + li, k-slots 8 lg .. 8 lg + 7) of an operand stored [k][m] in LDS (gemm.hip's km_frag)
PQ_DEV u32x4 kmajor_frag(const bf16_t* tile, int ld, int r0, int li, int lg) {
  const bf16_t* p0 = tile + (8 * lg + (li >> 2)) * ld + r0 + 4 * (li & 3);
  const v4i16r_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16r_t*)p0);
  const v4i16r_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16r_t*)(p0 + 4 * ld));
  const u32x2 lo = __builtin_bit_cast(u32x2, a), hi = __builtin_bit_cast(u32x2, b);
  return (u32x4){lo.x, lo.y, hi.x, hi.y};
}

PQ_DEV void wave_lds_fence() {   // order this wave's LDS writes before its following LDS reads (other lanes' data)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The workgroup's resident queries are rows [q_lo, q_lo + nq) of the scene (nq <= 32 NQP).  More than 128 queries (config
// 4: 200) run as TWO launches over the two halves of the rows; the second one ADDS its dK / dV onto the first one's
// (acc_kv: read-modify-write of the bf16 rows; each key belongs to exactly one wave of one workgroup per launch).
// in-kernel timeline (probe builds only, tools/probes/res_timeline.py): wave 0 of workgroup (0, 0, 0) stamps the 100 MHz clock
#ifdef PQ3D_RES_TIMELINE
#define RS_TL(i) do { if (d.ws && d.ksplit <= 1 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) ((long long*)d.ws)[i] = wall_clock64(); } while (0)
#else
#define RS_TL(i) do { } while (0)
#endif

// ZA (only without dropout / 3-D mask): the call has the zero key of add_zero_attn, so every log-sum-exp is >= 0 and the
// probability recomputed for a padded key (score 0: its K row is zeroed) is at most 1 -- it needs no forcing to 0 at all
// (its dK / dV rows are zeroed at the store, its dQ term multiplies the zero K row).
// MASK3: 0 = no 3-D mask, 1 = mask BYTES [B,Lq,Lk] + row_open flags (staged per (query pair, key group) through a wave-private
// LDS tile), 2 = mask BITS with row_open folded in (pq3d_mask_pack: one 32-bit word per query and 32-key group = exactly this
// kernel's key group; 32 lanes fetch the pair's 32 words, ds_bpermute hands every lane its 4 query rows' words)
template <int DH, int NQP, bool DROP, int MASK3, int RW, bool ZA = false>
__global__ __launch_bounds__(RW * 64) void attn_bwd_resident_kernel(const pq3d_attn_desc d, int q_lo, int nq, int acc_kv) {
  ATTN_KARG_PIN(d);
  ATTN_KARG_PIN_BWD(d);
  typedef AT<bf16_t, DH> A;
  typedef RT<DH> R;
  constexpr int NQ = NQP * 32;           // resident (padded) query rows
  constexpr int NS = A::NS, MT = A::MT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* Qs = (bf16_t*)smem;                                  // [NQ][LDR]
  bf16_t* dOs = Qs + NQ * R::LDR;                              // [NQ][LDR]
  float* Ls = (float*)(dOs + NQ * R::LDR);                     // [NQ]  -log2(e) * logsumexp (-inf past Lq)
  float* Ds = Ls + NQ;                                         // [NQ]  -delta
  float* dQs = (float*)smem;                                   // [NQ][LDQ] fp32: end-of-kernel sum over the waves, OVER Qs / dOs
  static_assert(NQ * R::LDQ * 4 <= 2 * NQ * R::LDR * 2, "the dQ reduction buffer must fit over the resident Q / dO tiles");
  bf16_t* wv = (bf16_t*)(Ds + NQ);                             // per wave: K tile + dS scratch (+ mask tile)
  constexpr int WV_ELEMS = R::KT_ELEMS + R::SC_ELEMS + (MASK3 == 1 ? (32 * R::MLD) / 2 : 0);
  uint8_t* ros = (uint8_t*)(wv + RW * WV_ELEMS);               // [NQ] row-open flags (MASK3)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const WgXyz wg = attn_wg_xyz((int)gridDim.x);
  const int b = wg.b, h = wg.h;
  const int KS = d.ksplit > 1 ? d.ksplit : 1;
  const int split = wg.x;
  const int bm = d.mask_bmod > 0 ? b % d.mask_bmod : b;
  const float sl2 = d.scale * 1.4426950408889634f;

  RS_TL(0);
  // ---- resident tiles: Q, dO (zero past Lq), delta = rowsum(dO * O), log-sum-exp
  {
    const long qoff = (long)b * d.q_sb + (long)h * d.q_sh, ooff = (long)b * d.o_sb + (long)h * d.o_sh;
    const long sbase = ((long)b * d.H + h) * d.Lq + q_lo;
    constexpr int CPR = A::CPR, TOTAL = NQ * CPR;
#pragma unroll
    for (int it = 0; it < (TOTAL + RW * 64 - 1) / (RW * 64); ++it) {
      const int c = tid + it * RW * 64;
      const int row = c / CPR, kc = c % CPR;
      const bool ok = c < TOTAL && row < nq;
      const int gr = q_lo + min(row, nq - 1);
      u32x4 q = *(const u32x4*)((const bf16_t*)d.q + qoff + (long)gr * d.q_sl + kc * 8);
      u32x4 g = *(const u32x4*)((const bf16_t*)d.dout + ooff + (long)gr * d.o_sl + kc * 8);
      const u32x4 o = *(const u32x4*)((const bf16_t*)d.o + ooff + (long)gr * d.o_sl + kc * 8);
      if (!ok) { q = (u32x4){0, 0, 0, 0}; g = (u32x4){0, 0, 0, 0}; }
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s += __uint_as_float(g[j] << 16) * __uint_as_float(o[j] << 16);
        s += __uint_as_float(g[j] & 0xffff0000u) * __uint_as_float(o[j] & 0xffff0000u);
      }
      // the CPR (4 or 8) chunks of a row sit in neighbouring lanes
#pragma unroll
      for (int o_ = 1; o_ < CPR; o_ <<= 1) s += __shfl_xor(s, o_, 64);
      if (c < TOTAL) {
        *(u32x4*)&Qs[row * R::LDR + kc * 8] = q;
        *(u32x4*)&dOs[row * R::LDR + kc * 8] = g;
        if (kc == 0) {
          Ds[row] = ok ? -s : 0.f;     // both rows are kept NEGATED: the loop adds them (accumulator start, fma addend)
          Ls[row] = ok ? d.lse[sbase + row] * -1.4426950408889634f : -INFINITY;
          if (ok && split == 0) d.delta[sbase + row] = s;
          if constexpr (MASK3 == 1) ros[row] = (ok && d.row_open) ? d.row_open[(long)bm * d.Lq + q_lo + row] : 0;
        }
      }
    }
  }
  RS_TL(1);
  __syncthreads();
  RS_TL(2);

  bf16_t* Kt = wv + wave * WV_ELEMS;            // [GK][LDR]  this wave's K tile, row-major
  bf16_t* sc = Kt + R::KT_ELEMS;                // [2][GK][LDS2] dS scratch, [key][query]
  uint8_t* mt_ = (uint8_t*)(sc + R::SC_ELEMS);  // [32][MLD] staged 3-D mask bytes of (query pair, key group) (MASK3)

  const long koff = (long)b * d.k_sb + (long)h * d.k_sh, voff = (long)b * d.v_sb + (long)h * d.v_sh;
  const bf16_t* const kbase = (const bf16_t*)d.k + koff;
  const bf16_t* const vbase = (const bf16_t*)d.v + voff;
  const int k_sl32 = (int)d.k_sl, v_sl32 = (int)d.v_sl;
  const uint8_t* kpm = d.kpm ? d.kpm + (long)b * d.Lk : nullptr;
  const int ngroups = (d.Lk + GK - 1) / GK;
  DropState dst;
  uint32_t drow0 = 0;
  if constexpr (DROP) {
    dst = drop_init(d.drop, d.drop_bmod > 0 ? b / d.drop_bmod : 0, d.Lk);
    drow0 = (uint32_t)(((long)(d.drop_bmod > 0 ? b % d.drop_bmod : b) * d.H + h) * d.Lq + q_lo);
  }

  // group g of the scene goes to (slice, wave) = ((g / RW) % KS, g % RW): round-robin, the padded tail is spread evenly
  auto group_of = [&](int it) { return (it * KS + split) * RW + wave; };
  u32x4 kf[2][NS], vf[2][NS], kfn[2][NS], vfn[2][NS];
  bool km[2], kmn[2];
  // 3-D mask: the [32 queries x 32 keys] byte tile of a (query pair, key group) is exactly one 16-byte chunk per lane;
  // it is requested one pair ahead of its use and parked in a wave-private LDS tile
  u32x4 mreg = (u32x4){0, 0, 0, 0};
  uint32_t wreg = 0;                          // MASK3 == 2: the mask word of query (lane & 31) of the requested pair
  const int mwords = (d.Lk + 31) >> 5;
  auto load_group = [&](int g, u32x4 (&kd)[2][NS], u32x4 (&vd)[2][NS], bool (&md)[2]) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const int key = g * GK + kt * 16 + li;
      const int ck = min(key, d.Lk - 1);
      // (32-bit products: a scene's K / V rows span < 2^31 elements -- checked on the host; the 64-bit multiplies were 50
      // vector-ALU instructions per group)
      row_frags<bf16_t, DH>(kd[kt], kbase, (long)(ck * k_sl32), lg);
      row_frags<bf16_t, DH>(vd[kt], vbase, (long)(ck * v_sl32), lg);
      md[kt] = key < d.Lk ? (kpm ? kpm[key] != 0 : false) : true;
    }
  };
  auto load_mask = [&](int g, int qp) {
    if constexpr (MASK3 == 2) {
      const int row = q_lo + min(qp * 32 + (lane & 31), nq - 1);
      wreg = d.mask_bits[((long)bm * d.Lq + row) * mwords + g];
    }
    if constexpr (MASK3 == 1) {
      const int row = q_lo + min(qp * 32 + (lane >> 1), nq - 1), half = lane & 1;   // chunk: (query row, 16-key half)
      const long kc = min((long)g * GK + half * 16, (long)d.Lk - 16);
      mreg = *(const u32x4*)(d.mask + ((long)bm * d.Lq + row) * d.Lk + kc);
    }
  };

  // dQ accumulators: C-layout tiles [query 4 lg + r][dh mt * 16 + li] of every query tile, summed over this wave's key
  // groups in registers and over the 4 waves through LDS once at the end.  (An fp32 LDS accumulator shared by the waves
  // -- ds_add_f32 per tile -- was measured first: LDS float atomics serialise per lane, ~1400 cycles per instruction,
  // 189 us instead of 40 us at config 2.)
  f32x4 accQ[NQP * 2][MT];
#pragma unroll
  for (int t = 0; t < NQP * 2; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) accQ[t][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int it = 0;
  int g = group_of(0);
  if (g < ngroups) { load_group(g, kf, vf, km); load_mask(g, 0); }
  for (; g < ngroups; ++it) {
    const int gn = group_of(it + 1);
    // ---- this group's operands into place; next group's loads in flight
    const bool all_masked = __all(km[0] && km[1]);
    if (!all_masked) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s = 0; s < NS; ++s)
          *(u32x4*)&Kt[(kt * 16 + li) * R::LDR + s * 32 + lg * 8] = km[kt] ? (u32x4){0, 0, 0, 0} : kf[kt][s];
    }
    // A padded key needs no per-score mask test: its K row is ZEROED here (so its scores are 0, P = exp2(-lse) is finite,
    // and it adds nothing to dQ = dS K), and its dK / dV rows are replaced by zeros at the store.
    u32x4 kcur[2][NS], vcur[2][NS];
    bool kmc[2] = {km[0], km[1]};
    // a padded key (lane li of key tile kt) gets its probability forced to 0 without a select per element: its score
    // accumulator STARTS at -1e30 (its K row is zeroed, so nothing is added), exp2 of that is exactly 0.  (With the score
    // left at 0, exp2(-lse) could overflow for a strongly negative lse and inf * 0 at the dK / dV store would be NaN.)
    const float sinit[2] = {(!ZA && !DROP && km[0]) ? -1e30f : 0.f, (!ZA && !DROP && km[1]) ? -1e30f : 0.f};
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        kcur[kt][s] = kmc[kt] ? (u32x4){0, 0, 0, 0} : kf[kt][s];
        vcur[kt][s] = vf[kt][s];
      }
    if (gn < ngroups) load_group(gn, kf, vf, km);
    const int nqp = (nq + 31) / 32;
    if (all_masked && gn < ngroups) load_mask(gn, 0);

    f32x4 accK[2][MT], accV[2][MT];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) { accK[kt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f}; accV[kt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    if (!all_masked) {
      wave_lds_fence();
      u32x4 ktf[MT][GK / 32];     // K as the B operand of dQ = dS K: column dh, k = keys
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) ktf[mt][0] = kmajor_frag(Kt, R::LDR, mt * 16, li, lg);
#pragma unroll
      for (int qp = 0; qp < NQP; ++qp) {
        if (qp >= nqp) break;
        if constexpr (MASK3 == 1) {   // park this pair's mask tile, request the next one (next pair, or pair 0 of the next group)
          *(u32x4*)&mt_[(lane >> 1) * R::MLD + (lane & 1) * 16] = mreg;
          if (qp + 1 < nqp) load_mask(g, qp + 1);
          else if (gn < ngroups) load_mask(gn, 0);
          wave_lds_fence();
        }
        uint32_t wq[2][4];            // MASK3 == 2: mask words of this lane's query rows (4 lg + r of both query tiles)
        if constexpr (MASK3 == 2) {
          const uint32_t wcur = wreg;
#pragma unroll
          for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int r = 0; r < 4; ++r) wq[qt][r] = (uint32_t)__shfl((int)wcur, qt * 16 + 4 * lg + r);
          if (qp + 1 < nqp) load_mask(g, qp + 1);
          else if (gn < ngroups) load_mask(gn, 0);
        }
        float pt[2][2][4], ds[2][2][4];   // [kt][qt][r]
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          const int q0 = qp * 32 + qt * 16;
          if (q0 >= nq) {   // uniform: a query tile that is all padding (100 queries: the last of 8) -- P = dS = 0, no arithmetic
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
              for (int r = 0; r < 4; ++r) { pt[kt][qt][r] = 0.f; ds[kt][qt][r] = 0.f; }
            continue;
          }
          u32x4 qa[NS], ga[NS];
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            qa[s] = rfrag<bf16_t>(&Qs[(q0 + li) * R::LDR], s, lg);
            ga[s] = rfrag<bf16_t>(&dOs[(q0 + li) * R::LDR], s, lg);
          }
          const f32x4 nL = *(const f32x4*)&Ls[q0 + 4 * lg], nD = *(const f32x4*)&Ds[q0 + 4 * lg];   // -lse log2(e), -delta
#pragma unroll
          for (int kt = 0; kt < 2; ++kt) {
            // without dropout the dP accumulator STARTS at -delta (the first MFMA takes the negated delta fragment as its C
            // operand): dP - delta costs no instruction at all (it was 64 v_sub_f32 per 32-key group)
            f32x4 sv = (f32x4){sinit[kt], sinit[kt], sinit[kt], sinit[kt]}, dp = DROP ? (f32x4){0.f, 0.f, 0.f, 0.f} : nD;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
              Mma<bf16_t>::mma(sv, qa[s], kcur[kt][s]);
              Mma<bf16_t>::mma(dp, ga[s], vcur[kt][s]);
            }
            if constexpr (!DROP) {
              // the softmax recompute is what this kernel spends its time on (in-kernel timeline at config 2: ~4 us per
              // 32-key group, 700 vector-ALU issue slots against 80 MFMAs): two elements per instruction where the ISA
              // has packed fp32 forms (fma, subtract, multiply), and the padded-key select only in groups that have one
              const f32x2 c2 = (f32x2){sl2, sl2};
              const f32x2 xa = __builtin_elementwise_fma(__builtin_shufflevector(sv, sv, 0, 1), c2, __builtin_shufflevector(nL, nL, 0, 1));
              const f32x2 xb = __builtin_elementwise_fma(__builtin_shufflevector(sv, sv, 2, 3), c2, __builtin_shufflevector(nL, nL, 2, 3));
              // P from the saved log-sum-exp; a padded key's score started at -1e30 (sinit), so its probability is exactly 0
              f32x2 pa = (f32x2){__builtin_amdgcn_exp2f(xa.x), __builtin_amdgcn_exp2f(xa.y)};
              f32x2 pb = (f32x2){__builtin_amdgcn_exp2f(xb.x), __builtin_amdgcn_exp2f(xb.y)};
              if constexpr (MASK3 != 0) {    // masked (query, key) pairs: probability exactly 0
                bool mk[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  if constexpr (MASK3 == 2) mk[r] = (wq[qt][r] >> (kt * 16 + li)) & 1u;
                  else mk[r] = (!ros[q0 + 4 * lg + r]) && (mt_[(qt * 16 + 4 * lg + r) * R::MLD + kt * 16 + li] != 0);
                }
                pa.x = mk[0] ? 0.f : pa.x; pa.y = mk[1] ? 0.f : pa.y; pb.x = mk[2] ? 0.f : pb.x; pb.y = mk[3] ? 0.f : pb.y;
              }
              const f32x2 da = pa * __builtin_shufflevector(dp, dp, 0, 1);   // dp already holds dP - delta
              const f32x2 db = pb * __builtin_shufflevector(dp, dp, 2, 3);
              pt[kt][qt][0] = pa.x; pt[kt][qt][1] = pa.y; pt[kt][qt][2] = pb.x; pt[kt][qt][3] = pb.y;
              ds[kt][qt][0] = da.x; ds[kt][qt][1] = da.y; ds[kt][qt][2] = db.x; ds[kt][qt][3] = db.y;
            } else
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int ql = q0 + 4 * lg + r;
              float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[r], sl2, nL[r]));   // P from the saved log-sum-exp
              // padded key: its K row is zeroed, so the score is 0 and exp2(-lse) can overflow for a strongly negative lse
              // (no zero key); inf * 0 at the dK / dV store would be NaN -> force the probability itself to 0
              p = kmc[kt] ? 0.f : p;
              if constexpr (MASK3 == 1) {
                const bool masked = (!ros[ql]) && (mt_[(qt * 16 + 4 * lg + r) * R::MLD + kt * 16 + li] != 0);
                p = masked ? 0.f : p;
              }
              if constexpr (MASK3 == 2) p = ((wq[qt][r] >> (kt * 16 + li)) & 1u) ? 0.f : p;
              if constexpr (DROP) {
                const float kc = drop_keep(dst, drow0 + (uint32_t)min(ql, nq - 1), (uint32_t)min(g * GK + kt * 16 + li, d.Lk - 1)) ? dst.scale : 0.f;
                pt[kt][qt][r] = p * kc;
                ds[kt][qt][r] = p * (dp[r] * kc + nD[r]);
              } else {
                pt[kt][qt][r] = p;
                ds[kt][qt][r] = p * dp[r];     // dp = dP - delta (accumulator start); 1/sqrt(d_h) is applied to dK / dQ at the end
              }
            }
          }
        }
        // ---- dV^T += dO^T P, dK^T += Q^T dS (contraction over the pair's 32 queries)
        u32x4 pf[2][1], dsf[2][1];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) { PackP<bf16_t, 2>::run(pt[kt], pf[kt]); PackP<bf16_t, 2>::run(ds[kt], dsf[kt]); }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const u32x4 gT = tfrag_tr(dOs, R::LDR, qp * 32, mt * 16, li, lg);
          const u32x4 qT = tfrag_tr(Qs, R::LDR, qp * 32, mt * 16, li, lg);
#pragma unroll
          for (int kt = 0; kt < 2; ++kt) {
            Mma<bf16_t>::mma(accV[kt][mt], gT, pf[kt][0]);
            Mma<bf16_t>::mma(accK[kt][mt], qT, dsf[kt][0]);
          }
        }
        // ---- dQ += dS K: dS through the [key][query] scratch (8-byte stores, transposing reads)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          bf16_t* st = sc + ((qp & 1) * 2 + qt) * GK * R::LDS2;
#pragma unroll
          for (int kt = 0; kt < 2; ++kt)
            *(u32x2*)&st[(kt * 16 + li) * R::LDS2 + 4 * lg] =
                (u32x2){pack_bf2(ds[kt][qt][0], ds[kt][qt][1]), pack_bf2(ds[kt][qt][2], ds[kt][qt][3])};
        }
        wave_lds_fence();
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          const u32x4 dsA = kmajor_frag(sc + ((qp & 1) * 2 + qt) * GK * R::LDS2, R::LDS2, 0, li, lg);   // row = query li, k = keys
          const int q0 = qp * 32 + qt * 16;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) Mma<bf16_t>::mma(accQ[qp * 2 + qt][mt], dsA, ktf[mt][0]);
          (void)q0;
        }
      }
      wave_lds_fence();   // the next group overwrites the K tile / scratch
    }
    // ---- dK, dV of this group: C-layout [dh 4 lg + r][key li] -> 4 consecutive channels of one key per lane
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const int key = g * GK + kt * 16 + li;
      if (key < d.Lk) {
        const long ko = koff + (long)(key * k_sl32);
        const long vo = voff + (long)(key * v_sl32);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const float ks_ = kmc[kt] ? 0.f : d.scale, vs_ = kmc[kt] ? 0.f : 1.f;   // padded keys: exact zeros
          u32x2* pk = (u32x2*)((bf16_t*)d.dk + ko + mt * 16 + 4 * lg);
          u32x2* pv = (u32x2*)((bf16_t*)d.dv + vo + mt * 16 + 4 * lg);
          float ak[4], av[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) { ak[r] = accK[kt][mt][r] * ks_; av[r] = accV[kt][mt][r] * vs_; }
          if (acc_kv) {   // second half of the queries: onto the first launch's rows
            const u32x2 ok_ = *pk, ov_ = *pv;
            ak[0] += __uint_as_float(ok_.x << 16); ak[1] += __uint_as_float(ok_.x & 0xffff0000u);
            ak[2] += __uint_as_float(ok_.y << 16); ak[3] += __uint_as_float(ok_.y & 0xffff0000u);
            av[0] += __uint_as_float(ov_.x << 16); av[1] += __uint_as_float(ov_.x & 0xffff0000u);
            av[2] += __uint_as_float(ov_.y << 16); av[3] += __uint_as_float(ov_.y & 0xffff0000u);
          }
          *pk = (u32x2){pack_bf2(ak[0], ak[1]), pack_bf2(ak[2], ak[3])};
          *pv = (u32x2){pack_bf2(av[0], av[1]), pack_bf2(av[2], av[3])};
        }
      }
    }
    g = gn;
    if (it < 5) RS_TL(3 + it);
  }

  // ---- dQ: sum of the 4 waves' register accumulators through LDS (no atomics), then out
  RS_TL(8);
  __syncthreads();
  RS_TL(9);   // every wave is done with the resident Q / dO tiles: the reduction buffer overlays them
  // two phases: the first half of the waves store into RW / 2 buffers (buffer 0 over the resident Q / dO tiles, the others
  // over the per-wave scratch, all dead now), the second half add onto them; the output loop below adds the buffers
  constexpr int NBUF = RW / 2;
  float* dQx = (float*)wv;   // buffers 1 .. NBUF - 1
  static_assert((NBUF - 1) * NQ * R::LDQ * 4 <= RW * WV_ELEMS * 2, "the extra dQ buffers must fit over the per-wave scratch");
  for (int ph = 0; ph < 2; ++ph) {
    if (wave / NBUF == ph) {
      const int bi = wave % NBUF;
      float* dst = bi == 0 ? dQs : dQx + (bi - 1) * NQ * R::LDQ;
#pragma unroll
      for (int t = 0; t < NQP * 2; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* p = &dst[(t * 16 + 4 * lg + r) * R::LDQ + mt * 16 + li];
            *p = (ph == 0 ? 0.f : *p) + accQ[t][mt][r] * d.scale;
          }
    }
    __syncthreads();
  }
  const long rows = (long)d.B * d.H * d.Lq;
  for (int i = tid; i < nq * (DH / 4); i += RW * 64) {
    const int q = i / (DH / 4), c0 = (i % (DH / 4)) * 4;
    float a[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      a[x] = dQs[q * R::LDQ + c0 + x];
#pragma unroll
      for (int bi = 1; bi < NBUF; ++bi) a[x] += dQx[(bi - 1) * NQ * R::LDQ + q * R::LDQ + c0 + x];
    }
    if (KS == 1) {
      const long off = (long)b * d.q_sb + (long)(q_lo + q) * d.q_sl + (long)h * d.q_sh + c0;
      *(u32x2*)((bf16_t*)d.dq + off) = (u32x2){pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3])};
    } else {
      float* po = d.ws + ((((long)split * d.B + b) * d.H + h) * d.Lq + q_lo + q) * DH + c0;
      *(float4*)po = make_float4(a[0], a[1], a[2], a[3]);
    }
  }
  (void)rows;
  RS_TL(10);
}

template <int DH, int NQP, int RW> size_t resident_lds(bool mask3) {   // mask3: the BYTE form (a staged tile per wave)
  typedef RT<DH> R;
  const size_t nq = NQP * 32;
  size_t b = 2 * nq * R::LDR * 2 + 2 * nq * 4;
  b += (size_t)RW * (R::KT_ELEMS + R::SC_ELEMS) * 2 + (mask3 ? (size_t)RW * 32 * R::MLD : 0);
  return b + nq + 16;
}

template <int DH, int NQP, bool DROP, int MASK3, int RW> void launch_res_w(const pq3d_attn_desc& d, hipStream_t s, int q_lo, int nq, int acc) {
  const int KS = d.ksplit > 1 ? d.ksplit : 1;
  const size_t lds = resident_lds<DH, NQP, RW>(MASK3 == 1);
  if constexpr (!DROP && !MASK3) {
    if (d.zero_attn) {
      auto kz = attn_bwd_resident_kernel<DH, NQP, DROP, MASK3, RW, true>;
      static std::atomic<unsigned> done_z{0};
      if (pq3d_enable_big_lds(kz, 160 * 1024, done_z)) { (void)hipGetLastError(); }
      hipLaunchKernelGGL(kz, dim3(KS, d.H, d.B), dim3(RW * 64), lds, s, d, q_lo, nq, acc);
      return;
    }
  }
  auto kern = attn_bwd_resident_kernel<DH, NQP, DROP, MASK3, RW>;
  static std::atomic<unsigned> attr_done{0};   // > 64 KB of dynamic LDS: opt-in once per (kernel instantiation, device)
  if (pq3d_enable_big_lds(kern, 160 * 1024, attr_done)) { (void)hipGetLastError(); }
  hipLaunchKernelGGL(kern, dim3(KS, d.H, d.B), dim3(RW * 64), lds, s, d, q_lo, nq, acc);
}

// 4 waves per workgroup; 8 (two per SIMD: the same latency hiding as two 4-wave workgroups of one (scene, head) on a CU,
// but Q / dO / O are loaded once and there are no dQ partials to combine) when the launch is not split over the keys and
// has at most one workgroup per CU -- config 2: 192 (scene, head) pairs unsplit; config 4: 96 pairs x 2 key slices
template <int DH, int NQP, bool DROP, int MASK3> void launch_res(const pq3d_attn_desc& d, hipStream_t s, int q_lo, int nq, int acc) {
  const int KS = d.ksplit > 1 ? d.ksplit : 1;
  if constexpr (DH == 32) {   // (d_h 64: the four dQ reduction buffers would not fit over the scratch)
    if ((long)d.B * d.H * KS <= 256 && d.Lk / KS >= 512) { launch_res_w<DH, NQP, DROP, MASK3, 8>(d, s, q_lo, nq, acc); return; }
  }
  launch_res_w<DH, NQP, DROP, MASK3, 4>(d, s, q_lo, nq, acc);
}

template <int DH, int NQP> void launch_res_flags(const pq3d_attn_desc& d, hipStream_t s, int q_lo, int nq, int acc) {
  const bool dr = d.drop.p > 0.f && d.drop.seed;
  const int m3 = d.mask != nullptr ? (d.mask_bits != nullptr ? 2 : 1) : 0;
  if (dr) {
    if (m3 == 2) launch_res<DH, NQP, true, 2>(d, s, q_lo, nq, acc);
    else if (m3 == 1) launch_res<DH, NQP, true, 1>(d, s, q_lo, nq, acc);
    else launch_res<DH, NQP, true, 0>(d, s, q_lo, nq, acc);
  } else {
    if (m3 == 2) launch_res<DH, NQP, false, 2>(d, s, q_lo, nq, acc);
    else if (m3 == 1) launch_res<DH, NQP, false, 1>(d, s, q_lo, nq, acc);
    else launch_res<DH, NQP, false, 0>(d, s, q_lo, nq, acc);
  }
}

template <int DH> void launch_res_rows(const pq3d_attn_desc& d, hipStream_t s, int q_lo, int nq, int acc) {
  if (nq <= 64) { launch_res_flags<DH, 2>(d, s, q_lo, nq, acc); return; }
#ifdef PQ3D_RES_NQP3   // probe builds: 65 .. 96 resident queries on 3 query pairs (the shipped stage-2 shape: 80 objects)
  if (nq <= 96) { launch_res_flags<DH, 3>(d, s, q_lo, nq, acc); return; }
#endif
  launch_res_flags<DH, 4>(d, s, q_lo, nq, acc);
}

template <int DH> bool launch_res_dh(const pq3d_attn_desc& d, hipStream_t s) {
  // up to 128 resident queries per workgroup (dQ accumulators: 64 registers per lane; 224 resident queries were built
  // and measured: 256 VGPRs, one wave per SIMD, 200 us = no better than the two-kernel path at config 4).  129 .. 256
  // queries: two launches over the halves of the rows, the second accumulating dK / dV (config 4: 2 x 100 queries)
  if (d.Lq <= 128) { launch_res_rows<DH>(d, s, 0, d.Lq, 0); return true; }
  if (d.Lq > 256) return false;
  const int h1 = (d.Lq + 1) / 2;
  launch_res_rows<DH>(d, s, 0, h1, 0);
  launch_res_rows<DH>(d, s, h1, d.Lq - h1, 1);
  return true;
}

// ------------------------------------------------------------------------------------------------ forward
// All-keys-resident forward for the same shape (N_q <= 128, d_h = 32, bf16, key-padding mask only).  The streaming
// forward of attention.hip walks 64-key tiles through a two-stage register/LDS pipeline with a workgroup barrier per
// tile; at 4 KB per tile the loop is bound by one global round trip per tile (24 us for 8 tiles at config 2), not by its
// 16 MFMAs.  Here the workgroup requests its WHOLE key/value slice (<= 1024 keys: 128 KB) up front -- every 16-byte load
// in flight at once --, parks it in LDS behind ONE barrier, and then each wave runs the online-softmax loop of its 16
// queries over all key blocks without any further synchronisation.  1024 keys per workgroup means config 2 needs no key
// split, so the combine launch goes away as well.  K rows are stored unpadded (64-byte rows: the 64 lanes of a fragment
// read cover 1 KB exactly once), V rows padded for the transposing reads.  Same arithmetic and rounding points as the
// streaming kernel (P and V in bf16, fp32 accumulation, exp2-free __expf); only the block order of the online softmax
// differs with the split count.
constexpr int FW = 8;            // compute waves = 16-query tiles
constexpr int FLW = 4;           // loader waves: bring in the second half of the slice while the first is multiplied
constexpr int FKMAX = 1024;      // keys per workgroup
constexpr int FK1 = FKMAX / 2;   // keys of stage 1 (loaded by the compute waves themselves)
constexpr int NB1 = FK1 / KB;
constexpr int FCH1 = FK1 * 4 / (FW * 64);    // 16-byte chunks of K (and of V) per compute thread, stage 1   (4)
constexpr int FCH2 = FK1 * 4 / (FLW * 64);   // ... per loader thread, stage 2                              (8)

template <bool DROP, bool TWO>
__global__ __launch_bounds__((FW + FLW) * 64) void attn_fwd_resident_kernel(const pq3d_attn_desc d) {
  ATTN_KARG_PIN(d);
  constexpr int DH = 32;
  typedef AT<bf16_t, DH> A;
  constexpr int LDK = DH, LDV = A::LDR;
  extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const WgXyz wg = attn_wg_xyz((int)gridDim.x);
  const int b = wg.b, h = wg.h, split = wg.x;
  const int KS = d.ksplit > 1 ? d.ksplit : 1;
  const int nkb = (d.Lk + KB - 1) / KB;
  const int kb_lo = (int)((long)nkb * split / KS), kb_hi = (int)((long)nkb * (split + 1) / KS);
  const int nb = kb_hi - kb_lo, key_lo = kb_lo * KB, nk = nb * KB;
  bf16_t* Ks = (bf16_t*)fsm;
  bf16_t* Vs = Ks + nk * LDK;
  uint8_t* kpm_s = (uint8_t*)(Vs + nk * LDV);
  const bool loader = wave >= FW;   // wave-uniform role
  const long koff = (long)b * d.k_sb + (long)h * d.k_sh, voff = (long)b * d.v_sb + (long)h * d.v_sh;
  // one 16-byte chunk of a K row and of the same V row: global -> registers (rows past the slice / past Lk: clamped
  // duplicates, never stored / masked through kpm_s)
  auto gload = [&](int c, u32x4& kx, u32x4& vx) {
    c = min(c, nk * 4 - 1);
    const int gk = min(key_lo + (c >> 2), d.Lk - 1), part = (c & 3) * 8;
    kx = *(const u32x4*)((const bf16_t*)d.k + koff + (long)gk * d.k_sl + part);
    vx = *(const u32x4*)((const bf16_t*)d.v + voff + (long)gk * d.v_sl + part);
  };
  auto park = [&](int c, const u32x4& kx, const u32x4& vx) {
    if (c < nk * 4) {
      *(u32x4*)&Ks[(c >> 2) * LDK + (c & 3) * 8] = kx;
      *(u32x4*)&Vs[(c >> 2) * LDV + (c & 3) * 8] = vx;
    }
  };
  const int q0 = wave * 16, myq = q0 + li;
  const bool wave_active = !loader && q0 < d.Lq, qvalid = !loader && myq < d.Lq;
  u32x4 qf[A::NS];
  // The load counter is per wave and in-order, and the compiler waits conservatively at loop headers: the waves that
  // multiply must not have loads in flight.  So the second half of the slice belongs to 4 LOADER waves that only fetch,
  // park and synchronise; the 8 compute waves fetch the first half (and the mask bytes and their Q fragments).
  if (!loader) {
    row_frags<bf16_t, DH>(qf, d.q, (long)b * d.q_sb + (long)min(myq, d.Lq - 1) * d.q_sl + (long)h * d.q_sh, lg);
    uint8_t kp[FKMAX / (FW * 64)];
#pragma unroll
    for (int i = 0; i < FKMAX / (FW * 64); ++i) {
      const int j = tid + i * FW * 64;
      kp[i] = (j < nk && key_lo + j < d.Lk) ? (d.kpm ? d.kpm[(long)b * d.Lk + key_lo + j] : 0) : 1;
    }
    u32x4 kr[FCH1], vr[FCH1];
#pragma unroll
    for (int i = 0; i < FCH1; ++i) gload(tid + i * FW * 64, kr[i], vr[i]);
#pragma unroll
    for (int i = 0; i < FKMAX / (FW * 64); ++i) {
      const int j = tid + i * FW * 64;
      if (j < nk) kpm_s[j] = kp[i];
    }
#pragma unroll
    for (int i = 0; i < FCH1; ++i) park(tid + i * FW * 64, kr[i], vr[i]);
  }
  u32x4 kr2[FCH2], vr2[FCH2];
  const int t2 = tid - FW * 64;
  if (loader && nb > NB1) {
#pragma unroll
    for (int i = 0; i < FCH2; ++i) gload(FK1 * 4 + t2 + i * FLW * 64, kr2[i], vr2[i]);
  }

  // online softmax in the base-2 domain: x2 = s * (scale * log2 e), p = 2^(x2 - m2) -- one multiply folded into the
  // scale, v_exp_f32 is a base-2 exponential
  const float sc2 = d.scale * 1.44269504088896341f;
  const bool zero0 = d.zero_attn && split == 0;   // the zero key (add_zero_attn) is the initial state of split 0 only
  float m = zero0 ? 0.f : -1e30f, l = (zero0 && lg == 0) ? 1.f : 0.f;
  f32x4 acc[A::MT];
#pragma unroll
  for (int mt = 0; mt < A::MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  DropState dst;
  uint32_t drow = 0;
  if constexpr (DROP) {
    dst = drop_init(d.drop, d.drop_bmod > 0 ? b / d.drop_bmod : 0, d.Lk);
    drow = (uint32_t)(((long)(d.drop_bmod > 0 ? b % d.drop_bmod : b) * d.H + h) * d.Lq + min(myq, d.Lq - 1));
  }
  auto block = [&](int t) __attribute__((always_inline)) {
    const bf16_t* Kt = Ks + t * KB * LDK;
    const bf16_t* Vt = Vs + t * KB * LDV;
    uint32_t mw[4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) mw[tt] = *(const uint32_t*)&kpm_s[t * KB + tt * 16 + 4 * lg];
    // a fully padded block contributes nothing (the 4 lane groups together hold all 64 key bytes)
    if (__all((mw[0] & mw[1] & mw[2] & mw[3]) == 0x01010101u)) return;
    // every LDS read of the block is issued before the first MFMA (K fragments, then the transposed V fragments that
    // are only needed after the softmax)
    u32x4 kf[4], vf[A::MT][2];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) kf[tt] = rfrag<bf16_t>(&Kt[(tt * 16 + li) * LDK], 0, lg);
#pragma unroll
    for (int mt = 0; mt < A::MT; ++mt)
#pragma unroll
      for (int u = 0; u < 2; ++u) vf[mt][u] = tfrag_tr(Vt, LDV, u * 32, mt * 16, li, lg);
    f32x4 sc[4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      sc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      Mma<bf16_t>::mma(sc[tt], kf[tt], qf[0]);
    }
    if (!__all((mw[0] | mw[1] | mw[2] | mw[3]) == 0u)) {   // padded keys in the block (the tail only): raw score -> -inf
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[tt][r] = ((mw[tt] >> (8 * r)) & 0xffu) ? -INFINITY : sc[tt][r];
    }
    // max over the RAW scores (sc2 > 0 commutes with max), then one fused multiply-subtract + one base-2 exponential
    // per element
    float mx = -INFINITY;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) mx = fmaxf(fmaxf(mx, fmaxf(sc[tt][0], sc[tt][1])), fmaxf(sc[tt][2], sc[tt][3]));
    mx = group_max(mx) * sc2;
    const float m_new = fmaxf(m, mx);
    const float alpha = __builtin_amdgcn_exp2f(m - m_new);
    float p[4][4];
    float rs0 = 0.f, rs1 = 0.f;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) p[tt][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[tt][r], sc2, -m_new));
      rs0 += p[tt][0] + p[tt][1];
      rs1 += p[tt][2] + p[tt][3];
    }
    l = l * alpha + (rs0 + rs1);   // per-lane partial (alpha is uniform over the query's 4 lanes)
    m = m_new;
    if constexpr (DROP) {   // the softmax denominator keeps every key; only the value contraction sees the mask
      const int k0 = key_lo + t * KB;
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const uint32_t cp = (uint32_t)(k0 + tt * 16 + 4 * lg) >> 1;
        const uint32_t w0 = drop_word(dst, drow, cp), w1 = drop_word(dst, drow, cp + 1);
        p[tt][0] = drop_keep_lo(dst, w0) ? p[tt][0] : 0.f;
        p[tt][1] = drop_keep_hi(dst, w0) ? p[tt][1] : 0.f;
        p[tt][2] = drop_keep_lo(dst, w1) ? p[tt][2] : 0.f;
        p[tt][3] = drop_keep_hi(dst, w1) ? p[tt][3] : 0.f;
      }
    }
    u32x4 pf[2];
    PackP<bf16_t, 4>::run(p, pf);
#pragma unroll
    for (int mt = 0; mt < A::MT; ++mt) {
      acc[mt] *= alpha;
#pragma unroll
      for (int u = 0; u < 2; ++u) Mma<bf16_t>::mma(acc[mt], vf[mt][u], pf[u]);
    }
  };
  // Two unpadded blocks as ONE online-softmax step (128 keys): a block is a single dependent chain (MFMA -> max -> cross-lane
  // max -> exp -> pack -> MFMA) and a SIMD holds 3 waves, so twice the independent work per chain step hides more of its latency
  // (8 back-to-back score MFMAs, 32 independent exponentials per lane).  Blocks with padded keys (the tail) take block().
  auto block2 = [&](int t) __attribute__((always_inline)) {
    const bf16_t* Kt = Ks + t * KB * LDK;
    const bf16_t* Vt = Vs + t * KB * LDV;
    uint32_t any = 0;
#pragma unroll
    for (int tt = 0; tt < 8; ++tt) any |= *(const uint32_t*)&kpm_s[t * KB + tt * 16 + 4 * lg];
    if (!__all(any == 0u)) { block(t); block(t + 1); return; }
    f32x4 sc[8];
    {
      u32x4 kf[8];
#pragma unroll
      for (int tt = 0; tt < 8; ++tt) kf[tt] = rfrag<bf16_t>(&Kt[(tt * 16 + li) * LDK], 0, lg);
#pragma unroll
      for (int tt = 0; tt < 8; ++tt) {
        sc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        Mma<bf16_t>::mma(sc[tt], kf[tt], qf[0]);
      }
    }
    u32x4 vf[A::MT][4];
#pragma unroll
    for (int mt = 0; mt < A::MT; ++mt)
#pragma unroll
      for (int u = 0; u < 4; ++u) vf[mt][u] = tfrag_tr(Vt + (u >> 1) * KB * LDV, LDV, (u & 1) * 32, mt * 16, li, lg);
    float mx = -INFINITY;
#pragma unroll
    for (int tt = 0; tt < 8; ++tt) mx = fmaxf(fmaxf(mx, fmaxf(sc[tt][0], sc[tt][1])), fmaxf(sc[tt][2], sc[tt][3]));
    mx = group_max(mx) * sc2;
    const float m_new = fmaxf(m, mx);
    const float alpha = __builtin_amdgcn_exp2f(m - m_new);
    float p[8][4];
    float rs0 = 0.f, rs1 = 0.f;
#pragma unroll
    for (int tt = 0; tt < 8; ++tt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) p[tt][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[tt][r], sc2, -m_new));
      rs0 += p[tt][0] + p[tt][1];
      rs1 += p[tt][2] + p[tt][3];
    }
    l = l * alpha + (rs0 + rs1);
    m = m_new;
    u32x4 pf[4];
    PackP<bf16_t, 8>::run(p, pf);
#pragma unroll
    for (int mt = 0; mt < A::MT; ++mt) {
      acc[mt] *= alpha;
#pragma unroll
      for (int u = 0; u < 4; ++u) Mma<bf16_t>::mma(acc[mt], vf[mt][u], pf[u]);
    }
  };
  auto run_blocks = [&](int t0, int t1) __attribute__((always_inline)) {
    int t = t0;
    if constexpr (TWO) {
      for (; t + 1 < t1; t += 2) block2(t);
    }
    for (; t < t1; ++t) block(t);
  };
  // two stages: the first 512 keys are multiplied while the loader waves bring in the rest of the slice
  __syncthreads();
  if (wave_active) run_blocks(0, min(nb, NB1));
  if (loader && nb > NB1) {
#pragma unroll
    for (int i = 0; i < FCH2; ++i) park(FK1 * 4 + t2 + i * FLW * 64, kr2[i], vr2[i]);
  }
  __syncthreads();
  if (wave_active) run_blocks(NB1, nb);
  l = group_sum(l);
  if (!qvalid) return;
  constexpr float LN2 = 0.693147180559945309f;
  if (KS == 1) {
    const float inv = (DROP ? dst.scale : 1.f) / l;
    bf16_t* op = (bf16_t*)d.o + (long)b * d.o_sb + (long)myq * d.o_sl + (long)h * d.o_sh + 4 * lg;
#pragma unroll
    for (int mt = 0; mt < A::MT; ++mt)
      *(u32x2*)(op + mt * 16) = (u32x2){pack_bf2(acc[mt][0] * inv, acc[mt][1] * inv), pack_bf2(acc[mt][2] * inv, acc[mt][3] * inv)};
    if (lg == 0) d.lse[((long)b * d.H + h) * d.Lq + myq] = m * LN2 + logf(l);
  } else {   // partial softmax state for attention.hip's combine kernel (same layout as the streaming forward)
    const long rows = (long)d.B * d.H * d.Lq, ridx = (((long)split * d.B + b) * d.H + h) * d.Lq + myq;
    float* po = d.ws + ridx * DH;
#pragma unroll
    for (int mt = 0; mt < A::MT; ++mt)
      *(float4*)(po + mt * 16 + 4 * lg) = make_float4(acc[mt][0], acc[mt][1], acc[mt][2], acc[mt][3]);
    if (lg == 0) {
      d.ws[(long)KS * rows * DH + ridx] = m * LN2;   // natural-log domain, as the combine kernel expects
      d.ws[(long)KS * rows * (DH + 1) + ridx] = l;
    }
  }
}

template <bool DROP, bool TWO> void launch_fwd_res(const pq3d_attn_desc& d, hipStream_t s, int KS, int nk_max) {
  const size_t lds = (size_t)nk_max * (32 * 2 + AT<bf16_t, 32>::LDR * 2 + 1) + 16;
  auto kern = attn_fwd_resident_kernel<DROP, TWO>;
  static std::atomic<unsigned> attr_done{0};   // > 64 KB of dynamic LDS: opt-in once per (kernel, device), to the CU's whole LDS
  if (pq3d_enable_big_lds(kern, 160 * 1024, attr_done)) { (void)hipGetLastError(); }
  hipLaunchKernelGGL(kern, dim3(KS, d.H, d.B), dim3((FW + FLW) * 64), lds, s, d);
}

}  // namespace

// Forward twin of the resident backward: bf16, d_h = 32, N_q <= 128, key-padding mask only, at most 1024 keys per
// (scene, head, split) workgroup.  Returns false otherwise (attention.hip's streaming forward); for ksplit > 1 the caller
// launches the combine kernel exactly as for the streaming kernel.
bool pq3d_resfwd_split_allowed();   // attention.hip: pq3d_attn_resident bit 5
bool pq3d_attn_fwd_resident_try(const pq3d_attn_desc& d, hipStream_t s) {
  if (d.ct != PQ3D_BF16 || d.dt != PQ3D_BF16 || d.bias || d.mask || d.dh != 32) return false;
  if (d.Lq > 128 || d.Lk < 128) return false;
  if ((d.q_sl & 7) || (d.k_sl & 7) || (d.v_sl & 7) || (d.o_sl & 3) || (d.q_sb & 7) || (d.k_sb & 7) || (d.v_sb & 7) || (d.o_sb & 3))
    return false;
  const int KS = d.ksplit > 1 ? d.ksplit : 1, nkb = (d.Lk + KB - 1) / KB;
  const int nb_max = (nkb + KS - 1) / KS;   // ceil: the largest slice
  if (nb_max * KB > FKMAX) return false;
  // key-split calls of longer scenes (config 5: 2048 keys as two slices of 1024): measured in isolation in round 2 the streaming
  // kernel was as fast (58 us against 63-76 us); in the step it is 75 + 6 us against this kernel, and the step says 6.535 ->
  // 6.463 ms with the slices here (tools/probes/ab_resfwd_split.sh) -- on by default since round 4, switch: pq3d_attn_resident bit 5
  if (KS > 1 && d.Lk > FKMAX && !pq3d_resfwd_split_allowed()) return false;
  // PQ3D_RESFWD_ONE=1: one 64-key block per online-softmax step (the form before round 5's two-block step; A/B measurements)
  static const bool one = [] { const char* e = getenv("PQ3D_RESFWD_ONE"); return e && atoi(e) != 0; }();
  if (d.drop.p > 0.f && d.drop.seed) launch_fwd_res<true, false>(d, s, KS, nb_max * KB);
  else if (one) launch_fwd_res<false, false>(d, s, KS, nb_max * KB);
  else launch_fwd_res<false, true>(d, s, KS, nb_max * KB);
  return true;
}

// Runs the all-queries-resident backward when the call has its shape (bf16, d_h 32 / 64, no additive bias, N_q <= 224,
// enough keys to be worth it); returns false for everything else (attention.hip's two-kernel path).  The caller launches
// the dQ combine kernel for ksplit > 1 exactly as for the two-kernel path.
bool pq3d_attn_bwd_resident_try(const pq3d_attn_desc& d, hipStream_t s) {
  if (d.ct != PQ3D_BF16 || d.bias || d.dbias) return false;
  if (d.dh != 32 && d.dh != 64) return false;
#ifndef PQ3D_RES_MIN_LK
#define PQ3D_RES_MIN_LK 128
#endif
  // d_h = 64 from 64 keys on: the shipped stage-2 cross-attention (80 objects per memory) is 342 -> 262 us per call here
  // against the two-kernel path (tools/probes/attn_s2_probe.py); at 32 keys (prompt tokens) the two-kernel path wins
  if (d.Lq > 256 || d.Lk < (d.dh == 64 ? (PQ3D_RES_MIN_LK < 64 ? PQ3D_RES_MIN_LK : 64) : PQ3D_RES_MIN_LK)) return false;
  if ((long)d.Lk * d.k_sl >= (1L << 31) || (long)d.Lk * d.v_sl >= (1L << 31)) return false;   // 32-bit row offsets inside a scene
  if (d.mask && ((d.Lk & 15) != 0 || (((uintptr_t)d.mask) & 15) != 0)) return false;
  if (d.dh == 32) return launch_res_dh<32>(d, s);
  return launch_res_dh<64>(d, s);
}
