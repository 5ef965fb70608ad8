// Split-bf16 cross-attention forward (compute mode 'bf16x3'): fp32-GRADE scores and value contraction on the bf16 matrix cores.
//
// Reference arithmetic: CrossAttentionLayer.forward_post (query_encoder.py:288-307) -> nn.MultiheadAttention with add_zero_attn
// (:268-270): softmax(q k^T / sqrt(d_h) [+ masks, + the zero key]) v in fp32.  The 'bf16' mode rounds K, V, Q, P and O to bf16
// (five sites of 0.8e-3 .. 1.5e-3 of the query scale each after 4 layers: profiles/rounding_sites_r05.txt), which is what keeps it
// at 3e-3 .. 7e-3 end to end.  Here every one of them is carried as a hi + lo bf16 pair:
//   * K and V arrive as two bf16 PLANES (hi = bf16(x), lo = bf16(x - hi): pq3d_gemm's PQ3D_ACT_PLANES epilogue) -- the bytes of an
//     fp32 tensor, but each plane is an MFMA operand as stored (no conversion in the loop, and the backward reads the hi plane
//     alone = exactly a 'bf16'-mode tensor);
//   * q (fp32) is split once into register fragments, P (fp32 registers) per 64-key block;
//   * S = K_lo q_hi + K_hi q_lo + K_hi q_hi and O += V_lo P_hi + V_hi P_lo + V_hi P_hi (small terms first): 3 MFMAs per product,
//     ~2^-17 relative per term; O leaves in fp32 (+ a bf16 copy for the backward's delta / weight gradient).
// Structure = attn_resident.hip's all-keys-resident forward: 8 compute waves (16 queries each) + 4 loader waves, the key/value
// slice parked in LDS behind one barrier, online softmax per wave without further synchronisation.  Two planes double the LDS per
// key (289 B), so a workgroup holds 512 keys at a time: the compute waves fetch and park the first 512, the loader waves hold
// the second 512 in registers meanwhile and park them over the first half once it is consumed (<= 1024 keys per key split as
// before: config 2 needs no split / combine launch).
// d_h = 64 (the head width of the reference's shipped decoders: hidden 768, 12 heads, configs/instseg_sceneverse.yaml:95,141) runs the
// same kernel with 256 keys per stage (both planes padded to 144-byte rows), two k steps per score tile and the K / V fragments
// fetched per tile instead of per block (168 registers per wave): shipped stage-1 / stage-2 decoder steps 14.7 / 15.4 ms in
// 'bf16x3' against 49.8 / 52.6 ms on the exact-f32 kernels the mode fell back to before, 13.6 / 13.7 ms in 'bf16'.
// New against the bf16 kernel: a 3-D self-mask as BIT WORDS (pq3d_mask_pack: row_open folded in) staged per 512-key stage, and up
// to 256 queries as two query halves in grid.x -- config 4's shape (200 queries, 4096 keys, self-mask) stays on this kernel.
#include <atomic>

#include "attn_common.h"

namespace {

constexpr int XW = 8;             // compute waves = 16-query tiles
constexpr int XLW = 4;            // loader waves
// per head size: keys resident at a time (16 KB of one K plane), 64-key blocks per stage, LDS row strides.  d_h = 32: K rows unpadded
// (64 B: the 289 B per key that let 512 keys stay); d_h = 64: both planes padded to 72 elements (144 B rows: 16 lanes of a fragment
// read cover 16 distinct 4-bank groups; unpadded 128 B rows would be 8-way conflicts), 256 keys per stage.
template <int DH> struct X3 {
  static constexpr int XK = 16384 / DH;              // 512 / 256
  static constexpr int XNB = XK / KB;                // 8 / 4
  static constexpr int CPR = DH / 8;                 // 16-byte chunks per key row and plane
  static constexpr int XCH1 = XK * CPR / (XW * 64);  // chunks of one plane per compute thread, stage 1 (4)
  static constexpr int XCH2 = XK * CPR / (XLW * 64); // ... per loader thread, stage 2 (8)
  static constexpr int MWR = 2 * XNB;                // mask words per query row and stage
  static constexpr int XMW = MWR + 1;                // + 1: conflict-free column reads
  static constexpr int MW1 = MWR * 128 / (XW * 64);  // mask words per compute thread (stage 1) / loader thread (stage 2)
  static constexpr int MW2 = MWR * 128 / (XLW * 64);
  static constexpr int LDK = DH == 32 ? 32 : AT<bf16_t, DH>::LDR;
  static constexpr int LDV = AT<bf16_t, DH>::LDR;
  static constexpr int KSD = DH / 32;                // k steps of a product over d_h
};

PQ_DEV void x3_split(const float* v, u32x4& hi, u32x4& lo) {
  hi = pack_frag<bf16_t>(v);
  float w[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    w[2 * j] = v[2 * j] - __uint_as_float(hi[j] << 16);
    w[2 * j + 1] = v[2 * j + 1] - __uint_as_float(hi[j] & 0xffff0000u);
  }
  lo = pack_frag<bf16_t>(w);
}

template <int DH, bool DROP, bool MASKB>
__global__ __launch_bounds__((XW + XLW) * 64) void attn_fwd_x3_kernel(const pq3d_attn_desc d, const int KS, const int per) {
  ATTN_KARG_PIN(d);
  typedef AT<bf16_t, DH> A;
  typedef X3<DH> X;
  constexpr int LDK = X::LDK, LDV = X::LDV, XK = X::XK, XNB = X::XNB, CPR = X::CPR, XCH1 = X::XCH1, XCH2 = X::XCH2, XMW = X::XMW,
                MWR = X::MWR, KSD = X::KSD;
  extern __shared__ __attribute__((aligned(16))) unsigned char xsm[];
  bf16_t* const Kh = (bf16_t*)xsm;
  bf16_t* const Kl = Kh + XK * LDK;
  bf16_t* const Vh = Kl + XK * LDK;
  bf16_t* const Vl = Vh + XK * LDV;
  uint8_t* const kpm_s = (uint8_t*)(Vl + XK * LDV);          // [2 XK] key padding of the whole slice
  uint32_t* const msk_s = (uint32_t*)(kpm_s + 2 * XK);       // [128][XMW] mask bit words of the current stage (MASKB)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lg = lane >> 4;
  const WgXyz wg = attn_wg_xyz(KS);
  const int b = wg.b, h = wg.h, split = wg.x % KS, qh = wg.x / KS;
  const int nkb = (d.Lk + KB - 1) / KB;
  const int kb_lo = (int)((long)nkb * split / KS), kb_hi = (int)((long)nkb * (split + 1) / KS);
  const int nb = kb_hi - kb_lo, key_lo = kb_lo * KB, nk = nb * KB;
  const bool loader = wave >= XW;   // wave-uniform role
  const long koff = (long)b * d.k_sb + (long)h * d.k_sh, voff = (long)b * d.v_sb + (long)h * d.v_sh;
  const int q_base = qh * per, q_end = min(q_base + per, d.Lq);
  const int q0 = q_base + wave * 16, myq = q0 + li;
  const bool wave_active = !loader && q0 < q_end, qvalid = !loader && myq < q_end;
  const int bm = d.mask_bmod > 0 ? b % d.mask_bmod : b;
  const int nwords = (d.Lk + 31) / 32;

  // one 16-byte chunk of a key row in all four planes: global -> registers (rows past the slice / past Lk: clamped duplicates,
  // never parked / masked through kpm_s)
  auto gload = [&](int stage, int c, u32x4& kh, u32x4& kl, u32x4& vh, u32x4& vl) {
    const int gk = min(key_lo + stage * XK + c / CPR, d.Lk - 1), part = (c % CPR) * 8;
    const long ko = koff + (long)gk * d.k_sl + part, vo = voff + (long)gk * d.v_sl + part;
    kh = *(const u32x4*)((const bf16_t*)d.k + ko);
    kl = *(const u32x4*)((const bf16_t*)d.k_lo + ko);
    vh = *(const u32x4*)((const bf16_t*)d.v + vo);
    vl = *(const u32x4*)((const bf16_t*)d.v_lo + vo);
  };
  auto park = [&](int c, const u32x4& kh, const u32x4& kl, const u32x4& vh, const u32x4& vl) {
    *(u32x4*)&Kh[(c / CPR) * LDK + (c % CPR) * 8] = kh;
    *(u32x4*)&Kl[(c / CPR) * LDK + (c % CPR) * 8] = kl;
    *(u32x4*)&Vh[(c / CPR) * LDV + (c % CPR) * 8] = vh;
    *(u32x4*)&Vl[(c / CPR) * LDV + (c % CPR) * 8] = vl;
  };
  // the stage's mask words of the workgroup's 128 query rows: word (q, w) = bits[bm, q_base + q, 2 (kb_lo + 8 stage) + w]
  auto mask_words = [&](int stage, int t, int nthreads, uint32_t (&mw)[X::MW2], int n) {
#pragma unroll
    for (int i = 0; i < n; ++i) {
      const int c = t + i * nthreads, q = c / MWR, w = c % MWR;
      const int gq = min(q_base + q, d.Lq - 1), gw = 2 * (kb_lo + XNB * stage) + w;
      mw[i] = gw < nwords ? d.mask_bits[((long)bm * d.Lq + gq) * nwords + gw] : 0u;
    }
  };
  auto mask_park = [&](int t, int nthreads, const uint32_t (&mw)[X::MW2], int n) {
#pragma unroll
    for (int i = 0; i < n; ++i) {
      const int c = t + i * nthreads;
      msk_s[(c / MWR) * XMW + (c % MWR)] = mw[i];
    }
  };

  // ---- loader waves: a code path of their own behind a SCALAR branch (`wave` is read through readfirstlane), so that the 128
  // registers that hold the second half of the slice are not live across the compute waves' code; same barrier sequence
  if (loader) {
    u32x4 kh2[XCH2], kl2[XCH2], vh2[XCH2], vl2[XCH2];
    uint32_t mw2[X::MW2];
    const int t2 = tid - XW * 64;
    if (nb > XNB) {
#pragma unroll
      for (int i = 0; i < XCH2; ++i) gload(1, t2 + i * XLW * 64, kh2[i], kl2[i], vh2[i], vl2[i]);
      if constexpr (MASKB) mask_words(1, t2, XLW * 64, mw2, X::MW2);
    }
    __syncthreads();   // (A) stage 1 parked
    if (nb > XNB) {
      __syncthreads();   // (B) every wave is done reading stage 1
#pragma unroll
      for (int i = 0; i < XCH2; ++i) {
        const int c = t2 + i * XLW * 64;
        if (c < (nk - XK) * CPR) park(c, kh2[i], kl2[i], vh2[i], vl2[i]);
      }
      if constexpr (MASKB) mask_park(t2, XLW * 64, mw2, X::MW2);
      __syncthreads();   // (C) stage 2 parked
    }
    return;
  }
  // ---- compute waves
  u32x4 qfh[KSD], qfl[KSD];
  {
#pragma unroll
    for (int ks = 0; ks < KSD; ++ks) {   // q row (fp32) -> hi / lo fragments: lane (i, g) holds k = 32 ks + 8 g .. + 7 of query i
      const long qo = (long)b * d.q_sb + (long)min(myq, d.Lq - 1) * d.q_sl + (long)h * d.q_sh + 32 * ks + 8 * lg;
      const float4 a0 = *(const float4*)((const float*)d.q + qo), a1 = *(const float4*)((const float*)d.q + qo + 4);
      const float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      x3_split(v, qfh[ks], qfl[ks]);
      if (d.q_bf && split == 0 && qvalid) *(u32x4*)((bf16_t*)d.q_bf + qo) = qfh[ks];
    }
    uint8_t kp[2 * XK / (XW * 64)];
#pragma unroll
    for (int i = 0; i < 2 * XK / (XW * 64); ++i) {
      const int j = tid + i * XW * 64;
      kp[i] = (j < nk && key_lo + j < d.Lk) ? (d.kpm ? d.kpm[(long)b * d.Lk + key_lo + j] : 0) : 1;
    }
    uint32_t mw[X::MW2];
    if constexpr (MASKB) mask_words(0, tid, XW * 64, mw, X::MW1);
    u32x4 kh[XCH1], kl[XCH1], vh[XCH1], vl[XCH1];
#pragma unroll
    for (int i = 0; i < XCH1; ++i) gload(0, tid + i * XW * 64, kh[i], kl[i], vh[i], vl[i]);
#pragma unroll
    for (int i = 0; i < 2 * XK / (XW * 64); ++i) kpm_s[tid + i * XW * 64] = kp[i];
    if constexpr (MASKB) mask_park(tid, XW * 64, mw, X::MW1);
#pragma unroll
    for (int i = 0; i < XCH1; ++i) park(tid + i * XW * 64, kh[i], kl[i], vh[i], vl[i]);
  }

  // online softmax in the base-2 domain (attn_resident.hip): x2 = s * (scale * log2 e), p = 2^(x2 - m2)
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
  const int qrow = (wave * 16 + li) * XMW;   // this lane's row of the staged mask words
  // t: block inside the resident stage, tg: block inside the key slice
  auto block = [&](int t, int tg) __attribute__((always_inline)) {
    uint32_t mwb[4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) mwb[tt] = *(const uint32_t*)&kpm_s[tg * KB + tt * 16 + 4 * lg];
    uint32_t nib[4] = {0u, 0u, 0u, 0u};   // MASKB: the lane's 4 keys of tile tt as a nibble (bit r = key 16 tt + 4 lg + r masked)
    if constexpr (MASKB) {
      const uint32_t w0 = msk_s[qrow + 2 * t], w1 = msk_s[qrow + 2 * t + 1];
      nib[0] = (w0 >> (4 * lg)) & 0xFu; nib[1] = (w0 >> (16 + 4 * lg)) & 0xFu;
      nib[2] = (w1 >> (4 * lg)) & 0xFu; nib[3] = (w1 >> (16 + 4 * lg)) & 0xFu;
      // a block no query of the wave attends to contributes nothing
      bool dead = true;
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const uint32_t k4 = ((mwb[tt] & 1u) | ((mwb[tt] >> 7) & 2u) | ((mwb[tt] >> 14) & 4u) | ((mwb[tt] >> 21) & 8u));
        nib[tt] |= k4;
        dead = dead && nib[tt] == 0xFu;
      }
      if (__all((int)dead)) return;
    } else {
      if (__all((mwb[0] & mwb[1] & mwb[2] & mwb[3]) == 0x01010101u)) return;   // fully padded block
    }
    const bf16_t* Kht = Kh + t * KB * LDK;
    const bf16_t* Klt = Kl + t * KB * LDK;
    const bf16_t* Vht = Vh + t * KB * LDV;
    const bf16_t* Vlt = Vl + t * KB * LDV;
    f32x4 sc[4];
    if constexpr (DH == 32) {
      u32x4 kfh[4], kfl[4];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        kfh[tt] = rfrag<bf16_t>(&Kht[(tt * 16 + li) * LDK], 0, lg);
        kfl[tt] = rfrag<bf16_t>(&Klt[(tt * 16 + li) * LDK], 0, lg);
      }
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        sc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        Mma<bf16_t>::mma(sc[tt], kfl[tt], qfh[0]);
        Mma<bf16_t>::mma(sc[tt], kfh[tt], qfl[0]);
        Mma<bf16_t>::mma(sc[tt], kfh[tt], qfh[0]);
      }
    } else {   // two k steps over d_h: fragments of one key tile at a time (register budget: 168 per wave)
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        sc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSD; ++ks) {
          const u32x4 kh_ = rfrag<bf16_t>(&Kht[(tt * 16 + li) * LDK], ks, lg), kl_ = rfrag<bf16_t>(&Klt[(tt * 16 + li) * LDK], ks, lg);
          Mma<bf16_t>::mma(sc[tt], kl_, qfh[ks]);
          Mma<bf16_t>::mma(sc[tt], kh_, qfl[ks]);
          Mma<bf16_t>::mma(sc[tt], kh_, qfh[ks]);
        }
      }
    }
    u32x4 vfh[DH == 32 ? A::MT : 1][2], vfl[DH == 32 ? A::MT : 1][2];
    if constexpr (DH == 32) {
#pragma unroll
      for (int mt = 0; mt < A::MT; ++mt)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          vfh[mt][u] = tfrag_tr(Vht, LDV, u * 32, mt * 16, li, lg);
          vfl[mt][u] = tfrag_tr(Vlt, LDV, u * 32, mt * 16, li, lg);
        }
    }
    if constexpr (MASKB) {
      if (!__all((nib[0] | nib[1] | nib[2] | nib[3]) == 0u)) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
          for (int r = 0; r < 4; ++r) sc[tt][r] = ((nib[tt] >> r) & 1u) ? -INFINITY : sc[tt][r];
      }
    } else {
      if (!__all((mwb[0] | mwb[1] | mwb[2] | mwb[3]) == 0u)) {   // padded keys in the block (the tail only): raw score -> -inf
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
          for (int r = 0; r < 4; ++r) sc[tt][r] = ((mwb[tt] >> (8 * r)) & 0xffu) ? -INFINITY : sc[tt][r];
      }
    }
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
      const int k0 = key_lo + tg * KB;
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
    // P as hi + lo bf16 fragments (PackP's layout: k index = tile rows)
    u32x4 pfh[2], pfl[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int tt = 2 * u + hf;
        const uint32_t h0 = pack_bf2(p[tt][0], p[tt][1]), h1 = pack_bf2(p[tt][2], p[tt][3]);
        pfh[u][2 * hf] = h0;
        pfh[u][2 * hf + 1] = h1;
        pfl[u][2 * hf] = pack_bf2(p[tt][0] - __uint_as_float(h0 << 16), p[tt][1] - __uint_as_float(h0 & 0xffff0000u));
        pfl[u][2 * hf + 1] = pack_bf2(p[tt][2] - __uint_as_float(h1 << 16), p[tt][3] - __uint_as_float(h1 & 0xffff0000u));
      }
#pragma unroll
    for (int mt = 0; mt < A::MT; ++mt) {
      acc[mt] *= alpha;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if constexpr (DH == 32) {
          Mma<bf16_t>::mma(acc[mt], vfl[mt][u], pfh[u]);
          Mma<bf16_t>::mma(acc[mt], vfh[mt][u], pfl[u]);
          Mma<bf16_t>::mma(acc[mt], vfh[mt][u], pfh[u]);
        } else {   // value fragments of one 16-column tile at a time
          const u32x4 vh_ = tfrag_tr(Vht, LDV, u * 32, mt * 16, li, lg), vl_ = tfrag_tr(Vlt, LDV, u * 32, mt * 16, li, lg);
          Mma<bf16_t>::mma(acc[mt], vl_, pfh[u]);
          Mma<bf16_t>::mma(acc[mt], vh_, pfl[u]);
          Mma<bf16_t>::mma(acc[mt], vh_, pfh[u]);
        }
      }
    }
  };
  __syncthreads();   // (A) stage 1 parked
  if (wave_active)
    for (int t = 0; t < min(nb, XNB); ++t) block(t, t);
  if (nb > XNB) {   // uniform over the workgroup
    __syncthreads();   // (B)
    __syncthreads();   // (C) the loader waves have parked stage 2 over stage 1
    if (wave_active)
      for (int t = XNB; t < nb; ++t) block(t - XNB, t);
  }
  l = group_sum(l);
  if (!qvalid) return;
  constexpr float LN2 = 0.693147180559945309f;
  if (KS == 1) {
    const float inv = (DROP ? dst.scale : 1.f) / l;
    const long oo = (long)b * d.o_sb + (long)myq * d.o_sl + (long)h * d.o_sh + 4 * lg;
#pragma unroll
    for (int mt = 0; mt < A::MT; ++mt) {
      const float4 v = make_float4(acc[mt][0] * inv, acc[mt][1] * inv, acc[mt][2] * inv, acc[mt][3] * inv);
      *(float4*)((float*)d.o + oo + mt * 16) = v;
      if (d.o_bf) *(u32x2*)((bf16_t*)d.o_bf + oo + mt * 16) = (u32x2){pack_bf2(v.x, v.y), pack_bf2(v.z, v.w)};
    }
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

template <int DH, bool DROP, bool MASKB> int launch_x3(const pq3d_attn_desc& d, hipStream_t s, int KS, int nqh, int per) {
  typedef X3<DH> X;
  const size_t lds = (size_t)X::XK * (2 * X::LDK * 2 + 2 * X::LDV * 2) + 2 * X::XK + (MASKB ? 128 * X::XMW * 4 : 0) + 16;
  static_assert((size_t)X::XK * (2 * X::LDK * 2 + 2 * X::LDV * 2) + 2 * X::XK + 128 * X::XMW * 4 + 16 <= 160 * 1024, "LDS");
  auto kern = attn_fwd_x3_kernel<DH, DROP, MASKB>;
  static std::atomic<unsigned> attr_done{0};   // > 64 KB of dynamic LDS: opt-in once per (kernel, device)
  if (int e = pq3d_enable_big_lds(kern, 160 * 1024, attr_done)) { pq3d_set_error(hipGetErrorString((hipError_t)e)); return e; }
  hipLaunchKernelGGL(kern, dim3(KS * nqh, d.H, d.B), dim3((XW + XLW) * 64), lds, s, d, KS, per);
  return 0;
}

}  // namespace

// pq3d_attn_fwd with k_lo set (attention.hip routes here; the caller launches the combine kernel for ksplit > 1)
int pq3d_attn_fwd_x3(const pq3d_attn_desc& d, hipStream_t s) {
  PQ_CHECK_ARG(d.dt == PQ3D_F32 && d.k_lo && d.v_lo, "pq3d_attn_fwd (split-bf16 planes): q / o must be fp32, k_lo and v_lo set");
  PQ_CHECK_ARG((d.dh == 32 || d.dh == 64) && !d.bias && d.Lq <= 256 && d.Lk >= 1,
               "pq3d_attn_fwd (split-bf16 planes): d_h = 32 / 64, <= 256 queries, no additive bias");
  PQ_CHECK_ARG(!d.mask || d.mask_bits, "pq3d_attn_fwd (split-bf16 planes): a 3-D mask must come as bit words (pq3d_mask_pack)");
  PQ_CHECK_ARG(!(d.mask && d.kpm), "pq3d_attn_fwd (split-bf16 planes): key padding and a 3-D mask are exclusive");
  PQ_CHECK_ARG(!((d.q_sl | d.q_sb | d.q_sh | d.o_sl | d.o_sb | d.o_sh) & 3) && !((d.k_sl | d.k_sb | d.k_sh | d.v_sl | d.v_sb | d.v_sh) & 7),
               "pq3d_attn_fwd (split-bf16 planes): strides must be multiples of 16 bytes");
  PQ_CHECK_ARG(((((uintptr_t)d.k_lo) | ((uintptr_t)d.v_lo) | ((uintptr_t)d.q_bf) | ((uintptr_t)d.o_bf)) & 15) == 0,
               "pq3d_attn_fwd (split-bf16 planes): planes must be 16-byte aligned");
  const int KS = d.ksplit > 1 ? d.ksplit : 1, nkb = (d.Lk + KB - 1) / KB;
  const int xk = d.dh == 32 ? X3<32>::XK : X3<64>::XK;
  PQ_CHECK_ARG(((nkb + KS - 1) / KS) * KB <= 2 * xk, "pq3d_attn_fwd (split-bf16 planes): at most 1024 (d_h = 64: 512) keys per key split");
  const int nqh = (d.Lq + 127) / 128, per = (d.Lq + nqh - 1) / nqh;
  const bool dr = d.drop.p > 0.f && d.drop.seed, mb = d.mask != nullptr;
  if (d.dh == 32) {
    if (dr) return mb ? launch_x3<32, true, true>(d, s, KS, nqh, per) : launch_x3<32, true, false>(d, s, KS, nqh, per);
    return mb ? launch_x3<32, false, true>(d, s, KS, nqh, per) : launch_x3<32, false, false>(d, s, KS, nqh, per);
  }
  if (dr) return mb ? launch_x3<64, true, true>(d, s, KS, nqh, per) : launch_x3<64, true, false>(d, s, KS, nqh, per);
  return mb ? launch_x3<64, false, true>(d, s, KS, nqh, per) : launch_x3<64, false, false>(d, s, KS, nqh, per);
}
