// Body of attn_ca.hip, compiled once per head size: CA_DH (32 / 64) is set by the including file, which wraps each inclusion
// in its own namespace.  See attn_ca.hip for the design; the tile idioms are those of attn_sa_body.h (transposed score
// tiles, the permuted key order of two 16-token C tiles as one 32-wide k step), on single bf16 planes.
constexpr int DH = CA_DH, LDH = DH + 8;   // LDS row: 40 (72) bf16 = 20 (36) dwords -> 16 rows hit 16 distinct 4-bank groups
constexpr int KS = DH / 32, OT = DH / 16;  // 32-wide k steps of a product over d_h; 16-row tiles of a [d_h][tokens] result
constexpr int CA_MAXT = 512;               // 8 waves: up to 128 queries / keys

PQ_DEV float xrow_sum(float v) {    // sum over the 4 lanes li, li + 16, li + 32, li + 48
  u32pair_c a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  u32pair_c b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
PQ_DEV float xrow_max(float v) {
  u32pair_c a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  u32pair_c b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// row-major fragment: token row `row`, d_h slots 8 lg .. 8 lg + 7 of k step ks (A or B operand of a product over d_h)
PQ_DEV u32x4 frag_rm(const bf16_t* pl, int row, int lg, int ks) { return *(const u32x4*)&pl[row * LDH + ks * 32 + lg * 8]; }
// acc += rows(plane)[row] . b over d_h
PQ_DEV void mmak(f32x4& acc, const bf16_t* pl, int row, int lg, const u32x4* b) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) Mma<bf16_t>::mma(acc, frag_rm(pl, row, lg, ks), b[ks]);
}
// transposed fragment (A operand, m = d_h index c0 + li, k = tokens): slots 0..3 = tokens t0 + 4 lg + 0..3, slots 4..7 =
// tokens t1 + 4 lg + 0..3 -- the token order in which two 16-token C tiles sit in a lane's registers
// has2 (wave-uniform): the second tile exists -- the planes hold whole 16-row tiles, not whole pairs (80 tokens = 5 tiles: 46 KB
// instead of 55 KB for the four planes of the shipped stage-2 shape, three workgroups per CU instead of two); an absent
// tile reads as zeros
PQ_DEV u32x4 frag_tr(const bf16_t* plane, int t0, int t1, int c0, int li, int lg, bool has2) {
  const bf16_t* p0 = plane + (t0 + 4 * lg + (li >> 2)) * LDH + c0 + 4 * (li & 3);
  const v4i16_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_t*)p0);
  const u32x2 x = __builtin_bit_cast(u32x2, a);
  u32x2 y = (u32x2){0, 0};
  if (has2) {
    const bf16_t* p1 = plane + (t1 + 4 * lg + (li >> 2)) * LDH + c0 + 4 * (li & 3);
    const v4i16_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_t*)p1);
    y = __builtin_bit_cast(u32x2, b);
  }
  return (u32x4){x.x, x.y, y.x, y.y};
}

// bf16 [L, DH] (token stride sl) -> plane [LP][LDH]; rows >= L are zero.  All loads of the thread in flight first.
template <int MAXC> PQ_DEV void stage_rows(bf16_t* pl, const bf16_t* src, long sl, int L, int LP, int tid, int nthr) {
  const int nch = LP * (DH / 8);
  for (int base = tid; base < nch; base += nthr * MAXC) {
    u32x4 a[MAXC];
#pragma unroll
    for (int u = 0; u < MAXC; ++u) {
      const int c = min(base + u * nthr, nch - 1), row = c / (DH / 8), x = (c % (DH / 8)) * 8;
      a[u] = *(const u32x4*)(src + (long)min(row, L - 1) * sl + x);
    }
#pragma unroll
    for (int u = 0; u < MAXC; ++u) {
      const int c = base + u * nthr, row = c / (DH / 8), x = (c % (DH / 8)) * 8;
      if (c < nch) *(u32x4*)&pl[row * LDH + x] = row < L ? a[u] : (u32x4){0, 0, 0, 0};
    }
  }
}

struct CaLds {
  bf16_t *Q, *K, *V, *G;
  float *kb, *lse, *dl;   // additive key term (0 / -inf) [LPk]; lse, delta [LPq]
};
// planes: TQ / TK rows (whole 16-row tiles); float rows: LPq / LPk entries (whole pairs of tiles)
PQ_DEV CaLds carve(unsigned char* sm, int TQ, int TK, int LPq, int LPk, bool bwd) {
  CaLds s;
  bf16_t* p = (bf16_t*)sm;
  s.Q = p; p += TQ * LDH;
  s.K = p; p += TK * LDH;
  s.V = p; p += TK * LDH;
  s.G = nullptr;
  if (bwd) { s.G = p; p += TQ * LDH; }
  s.kb = (float*)p;
  s.lse = s.kb + LPk;
  s.dl = s.lse + LPq;
  return s;
}
size_t ca_lds_bytes(int Lq, int Lk, bool bwd) {
  const size_t TQ = (Lq + 15) & ~15, TK = (Lk + 15) & ~15, LPq = (Lq + 31) & ~31, LPk = (Lk + 31) & ~31;
  return ((bwd ? 2 : 1) * TQ + 2 * TK) * LDH * sizeof(bf16_t) + (LPk + 2 * LPq) * sizeof(float) + 16;
}

// keep factors (0 / 1) of the 8 probabilities a lane holds for one row of the dropout site and the key columns c0 .. c0 + 3,
// c0 + 16 .. c0 + 19 (c0 % 4 == 0: two hash words per 4 columns)
PQ_DEV void drop_keep8(const DropState& dst, uint32_t row, int c0, float* kf) {
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const uint32_t cp = (uint32_t)(c0 + 16 * hh) >> 1;
    const uint32_t w0 = drop_word(dst, row, cp), w1 = drop_word(dst, row, cp + 1);
    kf[4 * hh + 0] = drop_keep_lo(dst, w0) ? 1.f : 0.f;
    kf[4 * hh + 1] = drop_keep_hi(dst, w0) ? 1.f : 0.f;
    kf[4 * hh + 2] = drop_keep_lo(dst, w1) ? 1.f : 0.f;
    kf[4 * hh + 3] = drop_keep_hi(dst, w1) ? 1.f : 0.f;
  }
}

template <bool DROP>
__global__ __launch_bounds__(CA_MAXT) void attn_ca_fwd_kernel(const pq3d_attn_desc d) {
  ATTN_KARG_PIN(d);
  extern __shared__ __attribute__((aligned(16))) unsigned char ca_sm[];
  const int Lq = d.Lq, Lk = d.Lk, LPq = (Lq + 31) & ~31, LPk = (Lk + 31) & ~31, TQ = (Lq + 15) & ~15, TK = (Lk + 15) & ~15;
  const CaLds S = carve(ca_sm, TQ, TK, LPq, LPk, false);
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const int b = blockIdx.y, h = blockIdx.x;
  const bf16_t* q = (const bf16_t*)d.q + (long)b * d.q_sb + (long)h * d.q_sh;
  const bf16_t* k = (const bf16_t*)d.k + (long)b * d.k_sb + (long)h * d.k_sh;
  const bf16_t* v = (const bf16_t*)d.v + (long)b * d.v_sb + (long)h * d.v_sh;
  stage_rows<2>(S.Q, q, d.q_sl, Lq, TQ, tid, nthr);
  stage_rows<2>(S.K, k, d.k_sl, Lk, TK, tid, nthr);
  stage_rows<2>(S.V, v, d.v_sl, Lk, TK, tid, nthr);
  for (int j = tid; j < LPk; j += nthr) S.kb[j] = (j < Lk && !(d.kpm && d.kpm[(long)b * Lk + j])) ? 0.f : -INFINITY;
  __syncthreads();
  const int q0 = wave * 16;
  if (q0 >= TQ) return;
  const int qrow = q0 + li;                       // this lane's query (column of the transposed score tiles)
  u32x4 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qf[ks] = frag_rm(S.Q, qrow, lg, ks);
  // the zero key of add_zero_attn (logit exactly 0, value 0) is the initial state of the online softmax
  float m_run = d.zero_attn ? 0.f : -INFINITY, l_run = d.zero_attn ? 1.f : 0.f;
  DropState dst;
  uint32_t drow = 0;
  if constexpr (DROP) {   // attention-probability dropout: the site is the matrix [B H Lq, Lk] (pq3d_hip.h)
    dst = drop_init(d.drop, d.drop_bmod > 0 ? b / d.drop_bmod : 0, Lk);
    drow = (uint32_t)(((long)(d.drop_bmod > 0 ? b % d.drop_bmod : b) * d.H + h) * Lq + min(qrow, Lq - 1));
  }
  f32x4 ot[OT];   // O^T: tile t = rows d_h 16 t + 4 lg + r, column = query
#pragma unroll
  for (int t = 0; t < OT; ++t) ot[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int t0 = 0; t0 < LPk; t0 += 32) {
    const bool has2 = t0 + 16 < TK;               // uniform: the pair's second key tile exists (else: -inf through kb)
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    mmak(s0, S.K, t0 + li, lg, qf);               // S^T = K Q^T: lane = query column, 4 consecutive keys per tile
    if (has2) mmak(s1, S.K, t0 + 16 + li, lg, qf);
    const float4 kb0 = *(const float4*)&S.kb[t0 + 4 * lg], kb1 = *(const float4*)&S.kb[t0 + 16 + 4 * lg];
    const float kbv[8] = {kb0.x, kb0.y, kb0.z, kb0.w, kb1.x, kb1.y, kb1.z, kb1.w};
    float sv[8], mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sv[j] = (j < 4 ? s0[j] : s1[j - 4]) * d.scale + kbv[j];
      mx = fmaxf(mx, sv[j]);
    }
    mx = xrow_max(mx);
    const float m_new = fmaxf(m_run, mx), m_use = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = __expf(m_run - m_use);   // m_run = -inf -> 0
    float p[8], ps = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { p[j] = __expf(sv[j] - m_use); ps += p[j]; }
    l_run = l_run * alpha + xrow_sum(ps);
    m_run = m_new;
    if constexpr (DROP) {   // the softmax denominator keeps every key; only the value contraction sees the mask
      float kf[8];
      drop_keep8(dst, drow, t0 + 4 * lg, kf);
#pragma unroll
      for (int j = 0; j < 8; ++j) p[j] *= kf[j];
    }
    const u32x4 pf = pack_frag<bf16_t>(p);
#pragma unroll
    for (int t = 0; t < OT; ++t) {
      ot[t] *= alpha;
      Mma<bf16_t>::mma(ot[t], frag_tr(S.V, t0, t0 + 16, 16 * t, li, lg, has2), pf);   // O^T += V^T P^T
    }
  }
  if (qrow < Lq) {
    const float inv = l_run > 0.f ? (DROP ? dst.scale : 1.f) / l_run : 0.f;
    bf16_t* o = (bf16_t*)d.o + (long)b * d.o_sb + (long)h * d.o_sh + (long)qrow * d.o_sl;
#pragma unroll
    for (int t = 0; t < OT; ++t)
      *(u32x2*)(o + 16 * t + 4 * lg) = (u32x2){pack_bf2(ot[t][0] * inv, ot[t][1] * inv), pack_bf2(ot[t][2] * inv, ot[t][3] * inv)};
    if (lg == 0) d.lse[((long)b * d.H + h) * Lq + qrow] = l_run > 0.f ? m_run + logf(l_run) : -INFINITY;
  }
}

template <bool DROP>
__global__ __launch_bounds__(CA_MAXT) void attn_ca_bwd_kernel(const pq3d_attn_desc d) {
  ATTN_KARG_PIN(d);
  ATTN_KARG_PIN_BWD(d);
  extern __shared__ __attribute__((aligned(16))) unsigned char ca_sm[];
  const int Lq = d.Lq, Lk = d.Lk, LPq = (Lq + 31) & ~31, LPk = (Lk + 31) & ~31, TQ = (Lq + 15) & ~15, TK = (Lk + 15) & ~15;
  const CaLds S = carve(ca_sm, TQ, TK, LPq, LPk, true);
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const int b = blockIdx.y, h = blockIdx.x;
  const bf16_t* q = (const bf16_t*)d.q + (long)b * d.q_sb + (long)h * d.q_sh;
  const bf16_t* k = (const bf16_t*)d.k + (long)b * d.k_sb + (long)h * d.k_sh;
  const bf16_t* v = (const bf16_t*)d.v + (long)b * d.v_sb + (long)h * d.v_sh;
  const bf16_t* o = (const bf16_t*)d.o + (long)b * d.o_sb + (long)h * d.o_sh;
  const bf16_t* g = (const bf16_t*)d.dout + (long)b * d.o_sb + (long)h * d.o_sh;
  const long sbase = ((long)b * d.H + h) * Lq;
  // lse and delta = rowsum(dO * O) of every query: DH / 8 lanes per row, requested before the operand staging waits
  {
    constexpr int CPR = DH / 8;
    for (int c = tid; c < LPq * CPR; c += nthr) {
      const int row = c / CPR, x = (c % CPR) * 8;
      const bool ok = row < Lq;
      const int gr = min(row, Lq - 1);
      const u32x4 gv = *(const u32x4*)(g + (long)gr * d.o_sl + x), ov = *(const u32x4*)(o + (long)gr * d.o_sl + x);
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s += __uint_as_float(gv[j] << 16) * __uint_as_float(ov[j] << 16);
        s += __uint_as_float(gv[j] & 0xffff0000u) * __uint_as_float(ov[j] & 0xffff0000u);
      }
#pragma unroll
      for (int o_ = 1; o_ < CPR; o_ <<= 1) s += __shfl_xor(s, o_, 64);   // the CPR chunks of a row sit in neighbouring lanes
      if (x == 0) {
        float l = INFINITY;                    // padded queries: lse = +inf -> P = 0
        if (ok) {
          l = d.lse[sbase + row];
          d.delta[sbase + row] = s;
          if (l == -INFINITY) l = INFINITY;    // fully masked row: all probabilities 0
        }
        S.dl[row] = ok ? s : 0.f;
        S.lse[row] = l;
      }
    }
  }
  stage_rows<2>(S.Q, q, d.q_sl, Lq, TQ, tid, nthr);
  stage_rows<2>(S.G, g, d.o_sl, Lq, TQ, tid, nthr);
  stage_rows<2>(S.K, k, d.k_sl, Lk, TK, tid, nthr);
  stage_rows<2>(S.V, v, d.v_sl, Lk, TK, tid, nthr);
  for (int j = tid; j < LPk; j += nthr) S.kb[j] = (j < Lk && !(d.kpm && d.kpm[(long)b * Lk + j])) ? 0.f : -INFINITY;
  __syncthreads();
  DropState dst;
  uint32_t drow0 = 0;     // site row of query 0 of this (scene, head)
  if constexpr (DROP) {
    dst = drop_init(d.drop, d.drop_bmod > 0 ? b / d.drop_bmod : 0, Lk);
    drow0 = (uint32_t)(((long)(d.drop_bmod > 0 ? b % d.drop_bmod : b) * d.H + h) * Lq);
  }
  // ---------------- phase A: wave = query block; transposed tiles (lane = query column, 4 keys per tile per lane)
  const int q0 = wave * 16;
  if (q0 < TQ) {
    const int qrow = q0 + li;
    u32x4 qf[KS], gf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { qf[ks] = frag_rm(S.Q, qrow, lg, ks); gf[ks] = frag_rm(S.G, qrow, lg, ks); }
    const float lse = S.lse[qrow], ndl = -S.dl[qrow];
    f32x4 at[OT];   // dQ^T
#pragma unroll
    for (int t = 0; t < OT; ++t) at[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int t0 = 0; t0 < LPk; t0 += 32) {
      const bool has2 = t0 + 16 < TK;                 // uniform: the pair's second key tile exists (else P = 0 through kb)
      f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
      const float pin = DROP ? 0.f : ndl;             // dP - delta: without dropout the accumulators start at -delta
      f32x4 p0 = {pin, pin, pin, pin}, p1 = p0;
      mmak(s0, S.K, t0 + li, lg, qf);
      mmak(p0, S.V, t0 + li, lg, gf);                 // dP^T = V dO^T
      if (has2) {
        mmak(s1, S.K, t0 + 16 + li, lg, qf);
        mmak(p1, S.V, t0 + 16 + li, lg, gf);
      }
      const float4 kb0 = *(const float4*)&S.kb[t0 + 4 * lg], kb1 = *(const float4*)&S.kb[t0 + 16 + 4 * lg];
      const float kbv[8] = {kb0.x, kb0.y, kb0.z, kb0.w, kb1.x, kb1.y, kb1.z, kb1.w};
      float ds[8], kf[8];
      if constexpr (DROP) drop_keep8(dst, drow0 + (uint32_t)min(qrow, Lq - 1), t0 + 4 * lg, kf);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float sv = (j < 4 ? s0[j] : s1[j - 4]) * d.scale + kbv[j];
        const float p = __expf(sv - lse);                        // masked / padded: exp(-inf) = 0
        const float dp = j < 4 ? p0[j] : p1[j - 4];
        if constexpr (DROP) ds[j] = p * (dp * (kf[j] * dst.scale) + ndl);   // dS = P (dP keep / (1 - p) - delta)
        else ds[j] = p * dp;
      }
      const u32x4 df = pack_frag<bf16_t>(ds);
#pragma unroll
      for (int t = 0; t < OT; ++t) Mma<bf16_t>::mma(at[t], frag_tr(S.K, t0, t0 + 16, 16 * t, li, lg, has2), df);   // dQ^T += K^T dS^T
    }
    if (qrow < Lq) {
      bf16_t* dq = (bf16_t*)d.dq + (long)b * d.q_sb + (long)h * d.q_sh + (long)qrow * d.q_sl;
#pragma unroll
      for (int t = 0; t < OT; ++t)
        *(u32x2*)(dq + 16 * t + 4 * lg) =
            (u32x2){pack_bf2(at[t][0] * d.scale, at[t][1] * d.scale), pack_bf2(at[t][2] * d.scale, at[t][3] * d.scale)};
    }
  }
  // ---------------- phase B: wave = key block; plain tiles (lane = key column, 4 queries per tile per lane)
  const int k0 = wave * 16;
  if (k0 >= TK) return;
  const int krow = k0 + li;
  u32x4 kf[KS], vf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) { kf[ks] = frag_rm(S.K, krow, lg, ks); vf[ks] = frag_rm(S.V, krow, lg, ks); }
  const float kbias = S.kb[krow];
  f32x4 dvt[OT], dkt[OT];
#pragma unroll
  for (int t = 0; t < OT; ++t) { dvt[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; dkt[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  for (int t0 = 0; t0 < LPq; t0 += 32) {
    const float4 l0 = *(const float4*)&S.lse[t0 + 4 * lg], l1 = *(const float4*)&S.lse[t0 + 16 + 4 * lg];
    const float4 e0 = *(const float4*)&S.dl[t0 + 4 * lg], e1 = *(const float4*)&S.dl[t0 + 16 + 4 * lg];
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    f32x4 p0 = {-e0.x, -e0.y, -e0.z, -e0.w}, p1 = {-e1.x, -e1.y, -e1.z, -e1.w};   // dP - delta
    if constexpr (DROP) { p0 = (f32x4){0.f, 0.f, 0.f, 0.f}; p1 = p0; }
    const bool has2 = t0 + 16 < TQ;             // uniform: the pair's second query tile exists (else lse = +inf: P = 0)
    mmak(s0, S.Q, t0 + li, lg, kf);             // S = Q K^T
    mmak(p0, S.G, t0 + li, lg, vf);             // dP = dO V^T
    if (has2) {
      mmak(s1, S.Q, t0 + 16 + li, lg, kf);
      mmak(p1, S.G, t0 + 16 + li, lg, vf);
    }
    const float ls[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
    const float dl[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
    float p[8], ds[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float sv = (j < 4 ? s0[j] : s1[j - 4]) * d.scale + kbias;
      const float dp = j < 4 ? p0[j] : p1[j - 4];
      p[j] = __expf(sv - ls[j]);
      if constexpr (DROP) {
        const int qr = min(t0 + (j < 4 ? 0 : 16) + 4 * lg + (j & 3), Lq - 1);
        const float kc = drop_keep(dst, drow0 + (uint32_t)qr, (uint32_t)min(krow, Lk - 1)) ? dst.scale : 0.f;
        ds[j] = p[j] * (dp * kc - dl[j]);
        p[j] *= kc;                               // dV sees the dropped, rescaled probabilities
      } else {
        ds[j] = p[j] * dp;
      }
    }
    const u32x4 pf = pack_frag<bf16_t>(p), df = pack_frag<bf16_t>(ds);
#pragma unroll
    for (int t = 0; t < OT; ++t) {
      Mma<bf16_t>::mma(dvt[t], frag_tr(S.G, t0, t0 + 16, 16 * t, li, lg, has2), pf);   // dV^T += dO^T P
      Mma<bf16_t>::mma(dkt[t], frag_tr(S.Q, t0, t0 + 16, 16 * t, li, lg, has2), df);   // dK^T += Q^T dS
    }
  }
  if (krow < Lk) {
    bf16_t* dk = (bf16_t*)d.dk + (long)b * d.k_sb + (long)h * d.k_sh + (long)krow * d.k_sl;
    bf16_t* dv = (bf16_t*)d.dv + (long)b * d.v_sb + (long)h * d.v_sh + (long)krow * d.v_sl;
#pragma unroll
    for (int t = 0; t < OT; ++t) {
      *(u32x2*)(dk + 16 * t + 4 * lg) =
          (u32x2){pack_bf2(dkt[t][0] * d.scale, dkt[t][1] * d.scale), pack_bf2(dkt[t][2] * d.scale, dkt[t][3] * d.scale)};
      *(u32x2*)(dv + 16 * t + 4 * lg) = (u32x2){pack_bf2(dvt[t][0], dvt[t][1]), pack_bf2(dvt[t][2], dvt[t][3])};
    }
  }
}
