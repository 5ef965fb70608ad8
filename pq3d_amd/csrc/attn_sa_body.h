// Body of attn_sa.hip, compiled once per head size: SA_DH (32 / 64) and SA_MAXT (threads per workgroup) are set by the
// including file, which wraps each inclusion in its own namespace.  See attn_sa.hip for the design.
constexpr int DH = SA_DH, LDH = DH + 8;   // LDS row: 40 (72) bf16 = 20 (36) dwords -> 16 rows hit 16 distinct 4-bank groups
constexpr int KS = DH / 32, OT = DH / 16;  // 32-wide k steps of a product over d_h; 16-row tiles of a [d_h][tokens] result

PQ_DEV float xrow_sum(float v) {    // sum over the 4 lanes li, li + 16, li + 32, li + 48
  u32pair_s a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  u32pair_s b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
PQ_DEV float xrow_max(float v) {
  u32pair_s a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  u32pair_s b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

struct HL { u32x4 hi, lo; };
PQ_DEV HL split8(const float* v) {
  HL r;
  r.hi = pack_frag<bf16_t>(v);
  float w[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    w[2 * j] = v[2 * j] - __uint_as_float(r.hi[j] << 16);
    w[2 * j + 1] = v[2 * j + 1] - __uint_as_float(r.hi[j] & 0xffff0000u);
  }
  r.lo = pack_frag<bf16_t>(w);
  return r;
}
// acc += a * b at fp32 grade: lo*hi + hi*lo + hi*hi (the order of mma_tile_x3 in gemm.hip)
PQ_DEV void mma3(f32x4& acc, const HL& a, const HL& b) {
  Mma<bf16_t>::mma(acc, a.lo, b.hi);
  Mma<bf16_t>::mma(acc, a.hi, b.lo);
  Mma<bf16_t>::mma(acc, a.hi, b.hi);
}
// row-major fragment: token row `row`, d_h slots 8 lg .. 8 lg + 7 (A or B operand of a product over d_h)
PQ_DEV HL frag_rm(const bf16_t* hi, const bf16_t* lo, int row, int lg, int ks = 0) {
  HL r;
  r.hi = *(const u32x4*)&hi[row * LDH + ks * 32 + lg * 8];
  r.lo = *(const u32x4*)&lo[row * LDH + ks * 32 + lg * 8];
  return r;
}
// a product over d_h: KS k steps, each lo*hi + hi*lo + hi*hi in sequence (d_h = 32: exactly mma3)
PQ_DEV void mma3k(f32x4& acc, const bf16_t* hi, const bf16_t* lo, int row, int lg, const HL* b) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) mma3(acc, frag_rm(hi, lo, row, lg, ks), b[ks]);
}
// transposed fragment (A operand, m = d_h index c0 + li, k = tokens): slots 0..3 = tokens t0 + 4 lg + 0..3, slots 4..7 =
// tokens t1 + 4 lg + 0..3 -- the token order in which two 16-token C tiles sit in a lane's registers
PQ_DEV u32x4 frag_tr1(const bf16_t* plane, int t0, int t1, int c0, int li, int lg) {
  const bf16_t* p0 = plane + (t0 + 4 * lg + (li >> 2)) * LDH + c0 + 4 * (li & 3);
  const bf16_t* p1 = plane + (t1 + 4 * lg + (li >> 2)) * LDH + c0 + 4 * (li & 3);
  const v4i16_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_t*)p0);
  const v4i16_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_t*)p1);
  const u32x2 x = __builtin_bit_cast(u32x2, a), y = __builtin_bit_cast(u32x2, b);
  return (u32x4){x.x, x.y, y.x, y.y};
}
PQ_DEV HL frag_tr(const bf16_t* hi, const bf16_t* lo, int t0, int t1, int c0, int li, int lg) {
  HL r;
  r.hi = frag_tr1(hi, t0, t1, c0, li, lg);
  r.lo = frag_tr1(lo, t0, t1, c0, li, lg);
  return r;
}

// fp32 [L, DH] (token stride sl) -> hi / lo planes [LP][LDH]; rows >= L are zero.  All loads of the thread in flight first.
template <int MAXC> PQ_DEV void stage_planes(bf16_t* hi, bf16_t* lo, const float* src, long sl, int L, int LP, int tid, int nthr) {
  const int nch = LP * (DH / 8);
  for (int base = tid; base < nch; base += nthr * MAXC) {
    float4 a[MAXC], b[MAXC];
#pragma unroll
    for (int u = 0; u < MAXC; ++u) {
      const int c = base + u * nthr, row = c / (DH / 8), x = (c % (DH / 8)) * 8;
      const float* p = src + (long)min(row, L - 1) * sl + x;
      a[u] = *(const float4*)p;
      b[u] = *(const float4*)(p + 4);
    }
#pragma unroll
    for (int u = 0; u < MAXC; ++u) {
      const int c = base + u * nthr, row = c / (DH / 8), x = (c % (DH / 8)) * 8;
      if (c < nch) {
        float v[8] = {a[u].x, a[u].y, a[u].z, a[u].w, b[u].x, b[u].y, b[u].z, b[u].w};
        if (row >= L) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = 0.f;
        }
        const HL s = split8(v);
        *(u32x4*)&hi[row * LDH + x] = s.hi;
        *(u32x4*)&lo[row * LDH + x] = s.lo;
      }
    }
  }
}

struct SaLds {
  bf16_t *Qh, *Ql, *Kh, *Kl, *Vh, *Vl, *Gh, *Gl;
  float *kb, *lse, *dl;   // additive key term (0 / -inf) [LPk]; lse, delta [LPq]
};
PQ_DEV SaLds carve(unsigned char* sm, int LPq, int LPk, bool bwd) {
  SaLds s;
  bf16_t* p = (bf16_t*)sm;
  s.Qh = p; p += LPq * LDH; s.Ql = p; p += LPq * LDH;
  s.Kh = p; p += LPk * LDH; s.Kl = p; p += LPk * LDH;
  s.Vh = p; p += LPk * LDH; s.Vl = p; p += LPk * LDH;
  s.Gh = s.Gl = nullptr;
  if (bwd) { s.Gh = p; p += LPq * LDH; s.Gl = p; p += LPq * LDH; }
  s.kb = (float*)p;
  s.lse = s.kb + LPk;
  s.dl = s.lse + LPq;
  return s;
}
size_t sa_lds_bytes(int Lq, int Lk, bool bwd, int dm_fold = 0) {
  const size_t LPq = (Lq + 15) & ~15, LPk = (Lk + 31) & ~31;
  return ((bwd ? 4 : 2) * LPq + 4 * LPk + dm_fold) * LDH * sizeof(bf16_t) + (LPk + 2 * LPq) * sizeof(float) + 16;
}

// fp32 weight slice [rows][DH] (row stride sl) -> single bf16 plane [rows][LDH] (k-major tile of a folded projection)
PQ_DEV void stage_wslice(bf16_t* dst, const float* src, long sl, int rows, int tid, int nthr) {
  const int nch = rows * (DH / 8);
  for (int base = tid; base < nch; base += nthr * 2) {
    float4 a[2], b[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int c = min(base + u * nthr, nch - 1);
      const float* p = src + (long)(c / (DH / 8)) * sl + (c % (DH / 8)) * 8;
      a[u] = *(const float4*)p;
      b[u] = *(const float4*)(p + 4);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int c = base + u * nthr;
      if (c < nch) {
        const float v[8] = {a[u].x, a[u].y, a[u].z, a[u].w, b[u].x, b[u].y, b[u].z, b[u].w};
        *(u32x4*)&dst[(c / (DH / 8)) * LDH + (c % (DH / 8)) * 8] = pack_frag<bf16_t>(v);
      }
    }
  }
}

// 4 consecutive bias values bias[row][c .. c + 3] (c % 4 == 0), 0 beyond the row / matrix
PQ_DEV void bias4(const float* bias, int row, int c, int Lq, int Lk, bool vec, float* out) {
  if (!bias || row >= Lq) { out[0] = out[1] = out[2] = out[3] = 0.f; return; }
  const float* p = bias + (long)row * Lk + c;
  if (vec && c + 3 < Lk) { const float4 t = *(const float4*)p; out[0] = t.x; out[1] = t.y; out[2] = t.z; out[3] = t.w; }
  else {
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = c + j < Lk ? p[j] : 0.f;
  }
}

// (Measured and removed, round 3: the head's additive bias copied into LDS with the operand loads, or requested 4 steps ahead.
// The bias is HBM-cold when the layer's attention runs -- written at the start of the step, 100+ launches earlier -- so
// whatever requests it earlier pays the same ~2 us in front of the operand staging instead: key loops 4.6 -> 3.6 us, staging
// 2.4 -> 4.0 us, 17.6 -> 19-21 us per backward launch.  tools/probes/sa_timeline.py)
// One wave, one 16-query block of the forward: online softmax over all keys of the staged planes.  lrow: the lane's query row in
// the Q planes; brow: its row of the head's additive bias (>= Lq: no bias -- a row that is not part of this scene).  Returns the
// unnormalised O^T tiles (tile t = rows d_h 16 t + 4 lg + r, column = query) with the running maximum and sum.  Shared by
// attn_sa_fwd_kernel and the self-attention step of chain_ffn.hip: the same instructions, hence the same bits.
struct SaFwdOut { f32x4 ot[OT]; float m, l; };
PQ_DEV SaFwdOut sa_fwd_block(const SaLds& S, int lrow, int brow, const float* bias, int Lq, int Lk, int LPk, float scale, int lg) {
  const bool bvec = (Lk & 3) == 0 && ((((uintptr_t)bias) & 15) == 0);
  HL qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qf[ks] = frag_rm(S.Qh, S.Ql, lrow, lg, ks);
  const int li = lrow & 15;
  float m_run = -INFINITY, l_run = 0.f;
  SaFwdOut r;
#pragma unroll
  for (int t = 0; t < OT; ++t) r.ot[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bn[8];
  bias4(bias, brow, 4 * lg, Lq, Lk, bvec, bn);
  bias4(bias, brow, 16 + 4 * lg, Lq, Lk, bvec, bn + 4);
  for (int t0 = 0; t0 < LPk; t0 += 32) {
    float bc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bc[j] = bn[j];
    if (t0 + 32 < LPk) {   // next pair's bias in flight during this pair's arithmetic
      bias4(bias, brow, t0 + 32 + 4 * lg, Lq, Lk, bvec, bn);
      bias4(bias, brow, t0 + 48 + 4 * lg, Lq, Lk, bvec, bn + 4);
    }
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    mma3k(s0, S.Kh, S.Kl, t0 + li, lg, qf);
    mma3k(s1, S.Kh, S.Kl, t0 + 16 + li, lg, qf);
    float sv[8];
    const float4 kb0 = *(const float4*)&S.kb[t0 + 4 * lg], kb1 = *(const float4*)&S.kb[t0 + 16 + 4 * lg];
    const float kbv[8] = {kb0.x, kb0.y, kb0.z, kb0.w, kb1.x, kb1.y, kb1.z, kb1.w};
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sv[j] = (j < 4 ? s0[j] : s1[j - 4]) * scale + bc[j] + kbv[j];
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
    const HL pf = split8(p);
#pragma unroll
    for (int t = 0; t < OT; ++t) {
      r.ot[t] *= alpha;
      mma3(r.ot[t], frag_tr(S.Vh, S.Vl, t0, t0 + 16, 16 * t, li, lg), pf);
    }
  }
  r.m = m_run; r.l = l_run;
  return r;
}

#ifndef SA_DEVICE_ONLY
__global__ __launch_bounds__(SA_MAXT) void attn_sa_fwd_kernel(const pq3d_attn_desc d) {
  ATTN_KARG_PIN(d);
  extern __shared__ __attribute__((aligned(16))) unsigned char sa_sm[];
  const int Lq = d.Lq, Lk = d.Lk, LPq = (Lq + 15) & ~15, LPk = (Lk + 31) & ~31;
  const SaLds S = carve(sa_sm, LPq, LPk, false);
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const int b = blockIdx.y, h = blockIdx.x;
  const float* q = (const float*)d.q + (long)b * d.q_sb + (long)h * d.q_sh;
  const float* k = (const float*)d.k + (long)b * d.k_sb + (long)h * d.k_sh;
  const float* v = (const float*)d.v + (long)b * d.v_sb + (long)h * d.v_sh;
  const float* gbias = d.bias ? d.bias + ((long)b * d.H + h) * Lq * (long)Lk : nullptr;
  stage_planes<2>(S.Qh, S.Ql, q, d.q_sl, Lq, LPq, tid, nthr);
  stage_planes<2>(S.Kh, S.Kl, k, d.k_sl, Lk, LPk, tid, nthr);
  stage_planes<2>(S.Vh, S.Vl, v, d.v_sl, Lk, LPk, tid, nthr);
  for (int j = tid; j < LPk; j += nthr) S.kb[j] = (j < Lk && !(d.kpm && d.kpm[(long)b * Lk + j])) ? 0.f : -INFINITY;
  __syncthreads();
  const int q0 = wave * 16;
  if (q0 >= LPq) return;
  const int qrow = q0 + li;                       // this lane's query (column of the transposed score tiles)
  const SaFwdOut r = sa_fwd_block(S, qrow, qrow, gbias, Lq, Lk, LPk, d.scale, lg);
  if (qrow < Lq) {
    const float inv = r.l > 0.f ? 1.f / r.l : 0.f;
    float* o = (float*)d.o + (long)b * d.o_sb + (long)h * d.o_sh + (long)qrow * d.o_sl;
#pragma unroll
    for (int t = 0; t < OT; ++t)
      *(float4*)(o + 16 * t + 4 * lg) = make_float4(r.ot[t][0] * inv, r.ot[t][1] * inv, r.ot[t][2] * inv, r.ot[t][3] * inv);
    if (lg == 0) d.lse[((long)b * d.H + h) * Lq + qrow] = r.l > 0.f ? r.m + logf(r.l) : -INFINITY;
  }
}

// in-kernel timeline (probe builds only, tools/probes/sa_timeline.py): wave 0 of workgroup (0, 0) stamps the 100 MHz clock
#undef SA_TL
#undef SA_TLV
#ifdef PQ3D_SA_TIMELINE
#define SA_TL(i) do { if (d.ws && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) ((long long*)d.ws)[i] = wall_clock64(); } while (0)
#define SA_TLV(i, val) do { float x_ = (val); asm volatile("v_mov_b32 %0, %0" : "+v"(x_)); SA_TL(i); } while (0)   // after `val` exists
#else
#define SA_TL(i) do { } while (0)
#define SA_TLV(i, val) do { } while (0)
#endif

__global__ __launch_bounds__(SA_MAXT) void attn_sa_bwd_kernel(const pq3d_attn_desc d) {
  ATTN_KARG_PIN(d);
  ATTN_KARG_PIN_BWD(d);
  extern __shared__ __attribute__((aligned(16))) unsigned char sa_sm[];
  const int Lq = d.Lq, Lk = d.Lk, LPq = (Lq + 15) & ~15, LPk = (Lk + 31) & ~31, LPq2 = (Lq + 31) & ~31;
  const SaLds S = carve(sa_sm, LPq2, LPk, true);   // query planes padded to whole PAIRS of 16-row tiles (phase B walks pairs)
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const int b = blockIdx.y, h = blockIdx.x;
  // grid.z = 2 (few (scene, head) units: most CUs would idle): workgroup z = 0 runs phase A (dQ, dbias, delta), z = 1 phase B
  // (dK, dV); both stage the same planes.  part 0: one workgroup runs both phases.
  // grid.z = 4: each phase on two workgroups, the 16-row blocks dealt out alternately (sub).
  const int part = gridDim.z > 1 ? (int)(blockIdx.z & 1) + 1 : 0, nsub = gridDim.z > 2 ? 2 : 1, sub = (int)blockIdx.z >> 1;
  const float* q = (const float*)d.q + (long)b * d.q_sb + (long)h * d.q_sh;
  const float* k = (const float*)d.k + (long)b * d.k_sb + (long)h * d.k_sh;
  const float* v = (const float*)d.v + (long)b * d.v_sb + (long)h * d.v_sh;
  const float* o = (const float*)d.o + (long)b * d.o_sb + (long)h * d.o_sh;
  const float* g = (const float*)d.dout + (long)b * d.o_sb + (long)h * d.o_sh;
  const long sbase = ((long)b * d.H + h) * Lq;
  SA_TL(0);
#if SA_DH == 32
  // folded out-projection backward (pq3d_attn_proj, DOUT): dO of this head is formed here from the gradient of the
  // projection's output and the head's 32 columns of the weight (staged k-major as one bf16 plane behind the float arrays)
  const bool fold = d.proj.mode == PQ3D_ATTN_PROJ_DOUT;
  bf16_t* const Wt = (bf16_t*)(S.dl + LPq2);
  const float* gbias = d.bias ? d.bias + ((long)b * d.H + h) * Lq * (long)Lk : nullptr;
  // folded projection: the gradient rows of this wave's first query block (dm = 256: all 8 k steps), its O row and lse are
  // requested before anything waits -- they are back by the time the weight slice is staged
  float4 xea[8], xeb[8], oe0 = {0.f, 0.f, 0.f, 0.f}, oe1 = {0.f, 0.f, 0.f, 0.f};
  float lse_e = 0.f;
  const bool early = fold && d.proj.dm == 256 && wave * 16 < Lq;   // uniform per wave
  if (early) {
    const int qr = min(wave * 16 + li, Lq - 1);
    const float* px = d.proj.x + ((long)b * Lq + qr) * 256 + lg * 8;
#pragma unroll
    for (int u = 0; u < 8; ++u) { xea[u] = *(const float4*)(px + u * 32); xeb[u] = *(const float4*)(px + u * 32 + 4); }
    const float* po = o + (long)qr * d.o_sl;
    oe0 = *(const float4*)(po + 4 * lg);
    oe1 = *(const float4*)(po + 16 + 4 * lg);
    lse_e = d.lse[sbase + qr];
  }
  if (fold) stage_wslice(Wt, d.proj.w[0] + h * DH, d.proj.dm, d.proj.dm, tid, nthr);
#else
  const bool fold = false;   // the folded out-projection exists at d_h = 32 only (host-checked)
  const float* gbias = d.bias ? d.bias + ((long)b * d.H + h) * Lq * (long)Lk : nullptr;
#endif
  stage_planes<2>(S.Qh, S.Ql, q, d.q_sl, Lq, LPq2, tid, nthr);
  if (!fold) stage_planes<2>(S.Gh, S.Gl, g, d.o_sl, Lq, LPq2, tid, nthr);
  stage_planes<2>(S.Kh, S.Kl, k, d.k_sl, Lk, LPk, tid, nthr);
  stage_planes<2>(S.Vh, S.Vl, v, d.v_sl, Lk, LPk, tid, nthr);
  for (int j = tid; j < LPk; j += nthr) S.kb[j] = (j < Lk && !(d.kpm && d.kpm[(long)b * Lk + j])) ? 0.f : -INFINITY;
  SA_TL(1);
#if SA_DH == 32
  if (fold) {
    __syncthreads();   // weight slice staged
    SA_TL(2);
    const int dm = d.proj.dm, nks = dm >> 5;
    const float* gx = d.proj.x + (long)b * Lq * dm;
    for (int rb = wave; rb * 16 < LPq2; rb += nthr >> 6) {
      // dO^T tile = W_slice^T (A: d_h index x out index, transposing read of the k-major plane) . x^T (B: lane = query,
      // 8 consecutive out indices straight from the row-major gradient rows): 4 consecutive d_h values per lane and tile
      const int qrow = rb * 16 + li;
      const bool ok = qrow < Lq;
      const float* px = gx + (long)min(qrow, Lq - 1) * dm + lg * 8;
      f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = {0.f, 0.f, 0.f, 0.f};
      const bool pre = early && rb == wave;   // uniform: the rows are already in registers
      if (pre) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float xv[8] = {xea[u].x, xea[u].y, xea[u].z, xea[u].w, xeb[u].x, xeb[u].y, xeb[u].z, xeb[u].w};
          const u32x4 bf = pack_frag<bf16_t>(xv);
          Mma<bf16_t>::mma(g0, km_frag(Wt, LDH, 0, u, li, lg), bf);
          Mma<bf16_t>::mma(g1, km_frag(Wt, LDH, 16, u, li, lg), bf);
        }
      }
      for (int ks0 = 0; ks0 < ((!pre && rb * 16 < Lq) ? nks : 0); ks0 += 4) {   // padding blocks: zeros, no loads
        float4 xa[4], xb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int ks = min(ks0 + u, nks - 1);
          xa[u] = *(const float4*)(px + ks * 32);
          xb[u] = *(const float4*)(px + ks * 32 + 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (ks0 + u < nks) {   // uniform
            const float xv[8] = {xa[u].x, xa[u].y, xa[u].z, xa[u].w, xb[u].x, xb[u].y, xb[u].z, xb[u].w};
            const u32x4 bf = pack_frag<bf16_t>(xv);
            Mma<bf16_t>::mma(g0, km_frag(Wt, LDH, 0, ks0 + u, li, lg), bf);
            Mma<bf16_t>::mma(g1, km_frag(Wt, LDH, 16, ks0 + u, li, lg), bf);
          }
        }
      }
      float gv[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { gv[j] = ok ? g0[j] : 0.f; gv[4 + j] = ok ? g1[j] : 0.f; }
      const HL gs = split8(gv);
      *(u32x2*)&S.Gh[qrow * LDH + 4 * lg] = (u32x2){gs.hi[0], gs.hi[1]};
      *(u32x2*)&S.Gh[qrow * LDH + 16 + 4 * lg] = (u32x2){gs.hi[2], gs.hi[3]};
      *(u32x2*)&S.Gl[qrow * LDH + 4 * lg] = (u32x2){gs.lo[0], gs.lo[1]};
      *(u32x2*)&S.Gl[qrow * LDH + 16 + 4 * lg] = (u32x2){gs.lo[2], gs.lo[3]};
      float s = 0.f, l = INFINITY;
      if (ok) {
        const float* po = o + (long)qrow * d.o_sl;
        const float4 t0 = pre ? oe0 : *(const float4*)(po + 4 * lg), t1 = pre ? oe1 : *(const float4*)(po + 16 + 4 * lg);
        s = ((gv[0] * t0.x + gv[1] * t0.y) + (gv[2] * t0.z + gv[3] * t0.w)) + ((gv[4] * t1.x + gv[5] * t1.y) + (gv[6] * t1.z + gv[7] * t1.w));
      }
      s = xrow_sum(s);
      if (lg == 0) {
        if (ok) {
          l = pre ? lse_e : d.lse[sbase + qrow];
          if (part != 2 && sub == 0) d.delta[sbase + qrow] = s;
          if (l == -INFINITY) l = INFINITY;
        }
        S.dl[qrow] = s;
        S.lse[qrow] = l;
      }
    }
  }
#endif
  for (int i = tid; i < (fold ? 0 : LPq2); i += nthr) {   // lse and delta = rowsum(dO * O) of every query
    float s = 0.f, l = INFINITY;             // padded queries: lse = +inf -> P = 0
    if (i < Lq) {
#pragma unroll
      for (int x = 0; x < DH; x += 4) {
        const float4 a = *(const float4*)(g + (long)i * d.o_sl + x), t = *(const float4*)(o + (long)i * d.o_sl + x);
        s += (a.x * t.x + a.y * t.y) + (a.z * t.z + a.w * t.w);
      }
      l = d.lse[sbase + i];
      if (part != 2 && sub == 0) d.delta[sbase + i] = s;
      if (l == -INFINITY) l = INFINITY;      // fully masked row: all probabilities 0
    }
    S.dl[i] = s;
    S.lse[i] = l;
  }
  SA_TL(3);
  __syncthreads();
  SA_TL(4);
  const float* bias = gbias;
  float* dbias = d.dbias ? d.dbias + ((long)b * d.H + h) * Lq * (long)Lk : nullptr;
  const bool bvec = (Lk & 3) == 0 && ((((uintptr_t)bias)) & 15) == 0;
  const bool dvec = (Lk & 3) == 0 && ((((uintptr_t)dbias)) & 15) == 0;
  // ---------------- phase A: wave = query block; transposed tiles (lane = query column, 4 keys per tile per lane)
  const int q0 = wave * 16;
  if (part != 2 && q0 < LPq && (wave & (nsub - 1)) == sub) {
    const int qrow = q0 + li;
    HL qf[KS], gf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { qf[ks] = frag_rm(S.Qh, S.Ql, qrow, lg, ks); gf[ks] = frag_rm(S.Gh, S.Gl, qrow, lg, ks); }
    const float lse = S.lse[qrow], dlt = S.dl[qrow];
    f32x4 at[OT];   // dQ^T
#pragma unroll
    for (int t = 0; t < OT; ++t) at[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bn[8];
    bias4(bias, qrow, 4 * lg, Lq, Lk, bvec, bn);
    bias4(bias, qrow, 16 + 4 * lg, Lq, Lk, bvec, bn + 4);
    for (int t0 = 0; t0 < LPk; t0 += 32) {
      float bc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) bc[j] = bn[j];
      if (t0 + 32 < LPk) {
        bias4(bias, qrow, t0 + 32 + 4 * lg, Lq, Lk, bvec, bn);
        bias4(bias, qrow, t0 + 48 + 4 * lg, Lq, Lk, bvec, bn + 4);
      }
      f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f}, p0 = {0.f, 0.f, 0.f, 0.f}, p1 = {0.f, 0.f, 0.f, 0.f};
      if (t0 == 0) SA_TLV(8, bc[0] + bc[7]);
      mma3k(s0, S.Kh, S.Kl, t0 + li, lg, qf);
      mma3k(s1, S.Kh, S.Kl, t0 + 16 + li, lg, qf);
      mma3k(p0, S.Vh, S.Vl, t0 + li, lg, gf);           // dP^T = V dO^T
      mma3k(p1, S.Vh, S.Vl, t0 + 16 + li, lg, gf);
      if (t0 == 0) SA_TLV(9, s0[0] + s1[0] + p0[0] + p1[0]);
      const float4 kb0 = *(const float4*)&S.kb[t0 + 4 * lg], kb1 = *(const float4*)&S.kb[t0 + 16 + 4 * lg];
      const float kbv[8] = {kb0.x, kb0.y, kb0.z, kb0.w, kb1.x, kb1.y, kb1.z, kb1.w};
      float ds[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float sv = (j < 4 ? s0[j] : s1[j - 4]) * d.scale + bc[j] + kbv[j];
        const float p = __expf(sv - lse);                        // masked / padded: exp(-inf) = 0
        ds[j] = p * ((j < 4 ? p0[j] : p1[j - 4]) - dlt);
      }
      if (dbias && qrow < Lq) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int c = t0 + 16 * hh + 4 * lg;
          float* p = dbias + (long)qrow * Lk + c;
          if (dvec && c + 3 < Lk) *(float4*)p = make_float4(ds[4 * hh], ds[4 * hh + 1], ds[4 * hh + 2], ds[4 * hh + 3]);
          else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (c + j < Lk) p[j] = ds[4 * hh + j];
          }
        }
      }
      if (t0 == 0) SA_TLV(10, ds[0] + ds[7]);
      const HL df = split8(ds);
#pragma unroll
      for (int t = 0; t < OT; ++t) mma3(at[t], frag_tr(S.Kh, S.Kl, t0, t0 + 16, 16 * t, li, lg), df);   // dQ^T += K^T dS^T
      if (t0 == 0) SA_TLV(11, at[0][0] + at[1][0]);
      if (t0 == 32) SA_TLV(12, at[0][0] + at[1][0]);
    }
    if (qrow < Lq) {
      float* dq = (float*)d.dq + (long)b * d.q_sb + (long)h * d.q_sh + (long)qrow * d.q_sl;
#pragma unroll
      for (int t = 0; t < OT; ++t)
        *(float4*)(dq + 16 * t + 4 * lg) = make_float4(at[t][0] * d.scale, at[t][1] * d.scale, at[t][2] * d.scale, at[t][3] * d.scale);
    }
  }
  SA_TL(5);
  // ---------------- phase B: wave = key block; plain tiles (lane = key column, 4 queries per tile per lane)
  const int k0 = wave * 16;
  if (part == 1 || k0 >= ((Lk + 15) & ~15) || (wave & (nsub - 1)) != sub) return;
  const int krow = k0 + li;
  HL kf[KS], vf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) { kf[ks] = frag_rm(S.Kh, S.Kl, krow, lg, ks); vf[ks] = frag_rm(S.Vh, S.Vl, krow, lg, ks); }
  const float kbias = S.kb[krow];
  f32x4 dvt[OT], dkt[OT];
#pragma unroll
  for (int t = 0; t < OT; ++t) { dvt[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; dkt[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  for (int t0 = 0; t0 < LPq2; t0 += 32) {
    // bias[q][key] of this lane's 8 query rows, all requested up front
    float bc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int qr = t0 + (j < 4 ? 0 : 16) + 4 * lg + (j & 3);
      bc[j] = (bias && qr < Lq && krow < Lk) ? bias[(long)qr * Lk + krow] : 0.f;
    }
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f}, p0 = {0.f, 0.f, 0.f, 0.f}, p1 = {0.f, 0.f, 0.f, 0.f};
    mma3k(s0, S.Qh, S.Ql, t0 + li, lg, kf);             // S = Q K^T
    mma3k(s1, S.Qh, S.Ql, t0 + 16 + li, lg, kf);
    mma3k(p0, S.Gh, S.Gl, t0 + li, lg, vf);             // dP = dO V^T
    mma3k(p1, S.Gh, S.Gl, t0 + 16 + li, lg, vf);
    const float4 l0 = *(const float4*)&S.lse[t0 + 4 * lg], l1 = *(const float4*)&S.lse[t0 + 16 + 4 * lg];
    const float4 e0 = *(const float4*)&S.dl[t0 + 4 * lg], e1 = *(const float4*)&S.dl[t0 + 16 + 4 * lg];
    const float ls[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
    const float dl[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
    float p[8], ds[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float sv = (j < 4 ? s0[j] : s1[j - 4]) * d.scale + bc[j] + kbias;
      p[j] = __expf(sv - ls[j]);
      ds[j] = p[j] * ((j < 4 ? p0[j] : p1[j - 4]) - dl[j]);
    }
    const HL pf = split8(p), df = split8(ds);
#pragma unroll
    for (int t = 0; t < OT; ++t) {
      mma3(dvt[t], frag_tr(S.Gh, S.Gl, t0, t0 + 16, 16 * t, li, lg), pf);   // dV^T += dO^T P
      mma3(dkt[t], frag_tr(S.Qh, S.Ql, t0, t0 + 16, 16 * t, li, lg), df);   // dK^T += Q^T dS
    }
  }
  SA_TL(6);
  if (krow < Lk) {
    float* dk = (float*)d.dk + (long)b * d.k_sb + (long)h * d.k_sh + (long)krow * d.k_sl;
    float* dv = (float*)d.dv + (long)b * d.v_sb + (long)h * d.v_sh + (long)krow * d.v_sl;
#pragma unroll
    for (int t = 0; t < OT; ++t) {
      *(float4*)(dk + 16 * t + 4 * lg) = make_float4(dkt[t][0] * d.scale, dkt[t][1] * d.scale, dkt[t][2] * d.scale, dkt[t][3] * d.scale);
      *(float4*)(dv + 16 * t + 4 * lg) = make_float4(dvt[t][0], dvt[t][1], dvt[t][2], dvt[t][3]);
    }
  }
  SA_TL(7);
}
#endif   // SA_DEVICE_ONLY
