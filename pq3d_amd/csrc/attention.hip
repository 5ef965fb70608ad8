// Fused masked multi-head attention for gfx950: forward (online softmax), backward as two recompute
// kernels (dQ: one workgroup per query chunk looping over keys; dK/dV: one workgroup per key chunk looping
// over queries) so no atomics are needed.  All three work on "swapped" 16x16 MFMA tiles so that softmax
// row statistics are per-lane scalars and the probability tile in C-layout can be fed straight back as
// the B operand of the second MFMA (no cross-lane transpose):
//   forward / dQ :  S^T[key][query] = K . Q^T      ->  O^T[dh][query]  += V^T[dh][key]  . P^T[key][query]
//   dK/dV        :  S  [query][key] = Q . K^T      ->  dV^T[dh][key]   += dO^T[dh][query] . P[query][key]
// Masks: key padding [B,Lk], 3-D [B,Lq,Lk] (head broadcast) with per-row "open" override, optional fp32
// additive bias [B,H,Lq,Lk], optional zero key (add_zero_attn) folded in as the initial state m=0, l=1.
#include "common.h"

#ifdef PQ3D_DEBUG_TIMING
__device__ long long pq3d_adbg[16];
#define ADBG(i, v) do { if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) pq3d_adbg[i] = (v); } while (0)
#define ANOW() __builtin_readcyclecounter()
extern "C" int pq3d_attn_debug_read(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pq3d_adbg), sizeof(long long) * 16); }
#else
#define ADBG(i, v)
#define ANOW() 0ll
#endif

#include <stdlib.h>

#include "attn_common.h"

bool pq3d_attn_bwd_resident_try(const pq3d_attn_desc& d, hipStream_t s);   // attn_resident.hip
bool pq3d_attn_fwd_resident_try(const pq3d_attn_desc& d, hipStream_t s);   // attn_resident.hip
int pq3d_attn_fwd_x3(const pq3d_attn_desc& d, hipStream_t s);               // attn_x3.hip
bool pq3d_attn_small_try(const pq3d_attn_desc& d, hipStream_t s, bool bwd);   // attn_small.hip
bool pq3d_attn_sa_try(const pq3d_attn_desc& d, hipStream_t s, bool bwd);      // attn_sa.hip
bool pq3d_attn_ca_try(const pq3d_attn_desc& d, hipStream_t s, bool bwd);      // attn_ca.hip
static int g_resident = 1, g_small = 1, g_resfwd = 1, g_sa = 1, g_ca = 1, g_resfwd_split = 1;
bool pq3d_resfwd_split_allowed() { return g_resfwd_split != 0; }
// bit 0: all-queries-resident backward, bit 1: small-sequence fp32 kernels, bit 2: all-keys-resident forward, bit 3: the
// split-bf16 MFMA self-attention kernels for compute type PQ3D_BF16X3, bit 4: the small bf16 cross-attention kernels
// bit 5: the all-keys-resident forward also for key-split calls (slices of <= 1024 keys of a longer scene) (all on by default)
extern "C" int pq3d_attn_resident(int enable) {
  const int old = g_resident | (g_small << 1) | (g_resfwd << 2) | (g_sa << 3) | (g_ca << 4) | (g_resfwd_split << 5);
  g_resident = enable & 1;
  g_small = (enable >> 1) & 1;
  g_resfwd = (enable >> 2) & 1;
  g_sa = (enable >> 3) & 1;
  g_ca = (enable >> 4) & 1;
  g_resfwd_split = (enable >> 5) & 1;   // the all-keys-resident forward also for key-split calls of > 1024 keys
  return old;
}

namespace {

// Software pipeline shared by the three attention kernels: LDS double buffering (ONE barrier per tile) and two
// register sets, so the global loads of tile t+3 are issued while tile t is being multiplied and are only waited for
// two iterations later.  load(tile, set) issues loads into register set `set`; store(set, buf) writes that set to LDS
// buffer `buf`; compute(tile, buf) consumes a buffer.  `set` / `buf` arrive as integral constants so every register
// array index is static.
template <int V> struct IC { static constexpr int value = V; };
template <typename LoadF, typename StoreF, typename ComputeF>
PQ_DEV void pipeline2(int n, LoadF load, StoreF store, ComputeF compute) {
  if (n <= 0) return;
  load(0, IC<0>{});
  if (n > 1) load(1, IC<1>{});
  store(IC<0>{}, IC<0>{});
  if (n > 2) load(2, IC<0>{});
  __syncthreads();
  for (int t = 0; t < n; t += 2) {
    compute(t, IC<0>{});
    if (t + 1 < n) { store(IC<1>{}, IC<1>{}); if (t + 3 < n) load(t + 3, IC<1>{}); }
    __syncthreads();
    if (t + 1 < n) {
      compute(t + 1, IC<1>{});
      if (t + 2 < n) { store(IC<0>{}, IC<0>{}); if (t + 4 < n) load(t + 4, IC<0>{}); }
      __syncthreads();
    }
  }
}

// Mask / bias fetch for the swapped layout (lane = query li; its keys in tile t are 4*lg + r, r = 0..3): everything
// that depends on WHETHER a mask or bias exists is a uniform branch around a block of loads (issued together), the
// per-element work is pure ALU.  Key-padding and 3-D mask bytes travel as one 32-bit word per (lane, tile).
struct MaskBias {
  uint32_t mw[4];     // byte r of mw[t] != 0  -> key (t, r) masked for this lane's query
  float bb[4][4];     // additive bias
};
// The 3-D mask words of one key block for this lane's query (4 x 4 key bytes).  Issued by the pipeline's load stage two
// to three tiles ahead of their use: as dependent loads inside the compute stage they cost +38 % (forward) / +63 %
// (backward) at config-4 shapes (tools/probes/attn_mask_cost_probe.py).  Caller guarantees d.mask != nullptr.
// With d.mask_bits (pq3d_mask_pack: bit words, open rows cleared) the 64 keys of a block are TWO words of the query's row
// instead of 16 bytes per lane: w[0], w[1] = the words, w[2] = w[3] = 0; fetch_mask_bias expands this lane's 4 bits per tile.
PQ_DEV void load_mask_words(uint32_t (&w)[4], const pq3d_attn_desc& d, int bm, int myq, int k0, int lg) {
  if (d.mask_bits) {   // uniform
    const int W = (d.Lk + 31) >> 5;
    const uint32_t* br = d.mask_bits + ((long)bm * d.Lq + min(myq, d.Lq - 1)) * W + (k0 >> 5);
    w[0] = br[0];
    w[1] = (k0 >> 5) + 1 < W ? br[1] : 0u;
    w[2] = 0; w[3] = 0;
    return;
  }
  const uint8_t* mr = d.mask + ((long)bm * d.Lq + min(myq, d.Lq - 1)) * d.Lk;
  const bool vec = (d.Lk & 3) == 0 && ((((uintptr_t)d.mask) & 3) == 0);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int gk0 = k0 + t * 16 + 4 * lg;
    if (vec) {
      w[t] = *(const uint32_t*)(mr + min(gk0, d.Lk - 4));
    } else {
      w[t] = 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) w[t] |= (uint32_t)mr[min(gk0 + r, d.Lk - 1)] << (8 * r);
    }
  }
}
template <bool MASK3>
PQ_DEV void fetch_mask_bias(MaskBias& mb, const pq3d_attn_desc& d, const uint8_t* kpm_s, const uint32_t (&mwords)[4],
                            int b, int h, int myq, bool qvalid, bool ro, int k0, int lg) {
#pragma unroll
  for (int t = 0; t < 4; ++t) mb.mw[t] = *(const uint32_t*)&kpm_s[t * 16 + 4 * lg];
  if constexpr (MASK3) {
    const bool use = qvalid && !ro;
    if (d.mask_bits) {   // uniform: bits 16 t' + 4 lg + r of word t / 2 -> byte flags r of mw[t]
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const uint32_t nib = (mwords[t >> 1] >> ((t & 1) * 16 + 4 * lg)) & 0xFu;
        const uint32_t by = (nib & 1u) | ((nib & 2u) << 7) | ((nib & 4u) << 14) | ((nib & 8u) << 21);
        mb.mw[t] |= use ? by : 0u;
      }
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) mb.mw[t] |= use ? mwords[t] : 0u;   // keys >= Lk are already masked through kpm_s
    }
  }
  if (d.bias) {
    const float* br = d.bias + (((long)b * d.H + h) * d.Lq + min(myq, d.Lq - 1)) * d.Lk;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) mb.bb[t][r] = br[min(k0 + t * 16 + 4 * lg + r, d.Lk - 1)];
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) mb.bb[t][r] = 0.f;
  }
}

// ------------------------------------------------------------------------------------------------ forward
template <typename CT, int DH, int NW, bool DROP, bool MASK3>
__global__ __launch_bounds__(NW * 64) void attn_fwd_kernel(const pq3d_attn_desc d) {
  ATTN_KARG_PIN(d);
  typedef AT<CT, DH> A;
  constexpr bool TRR = sizeof(CT) == 2;  // bf16: row-major V + transposing reads; f32: transposed LDS copy
  constexpr int KSZ = KB * A::LDR, VSZ = TRR ? KB * A::LDR : DH * A::LDT;
  __shared__ __attribute__((aligned(16))) CT Kbuf[2 * KSZ];
  __shared__ __attribute__((aligned(16))) CT Vbuf[2 * VSZ];
  __shared__ __attribute__((aligned(4))) uint8_t kpm_buf[2 * KB];
  __shared__ uint8_t blk_flag[MAXKB];
  __shared__ uint16_t blk_act[MAXKB];
  __shared__ int blk_cnt;
  constexpr int nthreads = NW * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int KS = d.ksplit > 1 ? d.ksplit : 1;
  const WgXyz wg = attn_wg_xyz(KS);
  const int b = wg.b, h = wg.h;
  const int split = wg.x % KS, qchunk = wg.x / KS;
  const int q0 = (qchunk * NW + wave) * 16;
  const int myq = q0 + li;
  const bool wave_active = q0 < d.Lq, qvalid = myq < d.Lq;

  if (A::DHK > DH) { zero_lds<CT, DH>(Kbuf, 2 * KSZ, tid, nthreads); __syncthreads(); }
  const int nact = active_key_blocks(d.kpm ? d.kpm + (long)b * d.Lk : nullptr, d.Lk, blk_flag, blk_act, &blk_cnt, tid,
                                     nthreads);
  const int ntot = nact < 0 ? -nact : nact;
  const int t_lo = (int)((long)ntot * split / KS), t_hi = (int)((long)ntot * (split + 1) / KS);   // this split's slice
  auto kblock = [&](int t) { return nact < 0 ? t_lo + t : (int)blk_act[t_lo + t]; };

  u32x4 qf[A::NS];
  row_frags<CT, DH>(qf, d.q, (long)b * d.q_sb + (long)min(myq, d.Lq - 1) * d.q_sl + (long)h * d.q_sh, lg);

  // m is uniform over the 4 lanes of a query; l is a PER-LANE partial row sum, reduced once after the loop.
  // The zero key (add_zero_attn) is the initial state of split 0 only.
  const bool zero0 = d.zero_attn && split == 0;
  float m = zero0 ? 0.f : -1e30f, l = (zero0 && lg == 0) ? 1.f : 0.f;
  f32x4 acc[A::MT];
#pragma unroll
  for (int mt = 0; mt < A::MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int bm = d.mask_bmod > 0 ? b % d.mask_bmod : b;
  const bool ro = (qvalid && d.row_open) ? d.row_open[(long)bm * d.Lq + myq] != 0 : false;
  const long koff = (long)b * d.k_sb + (long)h * d.k_sh, voff = (long)b * d.v_sb + (long)h * d.v_sh;
  // attention-probability dropout: this lane's query is one row of the [B*H*Lq, Lk] site
  DropState dst;
  uint32_t drow = 0;
  if constexpr (DROP) {
    dst = drop_init(d.drop, d.drop_bmod > 0 ? b / d.drop_bmod : 0, d.Lk);
    drow = (uint32_t)(((long)(d.drop_bmod > 0 ? b % d.drop_bmod : b) * d.H + h) * d.Lq + min(myq, d.Lq - 1));
  }

  TileRegs<CT, DH, KB, nthreads> kr[2], vr[2];
  uint8_t kpm_r[2] = {1, 1};
  // 3-D mask words in flight (per register set); those of the tiles sitting in the two LDS buffers are parked in LDS
  // too (each lane reads back exactly what it wrote).  MASK3 is a template parameter so that launches without a 3-D
  // mask keep their register / LDS budget (occupancy).
  uint32_t mask_r[MASK3 ? 2 : 1][4];
#pragma unroll
  for (int t = 0; t < 4; ++t) { mask_r[0][t] = 0; if constexpr (MASK3) mask_r[1][t] = 0; }
  __shared__ uint32_t mask_lds[MASK3 ? 2 * 4 * nthreads : 1];
  auto load = [&](int t, auto set) {
    constexpr int S = decltype(set)::value;
    const int k0 = kblock(t) * KB;
    kr[S].load(d.k, koff, d.k_sl, k0, d.Lk, tid);
    vr[S].load(d.v, voff, d.v_sl, k0, d.Lk, tid);
    if (tid < KB) kpm_r[S] = (k0 + tid < d.Lk) ? (d.kpm ? d.kpm[(long)b * d.Lk + k0 + tid] : 0) : 1;
    if constexpr (MASK3) load_mask_words(mask_r[S], d, bm, myq, k0, lg);
  };
  auto store = [&](auto set, auto buf) {
    constexpr int S = decltype(set)::value, Bf = decltype(buf)::value;
    kr[S].store(Kbuf + Bf * KSZ, nullptr, 0, tid);
    if (TRR) vr[S].store(Vbuf + Bf * VSZ, nullptr, 0, tid);
    else vr[S].store(nullptr, Vbuf + Bf * VSZ, A::LDT, tid);
    if (tid < KB) kpm_buf[Bf * KB + tid] = kpm_r[S];
    if constexpr (MASK3) {
#pragma unroll
      for (int t = 0; t < 4; ++t) mask_lds[(Bf * 4 + t) * nthreads + tid] = mask_r[S][t];
    }
  };
  auto compute = [&](int t, auto buf) {
    constexpr int Bf = decltype(buf)::value;
    if (!wave_active) return;
    const CT* Ks = Kbuf + Bf * KSZ;
    const CT* Vs = Vbuf + Bf * VSZ;
    const int k0 = kblock(t) * KB;
    float p[4][4];
    float mx = -INFINITY;
    MaskBias mb;
    uint32_t mwords[4] = {0, 0, 0, 0};
    if constexpr (MASK3) {
#pragma unroll
      for (int t = 0; t < 4; ++t) mwords[t] = mask_lds[(Bf * 4 + t) * nthreads + tid];
    }
    fetch_mask_bias<MASK3>(mb, d, kpm_buf + Bf * KB, mwords, b, h, myq, qvalid, ro, k0, lg);
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      f32x4 sc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < A::NS; ++st) Mma<CT>::mma(sc, rfrag<CT>(&Ks[(tt * 16 + li) * A::LDR], st, lg), qf[st]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float x = ((mb.mw[tt] >> (8 * r)) & 0xffu) ? -INFINITY : sc[r] * d.scale + mb.bb[tt][r];
        p[tt][r] = x;
        mx = fmaxf(mx, x);
      }
    }
    mx = group_max(mx);
    const float m_new = fmaxf(m, mx);
    const float alpha = fexp<CT>(m - m_new);
    float rs = 0.f;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        p[tt][r] = fexp<CT>(p[tt][r] - m_new);
        rs += p[tt][r];
      }
    l = l * alpha + rs;   // per-lane partial (alpha is uniform over the query's 4 lanes)
    m = m_new;
    if constexpr (DROP) {   // the softmax denominator keeps every key; only the value contraction sees the mask
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
    u32x4 pf[PackP<CT, 4>::STEPS];
    PackP<CT, 4>::run(p, pf);
#pragma unroll
    for (int mt = 0; mt < A::MT; ++mt) {
      acc[mt] *= alpha;
#pragma unroll
      for (int u = 0; u < PackP<CT, 4>::STEPS; ++u)
        Mma<CT>::mma(acc[mt], tfrag_any<CT>(Vs, A::LDR, Vs, A::LDT, u, mt, li, lg), pf[u]);
    }
  };
  pipeline2(t_hi - t_lo, load, store, compute);
  l = group_sum(l);

  if (!qvalid) return;
  if (KS == 1) {
    const float inv = (DROP ? dst.scale : 1.f) / l;
    const long ooff = (long)b * d.o_sb + (long)myq * d.o_sl + (long)h * d.o_sh;
#pragma unroll
    for (int mt = 0; mt < A::MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) store_elem(d.o, d.dt, ooff + mt * 16 + 4 * lg + r, acc[mt][r] * inv);
    if (lg == 0) d.lse[((long)b * d.H + h) * d.Lq + myq] = m + logf(l);
  } else {   // partial softmax state (un-normalised O, running max, row sum) for the combine kernel
    const long rows = (long)d.B * d.H * d.Lq, ridx = (((long)split * d.B + b) * d.H + h) * d.Lq + myq;
    float* po = d.ws + ridx * DH;
#pragma unroll
    for (int mt = 0; mt < A::MT; ++mt)
      *(float4*)(po + mt * 16 + 4 * lg) = make_float4(acc[mt][0], acc[mt][1], acc[mt][2], acc[mt][3]);
    if (lg == 0) {
      d.ws[(long)KS * rows * DH + ridx] = m;
      d.ws[(long)KS * rows * (DH + 1) + ridx] = l;
    }
  }
}

// merges the `ksplit` partial states of attn_fwd_kernel: one thread per (row, 4-channel group)
template <int DH>
__global__ void attn_fwd_combine_kernel(const pq3d_attn_desc d) {
  ATTN_KARG_PIN(d);
  const long rows = (long)d.B * d.H * d.Lq;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * (DH / 4)) return;
  const long row = idx / (DH / 4);
  const int c0 = (int)(idx % (DH / 4)) * 4;
  const int KS = d.ksplit;
  const float* pm = d.ws + (long)KS * rows * DH;
  const float* pl = d.ws + (long)KS * rows * (DH + 1);
  float M = -INFINITY;
  for (int s = 0; s < KS; ++s) M = fmaxf(M, pm[s * rows + row]);
  float Lsum = 0.f;
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s = 0; s < KS; ++s) {
    const float w = expf(pm[s * rows + row] - M);
    Lsum += pl[s * rows + row] * w;
    const float4 p = *(const float4*)(d.ws + (s * rows + row) * DH + c0);
    o.x += p.x * w; o.y += p.y * w; o.z += p.z * w; o.w += p.w * w;
  }
  const float inv = ((d.drop.p > 0.f && d.drop.seed) ? 1.f / (1.f - d.drop.p) : 1.f) / Lsum;
  const int q = (int)(row % d.Lq), h = (int)((row / d.Lq) % d.H), b = (int)(row / ((long)d.Lq * d.H));
  const long ooff = (long)b * d.o_sb + (long)q * d.o_sl + (long)h * d.o_sh + c0;
  store_elem(d.o, d.dt, ooff, o.x * inv); store_elem(d.o, d.dt, ooff + 1, o.y * inv);
  store_elem(d.o, d.dt, ooff + 2, o.z * inv); store_elem(d.o, d.dt, ooff + 3, o.w * inv);
  if (d.o_bf)   // split-bf16 forward (attn_x3.hip): the bf16 copy of the output the backward reads
    *(u32x2*)((bf16_t*)d.o_bf + ooff) = (u32x2){pack_bf2(o.x * inv, o.y * inv), pack_bf2(o.z * inv, o.w * inv)};
  if (c0 == 0) d.lse[row] = M + logf(Lsum);
}

// ------------------------------------------------------------------------------------------------ delta
template <typename CT, int DH>
__global__ void attn_delta_kernel(const pq3d_attn_desc d) {
  typedef AT<CT, DH> A;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)d.B * d.H * d.Lq;
  if (idx >= total) return;
  const int q = idx % d.Lq, h = (idx / d.Lq) % d.H, b = idx / ((long)d.Lq * d.H);
  const long off = (long)b * d.o_sb + (long)q * d.o_sl + (long)h * d.o_sh;
  u32x4 ro[A::CPR], rd[A::CPR];
#pragma unroll
  for (int c = 0; c < A::CPR; ++c) {
    ro[c] = *(const u32x4*)((const CT*)d.o + off + c * A::EPL);
    rd[c] = *(const u32x4*)((const CT*)d.dout + off + c * A::EPL);
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < A::CPR; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (sizeof(CT) == 2) {
        s += __uint_as_float(ro[c][j] << 16) * __uint_as_float(rd[c][j] << 16);
        s += __uint_as_float(ro[c][j] & 0xffff0000u) * __uint_as_float(rd[c][j] & 0xffff0000u);
      } else {
        s += __uint_as_float(ro[c][j]) * __uint_as_float(rd[c][j]);
      }
    }
  d.delta[idx] = s;
}

// ------------------------------------------------------------------------------------------------ dQ (+ dbias)
template <typename CT, int DH, int NW, bool DROP, bool MASK3>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dq_kernel(const pq3d_attn_desc d) {
  ATTN_KARG_PIN(d);
  ATTN_KARG_PIN_BWD(d);
  typedef AT<CT, DH> A;
  constexpr bool TRR = sizeof(CT) == 2;
  constexpr int KSZ = KB * A::LDR, TSZ = TRR ? 8 : DH * A::LDT;
  __shared__ __attribute__((aligned(16))) CT Kbuf[2 * KSZ];
  __shared__ __attribute__((aligned(16))) CT Vbuf[2 * KSZ];
  __shared__ __attribute__((aligned(16))) CT Ktbuf[2 * TSZ];
  __shared__ __attribute__((aligned(4))) uint8_t kpm_buf[2 * KB];
  __shared__ uint8_t blk_flag[MAXKB];
  __shared__ uint16_t blk_act[MAXKB];
  __shared__ int blk_cnt;
  constexpr int nthreads = NW * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int KS = d.ksplit > 1 ? d.ksplit : 1;
  const WgXyz wg = attn_wg_xyz(KS);
  const int b = wg.b, h = wg.h;
  const int split = wg.x % KS, qchunk = wg.x / KS;
  const int q0 = (qchunk * NW + wave) * 16;
  const int myq = q0 + li;
  const bool wave_active = q0 < d.Lq, qvalid = myq < d.Lq;

  if (A::DHK > DH) {
    zero_lds<CT, DH>(Kbuf, 2 * KSZ, tid, nthreads);
    zero_lds<CT, DH>(Vbuf, 2 * KSZ, tid, nthreads);
    __syncthreads();
  }
  const int nact = active_key_blocks(d.kpm ? d.kpm + (long)b * d.Lk : nullptr, d.Lk, blk_flag, blk_act, &blk_cnt, tid,
                                     nthreads);
  const int ntot = nact < 0 ? -nact : nact;
  const int t_lo = (int)((long)ntot * split / KS), t_hi = (int)((long)ntot * (split + 1) / KS);
  auto kblock = [&](int t) { return nact < 0 ? t_lo + t : (int)blk_act[t_lo + t]; };
  u32x4 qf[A::NS], dof[A::NS], of[A::NS];
  const int cq = min(myq, d.Lq - 1);
  row_frags<CT, DH>(qf, d.q, (long)b * d.q_sb + (long)cq * d.q_sl + (long)h * d.q_sh, lg);
  row_frags<CT, DH>(dof, d.dout, (long)b * d.o_sb + (long)cq * d.o_sl + (long)h * d.o_sh, lg);
  row_frags<CT, DH>(of, d.o, (long)b * d.o_sb + (long)cq * d.o_sl + (long)h * d.o_sh, lg);
  const long sidx = ((long)b * d.H + h) * d.Lq + myq;
  const float L = qvalid ? d.lse[sidx] : INFINITY;
  // delta = rowsum(dO * O), fused here (the lane's dO/O fragments cover its dh slice; reduce over the 4 lane groups)
  // and published for the dK/dV kernel that runs after this one
  float Dl = 0.f;
#pragma unroll
  for (int st = 0; st < A::NS; ++st)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (sizeof(CT) == 2) {
        Dl += __uint_as_float(dof[st][j] << 16) * __uint_as_float(of[st][j] << 16);
        Dl += __uint_as_float(dof[st][j] & 0xffff0000u) * __uint_as_float(of[st][j] & 0xffff0000u);
      } else {
        Dl += __uint_as_float(dof[st][j]) * __uint_as_float(of[st][j]);
      }
    }
  Dl = group_sum(Dl);
  if (qvalid && lg == 0 && split == 0) d.delta[sidx] = Dl;

  f32x4 acc[A::MT];
#pragma unroll
  for (int mt = 0; mt < A::MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int bm = d.mask_bmod > 0 ? b % d.mask_bmod : b;
  const bool ro = (qvalid && d.row_open) ? d.row_open[(long)bm * d.Lq + myq] != 0 : false;
  float* dbrow = d.dbias ? d.dbias + (((long)b * d.H + h) * d.Lq + min(myq, d.Lq - 1)) * d.Lk : nullptr;
  DropState dst;
  uint32_t drow = 0;
  if constexpr (DROP) {
    dst = drop_init(d.drop, d.drop_bmod > 0 ? b / d.drop_bmod : 0, d.Lk);
    drow = (uint32_t)(((long)(d.drop_bmod > 0 ? b % d.drop_bmod : b) * d.H + h) * d.Lq + min(myq, d.Lq - 1));
  }
  const long koff = (long)b * d.k_sb + (long)h * d.k_sh, voff = (long)b * d.v_sb + (long)h * d.v_sh;

  TileRegs<CT, DH, KB, nthreads> kr[2], vr[2];
  uint8_t kpm_r[2] = {1, 1};
  // 3-D mask words in flight (per register set); those of the tiles sitting in the two LDS buffers are parked in LDS
  // too (each lane reads back exactly what it wrote).  MASK3 is a template parameter so that launches without a 3-D
  // mask keep their register / LDS budget (occupancy).
  uint32_t mask_r[MASK3 ? 2 : 1][4];
#pragma unroll
  for (int t = 0; t < 4; ++t) { mask_r[0][t] = 0; if constexpr (MASK3) mask_r[1][t] = 0; }
  __shared__ uint32_t mask_lds[MASK3 ? 2 * 4 * nthreads : 1];
  auto load = [&](int t, auto set) {
    constexpr int S = decltype(set)::value;
    const int k0 = kblock(t) * KB;
    kr[S].load(d.k, koff, d.k_sl, k0, d.Lk, tid);
    vr[S].load(d.v, voff, d.v_sl, k0, d.Lk, tid);
    if (tid < KB) kpm_r[S] = (k0 + tid < d.Lk) ? (d.kpm ? d.kpm[(long)b * d.Lk + k0 + tid] : 0) : 1;
    if constexpr (MASK3) load_mask_words(mask_r[S], d, bm, myq, k0, lg);
  };
  auto store = [&](auto set, auto buf) {
    constexpr int S = decltype(set)::value, Bf = decltype(buf)::value;
    kr[S].store(Kbuf + Bf * KSZ, TRR ? nullptr : Ktbuf + Bf * TSZ, A::LDT, tid);
    vr[S].store(Vbuf + Bf * KSZ, nullptr, 0, tid);
    if (tid < KB) kpm_buf[Bf * KB + tid] = kpm_r[S];
    if constexpr (MASK3) {
#pragma unroll
      for (int t = 0; t < 4; ++t) mask_lds[(Bf * 4 + t) * nthreads + tid] = mask_r[S][t];
    }
  };
  auto compute = [&](int t, auto buf) {
    constexpr int Bf = decltype(buf)::value;
    if (!wave_active) return;
    const CT* Ks = Kbuf + Bf * KSZ;
    const CT* Vs = Vbuf + Bf * KSZ;
    const CT* Kt = Ktbuf + Bf * TSZ;
    const int k0 = kblock(t) * KB;
    float ds[4][4];
    MaskBias mb;
    uint32_t mwords[4] = {0, 0, 0, 0};
    if constexpr (MASK3) {
#pragma unroll
      for (int t = 0; t < 4; ++t) mwords[t] = mask_lds[(Bf * 4 + t) * nthreads + tid];
    }
    fetch_mask_bias<MASK3>(mb, d, kpm_buf + Bf * KB, mwords, b, h, myq, qvalid, ro, k0, lg);
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      f32x4 sc = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < A::NS; ++st) {
        Mma<CT>::mma(sc, rfrag<CT>(&Ks[(tt * 16 + li) * A::LDR], st, lg), qf[st]);
        Mma<CT>::mma(dp, rfrag<CT>(&Vs[(tt * 16 + li) * A::LDR], st, lg), dof[st]);
      }
      if constexpr (DROP) {   // dP = keep * dPd / (1-p); delta = rowsum(dO * O) already includes the mask
        const uint32_t cp = (uint32_t)(k0 + tt * 16 + 4 * lg) >> 1;
        const uint32_t w0 = drop_word(dst, drow, cp), w1 = drop_word(dst, drow, cp + 1);
        dp[0] = drop_keep_lo(dst, w0) ? dp[0] * dst.scale : 0.f;
        dp[1] = drop_keep_hi(dst, w0) ? dp[1] * dst.scale : 0.f;
        dp[2] = drop_keep_lo(dst, w1) ? dp[2] * dst.scale : 0.f;
        dp[3] = drop_keep_hi(dst, w1) ? dp[3] * dst.scale : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool masked = ((mb.mw[tt] >> (8 * r)) & 0xffu) != 0;
        const float pr = masked ? 0.f : fexp<CT>(sc[r] * d.scale + mb.bb[tt][r] - L);
        ds[tt][r] = pr * (dp[r] - Dl);
      }
    }
    if (dbrow && qvalid) {   // uniform on dbias; 4 consecutive keys per (lane, tile)
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gk = k0 + tt * 16 + 4 * lg + r;
          if (gk < d.Lk) dbrow[gk] = ds[tt][r];
        }
    }
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) ds[tt][r] *= d.scale;
    u32x4 dsf[PackP<CT, 4>::STEPS];
    PackP<CT, 4>::run(ds, dsf);
#pragma unroll
    for (int mt = 0; mt < A::MT; ++mt)
#pragma unroll
      for (int u = 0; u < PackP<CT, 4>::STEPS; ++u)
        Mma<CT>::mma(acc[mt], tfrag_any<CT>(Ks, A::LDR, Kt, A::LDT, u, mt, li, lg), dsf[u]);
  };
  pipeline2(t_hi - t_lo, load, store, compute);
  if (dbrow && nact >= 0 && qvalid && wave_active) {   // bias gradient of skipped (fully padded) key blocks is zero
    const int nkb = (d.Lk + KB - 1) / KB;
    for (int kb = 0; kb < nkb; ++kb)
      if (!blk_flag[kb])
        for (int j = lg; j < KB; j += 4)
          if (kb * KB + j < d.Lk) dbrow[kb * KB + j] = 0.f;
  }

  if (!qvalid) return;
  if (KS == 1) {
    const long off = (long)b * d.q_sb + (long)myq * d.q_sl + (long)h * d.q_sh;
#pragma unroll
    for (int mt = 0; mt < A::MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) store_elem(d.dq, d.dt, off + mt * 16 + 4 * lg + r, acc[mt][r]);
  } else {   // partial dQ of this key slice, summed by attn_dq_combine_kernel
    float* po = d.ws + ((((long)split * d.B + b) * d.H + h) * d.Lq + myq) * DH;
#pragma unroll
    for (int mt = 0; mt < A::MT; ++mt)
      *(float4*)(po + mt * 16 + 4 * lg) = make_float4(acc[mt][0], acc[mt][1], acc[mt][2], acc[mt][3]);
  }
}

template <int DH>
__global__ void attn_dq_combine_kernel(const pq3d_attn_desc d) {
  ATTN_KARG_PIN(d);
  ATTN_KARG_PIN_BWD(d);
  const long rows = (long)d.B * d.H * d.Lq;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * (DH / 4)) return;
  const long row = idx / (DH / 4);
  const int c0 = (int)(idx % (DH / 4)) * 4;
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s = 0; s < d.ksplit; ++s) {
    const float4 p = *(const float4*)(d.ws + (s * rows + row) * DH + c0);
    o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
  }
  const int q = (int)(row % d.Lq), h = (int)((row / d.Lq) % d.H), b = (int)(row / ((long)d.Lq * d.H));
  const long off = (long)b * d.q_sb + (long)q * d.q_sl + (long)h * d.q_sh + c0;
  store_elem(d.dq, d.dt, off, o.x); store_elem(d.dq, d.dt, off + 1, o.y);
  store_elem(d.dq, d.dt, off + 2, o.z); store_elem(d.dq, d.dt, off + 3, o.w);
}

// ------------------------------------------------------------------------------------------------ dK, dV
template <typename CT, int DH, bool DROP, bool MASK3, int NWK>
__global__ __launch_bounds__(NWK * 64) void attn_bwd_dkv_kernel(const pq3d_attn_desc d) {
  ATTN_KARG_PIN(d);
  ATTN_KARG_PIN_BWD(d);
  typedef AT<CT, DH> A;
  constexpr bool TRR = sizeof(CT) == 2;
  constexpr int QSZ = QB * A::LDR, TSZ = TRR ? 8 : DH * A::LDQ;
  __shared__ __attribute__((aligned(16))) CT Qbuf[2 * QSZ];
  __shared__ __attribute__((aligned(16))) CT dObuf[2 * QSZ];
  __shared__ __attribute__((aligned(16))) CT Qtbuf[2 * TSZ];
  __shared__ __attribute__((aligned(16))) CT dOtbuf[2 * TSZ];
  __shared__ float Lbuf[2 * QB], Dbuf[2 * QB];
  __shared__ uint8_t robuf[2 * QB];
  constexpr int MLD = NWK * 16 + 8;   // padded row of the staged 3-D mask tile [QB queries][64 keys] (bytes)
  __shared__ __attribute__((aligned(8))) uint8_t mbuf[MASK3 ? 2 * QB * MLD : 8];
  constexpr int nthreads = NWK * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const WgXyz wg = attn_wg_xyz((int)gridDim.x);
  const int b = wg.b, h = wg.h;
  const int key = (wg.x * NWK + wave) * 16 + li;
  const bool kvalid = key < d.Lk;

  if (A::DHK > DH) {
    zero_lds<CT, DH>(Qbuf, 2 * QSZ, tid, nthreads);
    zero_lds<CT, DH>(dObuf, 2 * QSZ, tid, nthreads);
    __syncthreads();
  }
  u32x4 kf[A::NS], vf[A::NS];
  const int ckey = min(key, d.Lk - 1);
  row_frags<CT, DH>(kf, d.k, (long)b * d.k_sb + (long)ckey * d.k_sl + (long)h * d.k_sh, lg);
  row_frags<CT, DH>(vf, d.v, (long)b * d.v_sb + (long)ckey * d.v_sl + (long)h * d.v_sh, lg);
  const bool kmasked = kvalid ? (d.kpm ? d.kpm[(long)b * d.Lk + key] != 0 : false) : true;
  if (__syncthreads_and(kmasked)) {   // whole key chunk padded: its dK, dV are exactly zero
    if (kvalid) {
      const long ko = (long)b * d.k_sb + (long)key * d.k_sl + (long)h * d.k_sh;
      const long vo = (long)b * d.v_sb + (long)key * d.v_sl + (long)h * d.v_sh;
      for (int c = lg; c < DH; c += 4) { store_elem(d.dk, d.dt, ko + c, 0.f); store_elem(d.dv, d.dt, vo + c, 0.f); }
    }
    return;
  }

  f32x4 accK[A::MT], accV[A::MT];
#pragma unroll
  for (int mt = 0; mt < A::MT; ++mt) {
    accK[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    accV[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const long qoff = (long)b * d.q_sb + (long)h * d.q_sh, ooff = (long)b * d.o_sb + (long)h * d.o_sh;
  const long sbase = ((long)b * d.H + h) * d.Lq;
  const int bm = d.mask_bmod > 0 ? b % d.mask_bmod : b;
  DropState dst;
  uint32_t drow0 = 0;
  if constexpr (DROP) {
    dst = drop_init(d.drop, d.drop_bmod > 0 ? b / d.drop_bmod : 0, d.Lk);
    drow0 = (uint32_t)(((long)(d.drop_bmod > 0 ? b % d.drop_bmod : b) * d.H + h) * d.Lq);
  }

  TileRegs<CT, DH, QB, nthreads> qr[2], dor[2];
  float l_r[2] = {INFINITY, INFINITY}, d_r[2] = {0.f, 0.f};
  uint8_t ro_r[2] = {0, 0};
  // 3-D mask tile of this key chunk for the 32 queries of a tile: thread (row = tid / (2 NWK), 8-key chunk = tid % (2 NWK)) moves 8
  // bytes global -> register -> LDS with the Q / dO tiles (prefetched 2-3 tiles ahead) instead of 8 dependent 1-byte
  // loads per lane inside the compute stage
  u32x2 mk_r[MASK3 ? 2 : 1];
  const int key0 = wg.x * NWK * 16;
  const bool mvec = MASK3 && (d.Lk & 7) == 0 && d.Lk >= 8 && ((((uintptr_t)d.mask) & 7) == 0);
  auto load = [&](int t, auto set) {
    constexpr int S = decltype(set)::value;
    const int qb = t * QB;
    qr[S].load(d.q, qoff, d.q_sl, qb, d.Lq, tid);
    dor[S].load(d.dout, ooff, d.o_sl, qb, d.Lq, tid);
    if constexpr (MASK3) {
      const uint8_t* mr = d.mask + ((long)bm * d.Lq + min(qb + tid / (NWK * 2), d.Lq - 1)) * d.Lk;
      const int kc = key0 + (tid % (NWK * 2)) * 8;
      if (mvec) {
        mk_r[S] = *(const u32x2*)(mr + min(kc, d.Lk - 8));   // clamped: keys >= Lk are masked through kvalid anyway
      } else {
        uint32_t w2[2] = {0, 0};   // rare path (Lk not a multiple of 8): rolled loop, keeps the register count down
#pragma unroll 1
        for (int j = 0; j < 8; ++j) w2[j >> 2] |= (uint32_t)mr[min(kc + j, d.Lk - 1)] << (8 * (j & 3));
        mk_r[S] = (u32x2){w2[0], w2[1]};
      }
    }
    if (tid < QB) {
      const int gq = qb + tid;
      l_r[S] = gq < d.Lq ? d.lse[sbase + gq] : INFINITY;
      d_r[S] = gq < d.Lq ? d.delta[sbase + gq] : 0.f;
      ro_r[S] = (gq < d.Lq && d.row_open) ? d.row_open[(long)bm * d.Lq + gq] : 0;
    }
  };
  auto store = [&](auto set, auto buf) {
    constexpr int S = decltype(set)::value, Bf = decltype(buf)::value;
    qr[S].store(Qbuf + Bf * QSZ, TRR ? nullptr : Qtbuf + Bf * TSZ, A::LDQ, tid);
    dor[S].store(dObuf + Bf * QSZ, TRR ? nullptr : dOtbuf + Bf * TSZ, A::LDQ, tid);
    if (tid < QB) { Lbuf[Bf * QB + tid] = l_r[S]; Dbuf[Bf * QB + tid] = d_r[S]; robuf[Bf * QB + tid] = ro_r[S]; }
    if constexpr (MASK3) *(u32x2*)&mbuf[(Bf * QB + tid / (NWK * 2)) * MLD + (tid % (NWK * 2)) * 8] = mk_r[S];
  };
  auto compute = [&](int t, auto buf) {
    constexpr int Bf = decltype(buf)::value;
    const CT* Qs = Qbuf + Bf * QSZ;
    const CT* dOs = dObuf + Bf * QSZ;
    const CT* Qt = Qtbuf + Bf * TSZ;
    const CT* dOt = dOtbuf + Bf * TSZ;
    const float* Ls = Lbuf + Bf * QB;
    const float* Ds = Dbuf + Bf * QB;
    const uint8_t* ro_s = robuf + Bf * QB;
    const int qb = t * QB;
    float pt[2][4], dsk[2][4];
    // mask / bias for this lane's key against the 8 queries (tt, r) it sees: uniform branches around grouped loads
    bool mk[2][4];
    float bb[2][4];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { mk[tt][r] = kmasked; bb[tt][r] = 0.f; }
    if constexpr (MASK3) {
      const uint8_t* ms = mbuf + Bf * QB * MLD + wave * 16 + li;   // this lane's key column of the staged tile
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ql = tt * 16 + 4 * lg + r;
          mk[tt][r] |= (!ro_s[ql]) && (ms[ql * MLD] != 0);
        }
    }
    if (d.bias) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) bb[tt][r] = d.bias[(sbase + min(qb + tt * 16 + 4 * lg + r, d.Lq - 1)) * d.Lk + ckey];
    }
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      f32x4 sc = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < A::NS; ++st) {
        Mma<CT>::mma(sc, rfrag<CT>(&Qs[(tt * 16 + li) * A::LDR], st, lg), kf[st]);
        Mma<CT>::mma(dp, rfrag<CT>(&dOs[(tt * 16 + li) * A::LDR], st, lg), vf[st]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ql = tt * 16 + 4 * lg + r;
        const float pr = mk[tt][r] ? 0.f : fexp<CT>(sc[r] * d.scale + bb[tt][r] - Ls[ql]);   // Ls = +inf past Lq
        float kc = 1.f;   // dropout factor of (query, key): 0 or 1/(1-p)
        if constexpr (DROP) kc = drop_keep(dst, drow0 + (uint32_t)min(qb + ql, d.Lq - 1), (uint32_t)ckey) ? dst.scale : 0.f;
        pt[tt][r] = pr * kc;
        dsk[tt][r] = pr * (dp[r] * kc - Ds[ql]) * d.scale;
      }
    }
    u32x4 pf[PackP<CT, 2>::STEPS], dsf[PackP<CT, 2>::STEPS];
    PackP<CT, 2>::run(pt, pf);
    PackP<CT, 2>::run(dsk, dsf);
#pragma unroll
    for (int mt = 0; mt < A::MT; ++mt)
#pragma unroll
      for (int u = 0; u < PackP<CT, 2>::STEPS; ++u) {
        Mma<CT>::mma(accV[mt], tfrag_any<CT>(dOs, A::LDR, dOt, A::LDQ, u, mt, li, lg), pf[u]);
        Mma<CT>::mma(accK[mt], tfrag_any<CT>(Qs, A::LDR, Qt, A::LDQ, u, mt, li, lg), dsf[u]);
      }
  };
  pipeline2((d.Lq + QB - 1) / QB, load, store, compute);

  if (kvalid) {
    const long ko = (long)b * d.k_sb + (long)key * d.k_sl + (long)h * d.k_sh;
    const long vo = (long)b * d.v_sb + (long)key * d.v_sl + (long)h * d.v_sh;
#pragma unroll
    for (int mt = 0; mt < A::MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        store_elem(d.dk, d.dt, ko + mt * 16 + 4 * lg + r, accK[mt][r]);
        store_elem(d.dv, d.dt, vo + mt * 16 + 4 * lg + r, accV[mt][r]);
      }
  }
}

__global__ void mask_row_all_kernel(const uint8_t* mask, uint8_t* row_open, long rows, long Lk) {
  const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const uint8_t* p = mask + row * Lk;
  int all = 1;
  if ((Lk & 15) == 0 && ((((uintptr_t)mask) & 15) == 0)) {   // 16 mask bytes per lane and load (a byte each: 14 us for 10 MB)
    for (long j = (long)lane * 16; j < Lk; j += 64 * 16) {
      const uint4 w = *(const uint4*)(p + j);
      // every byte non-zero <=> no zero byte in the word: (x - 0x01010101) & ~x & 0x80808080 == 0
      auto nz = [](unsigned x) { return (((x - 0x01010101u) & ~x & 0x80808080u) == 0u); };
      all &= (nz(w.x) && nz(w.y) && nz(w.z) && nz(w.w)) ? 1 : 0;
    }
  } else {
    for (long j = lane; j < Lk; j += 64) all &= (p[j] != 0);
  }
  all = __all(all);
  if (lane == 0) row_open[row] = all ? 1 : 0;
}

// mask bytes -> row_open flag + bit words (row_open folded in): one wave per row, two passes over the row (the second one
// hits the cache): pass 1 = is every byte non-zero, pass 2 = 32 bytes -> one word per lane step
__global__ void mask_pack_kernel(const uint8_t* __restrict__ mask, uint8_t* __restrict__ row_open, uint32_t* __restrict__ bits,
                                 long rows, long Lk) {
  const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const uint8_t* p = mask + row * Lk;
  const bool vec = (Lk & 15) == 0 && ((((uintptr_t)mask) & 15) == 0);
  auto nz = [](unsigned x) { return (((x - 0x01010101u) & ~x & 0x80808080u) == 0u); };
  int all = 1;
  if (vec) {
    for (long j = (long)lane * 16; j < Lk; j += 64 * 16) {
      const uint4 w = *(const uint4*)(p + j);
      all &= (nz(w.x) && nz(w.y) && nz(w.z) && nz(w.w)) ? 1 : 0;
    }
  } else {
    for (long j = lane; j < Lk; j += 64) all &= (p[j] != 0);
  }
  all = __all(all);
  if (lane == 0) row_open[row] = all ? 1 : 0;
  if (!bits) return;
  const long W = (Lk + 31) >> 5;
  // bit j of a word = byte j != 0: (b | b >> 1 | ... ) folded per byte with a multiply-gather of the 4 low bits of each dword
  auto pack4 = [](unsigned x) {   // 4 bytes -> 4 bits (bit i = byte i != 0)
    x |= x >> 4; x |= x >> 2; x |= x >> 1; x &= 0x01010101u;
    return ((x * 0x01020408u) >> 24) & 0xFu;   // bytes 0..3 -> bits 0..3
  };
  for (long w = lane; w < W; w += 64) {
    unsigned word = 0;
    const long k0 = w * 32;
    if (!all) {
      if (vec && k0 + 32 <= Lk) {
        const uint4 a = *(const uint4*)(p + k0), b = *(const uint4*)(p + k0 + 16);
        word = pack4(a.x) | (pack4(a.y) << 4) | (pack4(a.z) << 8) | (pack4(a.w) << 12) | (pack4(b.x) << 16) | (pack4(b.y) << 20) |
               (pack4(b.z) << 24) | (pack4(b.w) << 28);
      } else {
        for (int j = 0; j < 32 && k0 + j < Lk; ++j) word |= (p[k0 + j] != 0 ? 1u : 0u) << j;
      }
    }
    bits[row * W + w] = word;
  }
}

int check_desc(const pq3d_attn_desc& d) {
  PQ_CHECK_ARG(d.B >= 0 && d.H >= 1 && d.Lq >= 0 && d.Lk >= 0, "pq3d_attn: bad sizes");
  PQ_CHECK_ARG(d.dh == 16 || d.dh == 32 || d.dh == 64, "pq3d_attn: head dim must be 16, 32 or 64");
  PQ_CHECK_ARG(d.ct == PQ3D_F32 || d.ct == PQ3D_BF16, "pq3d_attn: bad compute type");
  PQ_CHECK_ARG(d.dt == PQ3D_F32 || d.dt == PQ3D_BF16, "pq3d_attn: bad storage dtype");
  PQ_CHECK_ARG(d.q && d.k && d.v && d.o && d.lse, "pq3d_attn: null q/k/v/o/lse");
  PQ_CHECK_ARG(d.dt == d.ct, "pq3d_attn: storage dtype of q/k/v/o must equal the compute type");
  {
    const int epl = d.ct == PQ3D_BF16 ? 8 : 4;
    const int64_t st[] = {d.q_sb, d.q_sl, d.q_sh, d.k_sb, d.k_sl, d.k_sh, d.v_sb, d.v_sl, d.v_sh, d.o_sb, d.o_sl, d.o_sh};
    for (int64_t x : st) PQ_CHECK_ARG(x % epl == 0, "pq3d_attn: strides must be multiples of 16 bytes");
    const void* ps[] = {d.q, d.k, d.v, d.o};
    for (const void* p : ps) PQ_CHECK_ARG((((uintptr_t)p) & 15) == 0, "pq3d_attn: q/k/v/o must be 16-byte aligned");
  }
  PQ_CHECK_ARG(d.Lk > 0 || d.zero_attn, "pq3d_attn: Lk == 0 needs zero_attn");
  PQ_CHECK_ARG(d.Lk <= MAXKB * KB, "pq3d_attn: Lk too large for the key-block skip list");
  PQ_CHECK_ARG(d.ksplit <= 1 || (d.ws != nullptr && d.ksplit <= 64), "pq3d_attn: ksplit needs a workspace (ksplit <= 64)");
  PQ_CHECK_ARG(d.ksplit <= 1 || d.dbias == nullptr, "pq3d_attn: dbias requires ksplit == 1");
  PQ_CHECK_DROP(d.drop, (int64_t)(d.drop_bmod > 0 ? d.drop_bmod : d.B) * d.H * d.Lq, d.Lk, "pq3d_attn");
  PQ_CHECK_ARG(d.drop_bmod <= 0 || d.B % d.drop_bmod == 0, "pq3d_attn: B must be a multiple of drop_bmod");
  return 0;
}

template <typename CT, int DH, bool DROP, bool MASK3> void launch_fwd_k(const pq3d_attn_desc& d, hipStream_t s, int tiles, int ks) {
  const dim3 g8(((tiles + 7) / 8) * ks, d.H, d.B), g4(ks, d.H, d.B);
  if (tiles > 4) hipLaunchKernelGGL((attn_fwd_kernel<CT, DH, 8, DROP, MASK3>), g8, dim3(512), 0, s, d);
  else hipLaunchKernelGGL((attn_fwd_kernel<CT, DH, 4, DROP, MASK3>), g4, dim3(256), 0, s, d);
}
template <typename CT, int DH> int launch_fwd(const pq3d_attn_desc& d, hipStream_t s) {
  const int tiles = (d.Lq + 15) / 16, ks = d.ksplit > 1 ? d.ksplit : 1;
  const bool dr = d.drop.p > 0.f && d.drop.seed, m3 = d.mask != nullptr;
  if (g_ca && pq3d_attn_ca_try(d, s, false)) {   // few queries AND few keys, bf16: one workgroup per (scene, head)
  } else if (g_resfwd && pq3d_attn_fwd_resident_try(d, s)) {
  } else if (dr) { if (m3) launch_fwd_k<CT, DH, true, true>(d, s, tiles, ks); else launch_fwd_k<CT, DH, true, false>(d, s, tiles, ks); }
  else { if (m3) launch_fwd_k<CT, DH, false, true>(d, s, tiles, ks); else launch_fwd_k<CT, DH, false, false>(d, s, tiles, ks); }
  if (ks > 1) {
    const long n = (long)d.B * d.H * d.Lq * (DH / 4);
    hipLaunchKernelGGL((attn_fwd_combine_kernel<DH>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d);
  }
  PQ_LAUNCH_CHECK();
  return 0;
}
template <typename CT, int DH, bool DROP, bool MASK3> void launch_bwd_k(const pq3d_attn_desc& d, hipStream_t s, int tiles, int ks) {
  const dim3 g8(((tiles + 7) / 8) * ks, d.H, d.B), g4(ks, d.H, d.B);
  if (tiles > 4) hipLaunchKernelGGL((attn_bwd_dq_kernel<CT, DH, 8, DROP, MASK3>), g8, dim3(512), 0, s, d);
  else hipLaunchKernelGGL((attn_bwd_dq_kernel<CT, DH, 4, DROP, MASK3>), g4, dim3(256), 0, s, d);
  if (ks > 1) {
    const long n = (long)d.B * d.H * d.Lq * (DH / 4);
    hipLaunchKernelGGL((attn_dq_combine_kernel<DH>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d);
  }
  if (d.Lk > 256) {   // measured (c4, B12 Lq200 Lk4096, 3-D mask): 128 keys per workgroup -1.4 % of the step; Lk = 100: 64
    const dim3 gk((d.Lk + 127) / 128, d.H, d.B);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<CT, DH, DROP, MASK3, 8>), gk, dim3(512), 0, s, d);
  } else if (d.Lk > 0) {
    const dim3 gk((d.Lk + 63) / 64, d.H, d.B);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<CT, DH, DROP, MASK3, 4>), gk, dim3(256), 0, s, d);
  }
}
template <typename CT, int DH> int launch_bwd(const pq3d_attn_desc& d, hipStream_t s) {
  const int tiles = (d.Lq + 15) / 16;   // the dQ kernel also produces delta = rowsum(dO * O) for the dK/dV kernel
  const int ks = d.ksplit > 1 ? d.ksplit : 1;
  if constexpr (sizeof(CT) == 2 && (DH == 32 || DH == 64)) {
    if (g_ca && pq3d_attn_ca_try(d, s, true)) { PQ_LAUNCH_CHECK(); return 0; }   // few queries AND few keys (attn_ca.hip)
    // cross-attention shape (few queries, many keys): single-pass backward with all queries resident (attn_resident.hip)
    if (g_resident && pq3d_attn_bwd_resident_try(d, s)) {
      if (ks > 1) {
        const long n = (long)d.B * d.H * d.Lq * (DH / 4);
        hipLaunchKernelGGL((attn_dq_combine_kernel<DH>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d);
      }
      PQ_LAUNCH_CHECK();
      return 0;
    }
  }
  const bool dr = d.drop.p > 0.f && d.drop.seed, m3 = d.mask != nullptr;
  if (dr) { if (m3) launch_bwd_k<CT, DH, true, true>(d, s, tiles, ks); else launch_bwd_k<CT, DH, true, false>(d, s, tiles, ks); }
  else { if (m3) launch_bwd_k<CT, DH, false, true>(d, s, tiles, ks); else launch_bwd_k<CT, DH, false, false>(d, s, tiles, ks); }
  PQ_LAUNCH_CHECK();
  return 0;
}

#define DISPATCH(fn)                                                          \
  if (d.ct == PQ3D_BF16) {                                                    \
    if (d.dh == 16) return fn<bf16_t, 16>(d, s);                              \
    if (d.dh == 32) return fn<bf16_t, 32>(d, s);                              \
    return fn<bf16_t, 64>(d, s);                                              \
  } else {                                                                    \
    if (d.dh == 16) return fn<float, 16>(d, s);                               \
    if (d.dh == 32) return fn<float, 32>(d, s);                               \
    return fn<float, 64>(d, s);                                               \
  }

}  // namespace

extern "C" int pq3d_attn_fwd(const pq3d_attn_desc* dp, void* stream) {
  PQ_DEVICE_GUARD(stream, dp ? dp->q : nullptr);
  PQ_CHECK_ARG(dp != nullptr, "pq3d_attn_fwd: null descriptor");
  pq3d_attn_desc d = *dp;
  // PQ3D_BF16X3: fp32 storage, fp32-GRADE arithmetic -- the split-bf16 MFMA kernels where the call has their shape
  // (attn_sa.hip: the decoder's self-attention), the exact-fp32 kernels everywhere else
  const bool x3 = d.ct == PQ3D_BF16X3;
  if (x3) d.ct = PQ3D_F32;
  if (int e = check_desc(d)) return e;
  PQ_CHECK_ARG(d.proj.mode == PQ3D_ATTN_PROJ_NONE, "pq3d_attn_fwd: proj.mode not supported in the forward");
  if (d.B == 0 || d.Lq == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (d.k_lo || d.v_lo) {   // keys / values as bf16 hi / lo planes: the split-bf16 cross-attention forward (attn_x3.hip) or nothing
    PQ_CHECK_ARG(x3, "pq3d_attn_fwd: k_lo / v_lo need compute type PQ3D_BF16X3");
    if (int e = pq3d_attn_fwd_x3(d, s)) return e;
    if (d.ksplit > 1) {
      const long n = (long)d.B * d.H * d.Lq * (d.dh / 4);
      if (d.dh == 32) hipLaunchKernelGGL((attn_fwd_combine_kernel<32>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d);
      else hipLaunchKernelGGL((attn_fwd_combine_kernel<64>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d);
    }
    PQ_LAUNCH_CHECK();
    return 0;
  }
  if (x3 && g_sa && pq3d_attn_sa_try(d, s, false)) { PQ_LAUNCH_CHECK(); return 0; }
  if (g_small && pq3d_attn_small_try(d, s, false)) { PQ_LAUNCH_CHECK(); return 0; }
  DISPATCH(launch_fwd)
}

extern "C" int pq3d_attn_bwd(const pq3d_attn_desc* dp, void* stream) {
  PQ_DEVICE_GUARD(stream, dp ? dp->q : nullptr);
  PQ_CHECK_ARG(dp != nullptr, "pq3d_attn_bwd: null descriptor");
  pq3d_attn_desc d = *dp;
  const bool x3 = d.ct == PQ3D_BF16X3;
  if (x3) d.ct = PQ3D_F32;
  if (int e = check_desc(d)) return e;
  const bool fold = d.proj.mode == PQ3D_ATTN_PROJ_DOUT;
  PQ_CHECK_ARG(d.proj.mode == PQ3D_ATTN_PROJ_NONE || fold, "pq3d_attn_bwd: unknown proj.mode");
  PQ_CHECK_ARG((d.dout || fold) && d.dq && d.dk && d.dv && d.delta, "pq3d_attn_bwd: null dout/dq/dk/dv/delta");
  PQ_CHECK_ARG((((uintptr_t)d.dout) & 15) == 0, "pq3d_attn_bwd: dout must be 16-byte aligned");
  if (fold) PQ_CHECK_ARG(d.proj.x && d.proj.w[0] && d.proj.dm == d.H * d.dh && (d.proj.dm % 32) == 0 &&
                             ((((uintptr_t)d.proj.x) | ((uintptr_t)d.proj.w[0])) & 15) == 0,
                         "pq3d_attn_bwd: proj (DOUT) needs x, w[0] (16-byte aligned) and dm = H * dh, a multiple of 32");
  if (d.B == 0 || d.Lq == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (x3 && g_sa && pq3d_attn_sa_try(d, s, true)) { PQ_LAUNCH_CHECK(); return 0; }
  PQ_CHECK_ARG(!fold, "pq3d_attn_bwd: proj is served by the split-bf16 self-attention kernels only (PQ3D_BF16X3, d_h = 32, "
                      "<= 240 tokens, within LDS): not this call -- run the projection as its own pq3d_gemm");
  if (g_small && pq3d_attn_small_try(d, s, true)) { PQ_LAUNCH_CHECK(); return 0; }
  DISPATCH(launch_bwd)
}

extern "C" int pq3d_mask_pack(const uint8_t* mask, uint8_t* row_open, uint32_t* bits, int64_t rows, int64_t Lk, void* stream) {
  PQ_DEVICE_GUARD(stream, mask);
  PQ_CHECK_ARG(mask && row_open && rows >= 0 && Lk >= 0, "pq3d_mask_pack: bad args");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(mask_pack_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, mask, row_open, bits,
                     (long)rows, (long)Lk);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_mask_row_all(const uint8_t* mask, uint8_t* row_open, int64_t rows, int64_t Lk, void* stream) {
  PQ_DEVICE_GUARD(stream, mask);
  PQ_CHECK_ARG(mask && row_open && rows >= 0 && Lk >= 0, "pq3d_mask_row_all: bad args");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(mask_row_all_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, mask,
                     row_open, (long)rows, (long)Lk);
  PQ_LAUNCH_CHECK();
  return 0;
}
