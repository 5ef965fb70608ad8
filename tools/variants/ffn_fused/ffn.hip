// The decoder's feed-forward sublayer (FFNLayer, query_encoder.py:371-388: linear2(dropout(act(linear1(x))))) as ONE
// launch at fp32 grade (split-bf16 products, the arithmetic of PQ3D_BF16X3 in gemm.hip).  As two grouped pq3d_gemm launches
// the pair costs 11 + 13 us at config 2 (800 rows, d = 256, F = 2048): the first product's 64x64 tiles re-read x and the
// fp32 weights from L2 (53 MB for a 1.7 GFLOP product), the hidden activations make a round trip through HBM, and the second
// launch waits for the first.  Here a workgroup owns 64 rows and one 256-wide slice of the hidden layer:
//   * the x tile is split once into hi / lo bf16 planes in LDS and stays there;
//   * 4 steps of phase 1: 64 rows of W1 (64 hidden units x 256 inputs) go global -> registers -> hi / lo planes -> LDS, all
//     loads of the step in flight at once and the next step's requested as soon as this one is parked (gemm_wk.hip's
//     schedule).  Each of the 8 waves (4 row blocks x 2 halves of the step's hidden units) forms the TRANSPOSED tile
//     H^T = W1 x^T of its 16 rows x 32 hidden units: the MFMA C layout (lane = row, 4 consecutive hidden units per tile)
//     is, for two tiles side by side, exactly a B operand over 32 hidden units in a permuted order -- bias, activation,
//     dropout and the store of h (float4 per lane) happen on those registers and the activated values never leave them;
//   * 4 steps of phase 2: the matching 64 columns of W2 (256 outputs x 64 hidden units) are staged the same way and read
//     back in the SAME permuted order of hidden units (two 8-byte reads per fragment), Y^T += W2 H^T on 16 output tiles;
//   * the two waves of a row block add their halves through LDS and the workgroup writes its [64 x 256] PARTIAL sum of
//     slice s (+ linear2's bias in slice 0).  The F / 256 partial sums are added in a fixed order by the LayerNorm that
//     follows (pq3d_add_ln_fwd, sum_branches) -- deterministic, no atomics, row-independent.
// Weight traffic from L2: (R / 64) x (W1 + W2) = 52 MB at config 2 (as much as the first product alone moved before); 8
// dependent staging steps of 64 KB per workgroup.  fp32 x / h / partial sums; d = 256 only (the decoder's width).
//
// MEASURED (round 3, config 2, same box): 34 us per launch against 11 + 13 us for the two pq3d_gemm launches -- the fused
// executor keeps the pair (PQ3D_FFN_FUSE=1 selects this kernel; parity tests run it either way).  Why: at fp32 grade every
// MFMA triple needs its operands twice (hi and lo planes), and with 8 waves on a 64-row x 64-hidden-unit step a wave owns
// one row block x two hidden tiles -- 6 ds_read_b128 per 6 MFMAs (phase 1: 384 LDS clocks per k step for the CU against
// 192 MFMA clocks per SIMD) and 4 ds_read_b64 per MFMA triple in phase 2 (2-way bank conflicts at this row stride): the
// LDS pipe, not the matrix pipe, sets the step time (~1.3-1.7 us x 8 steps), and one workgroup per CU (141 KB of LDS)
// on 104 of the 256 CUs leaves nothing to overlap the 8 dependent weight-staging round trips with.  Larger per-wave tiles
// (2 x 2 register blocking) would halve the LDS traffic but need either 128-row tiles (x planes 135 KB) or 128-unit steps
// (W1 planes 135 KB) -- both beyond what fits next to the other operand; the two-launch form spreads the same work over
// 416 + 208 workgroups and hides its staging latency with occupancy instead.
#include <atomic>

#include "gemm_common.h"

namespace {

constexpr int FT = 512;               // threads: 8 waves = 4 row blocks x 2 hidden-unit halves of a step
constexpr int FTM = 64;               // rows per workgroup
constexpr int FS = 256;               // hidden units per workgroup (slice)
constexpr int FD = 256;               // model width (K of phase 1, N of phase 2)
constexpr int LDX = FD + 8;           // row stride of the x / W1 planes (bf16 elements)
constexpr int LDW2 = 64 + 8;          // row stride of the W2 step planes [256 outputs][64 hidden units]
constexpr int LDRED = FD + 4;         // row stride of the fp32 reduction buffer [64 rows][256 outputs]
constexpr size_t XPLANE = (size_t)FTM * LDX;                                        // elements per x plane
constexpr size_t WSTAGE = (size_t)FD * LDW2 > (size_t)64 * LDX ? (size_t)FD * LDW2 : (size_t)64 * LDX;   // per plane
constexpr size_t FFN_LDS = (2 * XPLANE + 2 * WSTAGE) * sizeof(bf16_t);
static_assert((size_t)FTM * LDRED * sizeof(float) <= 2 * WSTAGE * sizeof(bf16_t), "the reduction buffer overlays the weight stage");

struct HL8 { u32x4 hi, lo; };
PQ_DEV HL8 split8(const float* v) {
  HL8 r;
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
PQ_DEV void mma3(f32x4& acc, const HL8& a, const HL8& b) {   // lo*hi + hi*lo + hi*hi: the order of gemm.hip's split products
  Mma<bf16_t>::mma(acc, a.lo, b.hi);
  Mma<bf16_t>::mma(acc, a.hi, b.lo);
  Mma<bf16_t>::mma(acc, a.hi, b.hi);
}

template <bool GELU, bool DROP>
__global__ __launch_bounds__(FT) void ffn_fwd_kernel(const pq3d_ffn_desc d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ffn_sm[];
  bf16_t* const Xh = (bf16_t*)ffn_sm;
  bf16_t* const Xl = Xh + XPLANE;
  bf16_t* const Wh = Xl + XPLANE;
  bf16_t* const Wl = Wh + WSTAGE;
  float* const red = (float*)Wh;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const int rb = wave & 3, fh = wave >> 2;
  const int m0 = blockIdx.x * FTM, s = blockIdx.y;
  const int F = d.F;
  const float* w1 = d.w1 + (long)s * FS * FD;   // rows s*256 .. of [F][256]
  const float* w2 = d.w2 + (long)s * FS;        // columns s*256 .. of [256][F]

  // ---- staging: 2048 chunks of 8 floats per step, 4 per thread (8 float4 in flight)
  float4 ra[4], rbv[4];
  auto issue_x = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * FT, row = c >> 5, k = (c & 31) * 8;
      const float* p = d.x + (long)min(m0 + row, d.R - 1) * FD + k;
      ra[i] = *(const float4*)p;
      rbv[i] = *(const float4*)(p + 4);
    }
  };
  auto issue_w1 = [&](int j) {   // hidden units j*64 .. j*64 + 63 of the slice: [64][256]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * FT, row = c >> 5, k = (c & 31) * 8;
      const float* p = w1 + (long)(j * 64 + row) * FD + k;
      ra[i] = *(const float4*)p;
      rbv[i] = *(const float4*)(p + 4);
    }
  };
  auto issue_w2 = [&](int j) {   // [256 outputs][64 hidden units j*64 ..]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * FT, row = c >> 3, k = (c & 7) * 8;
      const float* p = w2 + (long)row * F + j * 64 + k;
      ra[i] = *(const float4*)p;
      rbv[i] = *(const float4*)(p + 4);
    }
  };
  auto park = [&](bf16_t* ph, bf16_t* pl, int shift, int ld) {   // shift: log2(chunks per row)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * FT, row = c >> shift, k = (c & ((1 << shift) - 1)) * 8;
      const float v[8] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w, rbv[i].x, rbv[i].y, rbv[i].z, rbv[i].w};
      const HL8 t = split8(v);
      *(u32x4*)&ph[row * ld + k] = t.hi;
      *(u32x4*)&pl[row * ld + k] = t.lo;
    }
  };

  issue_x();
  park(Xh, Xl, 5, LDX);
  issue_w1(0);

  const int row_l = rb * 16 + li;            // this lane's row of the tile (B operand column / C layout column)
  const long grow = m0 + row_l;
  const bool row_ok = grow < d.R;
  DropState dst;
  if constexpr (DROP) dst = drop_init(d.drop, 0, F);

  HL8 hp[4];                                  // activated hidden units of the 4 steps, B-operand layout (permuted order)
  // ---------------- phase 1: H^T = act(W1 x^T + b1), 4 steps of 64 hidden units (32 per wave: two tiles)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (j > 0) __syncthreads();               // the previous step's fragment reads are done
    park(Wh, Wl, 5, LDX);
    __syncthreads();
    if (j < 3) issue_w1(j + 1); else issue_w2(0);
    const int f0 = s * FS + j * 64 + fh * 32;           // first hidden unit of this wave's pair of tiles
    const float4 b0 = *(const float4*)(d.b1 + f0 + 4 * lg), b1v = *(const float4*)(d.b1 + f0 + 16 + 4 * lg);
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < FD / 32; ++ks) {
      const int ox = row_l * LDX + ks * 32 + lg * 8;
      HL8 xb, wa0, wa1;
      xb.hi = *(const u32x4*)&Xh[ox];
      xb.lo = *(const u32x4*)&Xl[ox];
      const int o0 = (fh * 32 + li) * LDX + ks * 32 + lg * 8, o1 = o0 + 16 * LDX;
      wa0.hi = *(const u32x4*)&Wh[o0]; wa0.lo = *(const u32x4*)&Wl[o0];
      wa1.hi = *(const u32x4*)&Wh[o1]; wa1.lo = *(const u32x4*)&Wl[o1];
      mma3(a0, wa0, xb);
      mma3(a1, wa1, xb);
    }
    // a0[r] = pre-activation of (row, hidden unit f0 + 4 lg + r), a1[r]: f0 + 16 + 4 lg + r
    float v[8] = {a0[0] + b0.x, a0[1] + b0.y, a0[2] + b0.z, a0[3] + b0.w, a1[0] + b1v.x, a1[1] + b1v.y, a1[2] + b1v.z, a1[3] + b1v.w};
    if (d.pre && row_ok) {
      float* pp = d.pre + grow * F + f0 + 4 * lg;
      *(float4*)pp = make_float4(v[0], v[1], v[2], v[3]);
      *(float4*)(pp + 16) = make_float4(v[4], v[5], v[6], v[7]);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = GELU ? gelu_f(v[q]) : fmaxf(v[q], 0.f);
    if constexpr (DROP) {   // site: h viewed as [R, F] (pq3d_gemm's epilogue: word = pair of columns)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; q += 2) {
          const uint32_t w = drop_word(dst, (uint32_t)grow, (uint32_t)(f0 + t * 16 + 4 * lg + q) >> 1);
          v[t * 4 + q] = drop_keep_lo(dst, w) ? v[t * 4 + q] * dst.scale : 0.f;
          v[t * 4 + q + 1] = drop_keep_hi(dst, w) ? v[t * 4 + q + 1] * dst.scale : 0.f;
        }
    }
    if (row_ok) {
      float* ph = d.h + grow * F + f0 + 4 * lg;
      *(float4*)ph = make_float4(v[0], v[1], v[2], v[3]);
      *(float4*)(ph + 16) = make_float4(v[4], v[5], v[6], v[7]);
    }
    hp[j] = split8(v);
  }

  // ---------------- phase 2: Y^T += W2 H^T, 4 steps of 64 hidden units; 16 output tiles per wave
  f32x4 y[16];
#pragma unroll
  for (int ot = 0; ot < 16; ++ot) y[ot] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    __syncthreads();
    park(Wh, Wl, 3, LDW2);
    __syncthreads();
    if (j < 3) issue_w2(j + 1);
#pragma unroll
    for (int ot = 0; ot < 16; ++ot) {
      // A fragment: output ot*16 + li, hidden units in the order the two C tiles sit in the lanes of hp:
      // slots 0..3 = fh*32 + 4 lg + 0..3, slots 4..7 = fh*32 + 16 + 4 lg + 0..3
      const int o = (ot * 16 + li) * LDW2 + fh * 32 + 4 * lg;
      const u32x2 h0 = *(const u32x2*)&Wh[o], h1 = *(const u32x2*)&Wh[o + 16];
      const u32x2 l0 = *(const u32x2*)&Wl[o], l1 = *(const u32x2*)&Wl[o + 16];
      HL8 wa;
      wa.hi = (u32x4){h0.x, h0.y, h1.x, h1.y};
      wa.lo = (u32x4){l0.x, l0.y, l1.x, l1.y};
      mma3(y[ot], wa, hp[j]);
    }
  }
  // ---------------- the two halves of a row block through LDS, then the partial sum of this slice
  __syncthreads();   // weight stage dead
  if (fh == 1) {
#pragma unroll
    for (int ot = 0; ot < 16; ++ot)
      *(float4*)&red[row_l * LDRED + ot * 16 + 4 * lg] = make_float4(y[ot][0], y[ot][1], y[ot][2], y[ot][3]);
  }
  __syncthreads();
  if (fh == 0 && row_ok) {
    float* out = d.zp + ((long)s * d.R + grow) * FD;
#pragma unroll
    for (int ot = 0; ot < 16; ++ot) {
      const float4 t = *(const float4*)&red[row_l * LDRED + ot * 16 + 4 * lg];
      float4 r = make_float4(y[ot][0] + t.x, y[ot][1] + t.y, y[ot][2] + t.z, y[ot][3] + t.w);
      if (s == 0 && d.b2) {
        const float4 bb = *(const float4*)(d.b2 + ot * 16 + 4 * lg);
        r.x += bb.x; r.y += bb.y; r.z += bb.z; r.w += bb.w;
      }
      *(float4*)(out + ot * 16 + 4 * lg) = r;
    }
  }
}

bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

template <bool GELU, bool DROP> int ffn_launch(const pq3d_ffn_desc& d, hipStream_t s) {
  auto kern = ffn_fwd_kernel<GELU, DROP>;
  static std::atomic<unsigned> done{0};
  if (int e = pq3d_enable_big_lds(kern, (int)FFN_LDS, done)) return e;
  hipLaunchKernelGGL(kern, dim3((d.R + FTM - 1) / FTM, d.F / FS), dim3(FT), FFN_LDS, s, d);
  return 0;
}

}  // namespace

extern "C" int pq3d_ffn_fwd(const pq3d_ffn_desc* dp, void* stream) {
  PQ_DEVICE_GUARD(stream, dp ? dp->x : nullptr);
  PQ_CHECK_ARG(dp != nullptr, "pq3d_ffn_fwd: null descriptor");
  const pq3d_ffn_desc d = *dp;
  PQ_CHECK_ARG(d.R >= 0 && d.d == FD && d.F >= FS && d.F % FS == 0 && d.F / FS <= PQ3D_MAX_GROUPS,
               "pq3d_ffn_fwd: d must be 256 and F a multiple of 256 (at most 32 slices)");
  PQ_CHECK_ARG(d.act == PQ3D_ACT_RELU || d.act == PQ3D_ACT_GELU, "pq3d_ffn_fwd: act must be PQ3D_ACT_RELU or PQ3D_ACT_GELU");
  PQ_CHECK_ARG(d.x && d.w1 && d.b1 && d.w2 && d.h && d.zp, "pq3d_ffn_fwd: null x / w1 / b1 / w2 / h / zp");
  PQ_CHECK_ARG(al16(d.x) && al16(d.w1) && al16(d.b1) && al16(d.w2) && al16(d.b2) && al16(d.h) && al16(d.pre) && al16(d.zp),
               "pq3d_ffn_fwd: every pointer must be 16-byte aligned");
  const bool dr = d.drop.p > 0.f && d.drop.seed != nullptr;
  PQ_CHECK_ARG((long)d.R * (d.F / 2) < (1L << 32) || !dr, "pq3d_ffn_fwd: dropout site too large");
  if (d.R == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const bool gelu = d.act == PQ3D_ACT_GELU;
  int e;
  if (gelu) e = dr ? ffn_launch<true, true>(d, s) : ffn_launch<true, false>(d, s);
  else e = dr ? ffn_launch<false, true>(d, s) : ffn_launch<false, false>(d, s);
  if (e) return e;
  PQ_LAUNCH_CHECK();
  return 0;
}
