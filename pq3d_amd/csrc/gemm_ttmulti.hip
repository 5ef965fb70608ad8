// Heterogeneous batch of short-reduction weight-gradient products in ONE launch.
//
//   dW_p[M_p, N_p] += A_p[K_p, M_p]^T (B_p[K_p, N_p] + B2_p[K_p, N_p]),   colsum_p[M_p] += column sums of A_p,   p < n
//
// These are the deferred dW = g^T (x [+ x2]) products of a backward pass over R = B * N_q query rows (nn.Linear weight
// gradients of the decoder layers: self-attention / cross-attention projections, FFN; reference: autograd of F.linear at
// query_encoder.py:194,268-270,385-388, transformers.py:190-193,239).  gemm_wktt_kernel runs them as one launch per (shape,
// dtype) bucket: at config 2 five dependent launches of 15-23 us each for ~0.1 GFLOP apiece -- latency chains (every launch:
// cold kernel arguments, one round of workgroups waiting on its first 256-row chunk) with nothing else to run beside them.
// Here every product of the flush is a PROBLEM in one table; a workgroup finds its (problem, tile, k-slice) from a prefix of
// workgroup counts and runs gemm_wktt's chunk loop with the operand dtypes read from the table (uniform branches).  Same
// arithmetic as gemm_wktt_kernel: 256 reduction rows staged per chunk, operands rounded to bf16 once, 64 x 64 tiles on 8 waves,
// fp32 accumulation, fp32 atomics into the (pre-zeroed / accumulating) gradient slots, bias gradient on the matrix pipe.
#include <atomic>

#include "gemm_common.h"

namespace {

constexpr int WT = 512;
constexpr int MAXP = PQ3D_TT_MAX_PROBLEMS;

struct TtProb {          // 64 bytes
  const void* A;
  const void* B;
  const float* B2;
  float* C;
  float* colsum;
  int32_t M, N, K, lda, ldb;
  int32_t flags;         // bit 0: A is bf16, bit 1: B is bf16
};
struct TtTable {
  int32_t n, pad;
  int32_t wg_end[MAXP];  // exclusive end of every problem's workgroup range
  int32_t sk[MAXP];      // k-slices of the problem
  TtProb p[MAXP];
};
static_assert(sizeof(TtTable) <= 4096, "the problem table travels as a kernel argument");

struct Raw8 {            // 8 source elements in flight (fp32: 8 registers, bf16: 4) + the optional fp32 addend
  float4 a, b;
};
PQ_DEV void raw_load(Raw8& r, const void* base, bool is_bf16, long off) {
  if (is_bf16) { const u32x4 v = *(const u32x4*)((const bf16_t*)base + off); r.a = __builtin_bit_cast(float4, v); }
  else { const float* p = (const float*)base + off; r.a = *(const float4*)p; r.b = *(const float4*)(p + 4); }
}
PQ_DEV u32x4 raw_pack(const Raw8& r, bool is_bf16, const Raw8* add) {
  if (is_bf16) return __builtin_bit_cast(u32x4, r.a);
  float4 a = r.a, b = r.b;
  if (add) { a.x += add->a.x; a.y += add->a.y; a.z += add->a.z; a.w += add->a.w; b.x += add->b.x; b.y += add->b.y; b.z += add->b.z; b.w += add->b.w; }
  return (u32x4){pack_bf2(a.x, a.y), pack_bf2(a.z, a.w), pack_bf2(b.x, b.y), pack_bf2(b.z, b.w)};
}

// Tile shapes (8 waves as 4 x 2, wave tile (TM / 4) x (TN / 2)):
//    64 x  64, 128 reduction rows per chunk: any M, N (multiples of 8)
//   256 x 128,  64 reduction rows per chunk: M % 256 == 0, N % 128 == 0 (every d = 256 / 768 / 512 / 2048-wide weight of the
//               decoder and the caption body).  At 64 x 64 each g tile is re-read by every n-tile and each x tile by every m-tile:
//               ~1 GB of L2 -> LDS traffic for the ~50 weight gradients of a config-2 backward, 94 us at the L2's rate whether
//               launched as five kernels or one.  The wide tile reads a [256 x 256] problem's g twice and x once (2.5-3x less).
template <int TM, int TN, int KC>
__global__ __launch_bounds__(WT) void gemm_tt_multi_kernel(const TtTable t) {
  constexpr int LDA = TM + 8, LDB = TN + 8, CPA = TM / 8, CPB = TN / 8;
  constexpr int NA = KC * CPA / WT, NB = KC * CPB / WT;          // 8-element chunks per thread per operand
  constexpr int MI = TM / 64, NJ = TN / 32;                      // 16 x 16 MFMA blocks per wave
  static_assert(NA >= 1 && NB >= 1 && KC % 32 == 0, "tile shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char tm_smem[];
  bf16_t* const At = (bf16_t*)tm_smem;        // [KC][LDA]
  bf16_t* const Bt = At + KC * LDA;           // [KC][LDB]
  // which problem: the table is tiny and uniform -> a scalar scan
  int pi = 0;
  while (pi + 1 < t.n && (int)blockIdx.x >= t.wg_end[pi]) ++pi;
  const TtProb q = t.p[pi];
  const int local = (int)blockIdx.x - (pi > 0 ? t.wg_end[pi - 1] : 0), sk = t.sk[pi];
  const int tm = (q.M + TM - 1) / TM, tn = (q.N + TN - 1) / TN;
  // n-tile fastest, then m-tile, then slice: the n-tiles of one m-tile (same g slab) are neighbours in dispatch order
  const int ny = local % tn, rest = local / tn, mx = rest % tm, split = rest / tm;
  const bool a16 = q.flags & 1, b16 = q.flags & 2, has2 = q.B2 != nullptr && !b16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const int wm = (wave >> 1) * (TM / 4), wn = (wave & 1) * (TN / 2);
  const int m0 = mx * TM, n0 = ny * TN;
  const int nck = (q.K + KC - 1) / KC, per = (nck + sk - 1) / sk;
  const int c0 = split * per, c1 = min(nck, c0 + per);
  if (c0 >= c1) return;
  f32x4 acc[MI][NJ], accb[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  float* cs_out = ny == 0 ? q.colsum : nullptr;
  const bool do_cs = cs_out != nullptr && wn == 0;
  Raw8 ra[NA], rb[NB], rb2[NB];
  auto issue = [&](int ck) {
    const int k0 = ck * KC;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int c = tid + i * WT, k = min(k0 + c / CPA, q.K - 1), x = (c % CPA) * 8;
      raw_load(ra[i], q.A, a16, (long)k * q.lda + min(m0 + x, q.M - 8));
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int c = tid + i * WT, k = min(k0 + c / CPB, q.K - 1), x = (c % CPB) * 8;
      const long off = (long)k * q.ldb + min(n0 + x, q.N - 8);
      raw_load(rb[i], q.B, b16, off);
      if (has2) raw_load(rb2[i], q.B2, false, off);
    }
  };
  auto put = [&](int ck) {
    const int k0 = ck * KC;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int c = tid + i * WT, kr = c / CPA;
      *(u32x4*)&At[kr * LDA + (c % CPA) * 8] = k0 + kr < q.K ? raw_pack(ra[i], a16, nullptr) : (u32x4){0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int c = tid + i * WT, kr = c / CPB;
      *(u32x4*)&Bt[kr * LDB + (c % CPB) * 8] = k0 + kr < q.K ? raw_pack(rb[i], b16, has2 ? &rb2[i] : nullptr) : (u32x4){0, 0, 0, 0};
    }
  };
  issue(c0);
  for (int ck = c0; ck < c1; ++ck) {
    if (ck > c0) __syncthreads();
    put(ck);
    __syncthreads();
    const int nks = (min(KC, q.K - ck * KC) + 31) >> 5;
    if (ck + 1 < c1) issue(ck + 1);
#pragma unroll
    for (int ks = 0; ks < KC / 32; ++ks) {
      if (ks < nks) {   // uniform
        u32x4 a[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) a[i] = km_frag(At, LDA, wm + i * 16, ks, li, lg);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const u32x4 bf = km_frag(Bt, LDB, wn + j * 16, ks, li, lg);
#pragma unroll
          for (int i = 0; i < MI; ++i) Mma<bf16_t>::mma(acc[i][j], a[i], bf);
        }
        if (do_cs) {
          const u32x4 ones = (u32x4){0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
#pragma unroll
          for (int i = 0; i < MI; ++i) Mma<bf16_t>::mma(accb[i], a[i], ones);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = n0 + wn + j * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm + i * 16 + lg * 4 + r;
        if (row < q.M && col < q.N) unsafeAtomicAdd(q.C + (long)row * q.N + col, acc[i][j][r]);
      }
    }
  if (do_cs && li == 0) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm + i * 16 + lg * 4 + r;
        if (row < q.M) unsafeAtomicAdd(&cs_out[row], accb[i][r]);
      }
  }
}

bool g_tt_wide = true;   // pq3d_gemm_tt_multi_wide(0): every problem on the 64 x 64 tile (A/B measurements)

template <int TM, int TN, int KC>
int tt_multi_launch(const pq3d_tt_problem* const* probs, int n, hipStream_t s) {
  TtTable t;
  t.n = n; t.pad = 0;
  long tiles = 0;
  for (int i = 0; i < n; ++i) tiles += (long)((probs[i]->M + TM - 1) / TM) * ((probs[i]->N + TN - 1) / TN);
  // k-slices: up to one round of workgroups when the flush is small (two 64-tile / one wide-tile workgroup per CU)
  const long round = TM == 64 ? 512 : 256;
  int end = 0;
  for (int i = 0; i < n; ++i) {
    const pq3d_tt_problem& p = *probs[i];
    const int tl = ((p.M + TM - 1) / TM) * ((p.N + TN - 1) / TN), nck = (p.K + KC - 1) / KC;
    int sk = (int)(round / (tiles > 0 ? tiles : 1));
    if (sk < 1) sk = 1;
    if (sk > nck) sk = nck;
    end += tl * sk;
    t.wg_end[i] = end; t.sk[i] = sk;
    TtProb& q = t.p[i];
    q.A = p.A; q.B = p.B; q.B2 = p.B2; q.C = p.C; q.colsum = p.colsum;
    q.M = p.M; q.N = p.N; q.K = p.K; q.lda = (int32_t)p.lda; q.ldb = (int32_t)p.ldb;
    q.flags = (p.dtA == PQ3D_BF16 ? 1 : 0) | (p.dtB == PQ3D_BF16 ? 2 : 0);
  }
  for (int i = n; i < MAXP; ++i) { t.wg_end[i] = end; t.sk[i] = 1; t.p[i] = TtProb{}; }
  constexpr size_t lds = (size_t)KC * (TM + 8 + TN + 8) * sizeof(bf16_t);
  auto kern = gemm_tt_multi_kernel<TM, TN, KC>;
  static std::atomic<unsigned> done{0};
  if (int e = pq3d_enable_big_lds(kern, (int)lds, done)) { pq3d_set_error(hipGetErrorString((hipError_t)e)); return e; }
  hipLaunchKernelGGL(kern, dim3((unsigned)end), dim3(WT), lds, s, t);
  return 0;
}

}  // namespace

// Constraints (checked): every M, N a multiple of 8 and >= 8; lda, ldb multiples of 8; A, B, B2 16-byte aligned; B2 only with an
// fp32 B; C rows contiguous (ld = N).  C and colsum must hold valid numbers (zeroed or running gradient slots).

extern "C" int pq3d_gemm_tt_multi(const pq3d_tt_problem* probs, int32_t n, void* stream) {
  PQ_DEVICE_GUARD(stream, probs && n > 0 ? probs[0].C : nullptr);
  PQ_CHECK_ARG(probs && n >= 1 && n <= MAXP, "pq3d_gemm_tt_multi: 1 .. PQ3D_TT_MAX_PROBLEMS problems");
  const pq3d_tt_problem* wide[MAXP];
  const pq3d_tt_problem* small[MAXP];
  int nw = 0, ns = 0;
  for (int i = 0; i < n; ++i) {
    const pq3d_tt_problem& p = probs[i];
    PQ_CHECK_ARG(p.A && p.B && p.C && p.M >= 8 && p.N >= 8 && p.K >= 1 && p.M % 8 == 0 && p.N % 8 == 0 && p.lda % 8 == 0 &&
                 p.ldb % 8 == 0 && p.lda >= p.M && p.ldb >= p.N, "pq3d_gemm_tt_multi: bad problem (sizes / leading dimensions)");
    PQ_CHECK_ARG((p.dtA == PQ3D_F32 || p.dtA == PQ3D_BF16) && (p.dtB == PQ3D_F32 || p.dtB == PQ3D_BF16) &&
                 (!p.B2 || p.dtB == PQ3D_F32), "pq3d_gemm_tt_multi: dtypes are f32 / bf16; B2 needs an fp32 B");
    PQ_CHECK_ARG(((((uintptr_t)p.A) | ((uintptr_t)p.B) | ((uintptr_t)p.B2)) & 15) == 0, "pq3d_gemm_tt_multi: operands must be 16-byte aligned");
    PQ_CHECK_ARG((long)p.K * p.lda < (1L << 31) && (long)p.K * p.ldb < (1L << 31), "pq3d_gemm_tt_multi: operand too large");
    if (g_tt_wide && p.M % 256 == 0 && p.N % 128 == 0) wide[nw++] = &p; else small[ns++] = &p;
  }
  hipStream_t s = (hipStream_t)stream;
  // the wide tile runs ONE workgroup per CU (200 VGPRs, 2 waves per SIMD): it wins while its workgroups fit about one round
  // (config 2 / 4 decoder flush: 208 tiles, -1.7 % / -0.6 % of the step); a flush of many short reductions (the caption
  // body at config 5: ~660 wide tiles over 512 rows) is faster on two 64-tile workgroups per CU (measured +0.5 % of the step
  // with the wide tile) -> those go back to the small class
  long wt = 0;
  for (int i = 0; i < nw; ++i) wt += (long)(wide[i]->M / 256) * (wide[i]->N / 128);
  if (wt > 400) { for (int i = 0; i < nw; ++i) small[ns++] = wide[i]; nw = 0; }
  if (nw) if (int e = tt_multi_launch<256, 128, 64>(wide, nw, s)) return e;
  if (ns) if (int e = tt_multi_launch<64, 64, 128>(small, ns, s)) return e;
  PQ_LAUNCH_CHECK();
  return 0;
}
extern "C" int pq3d_gemm_tt_multi_wide(int32_t on) { g_tt_wide = on != 0; return 0; }
