/* pq3d_hip.h -- C ABI of libpq3d_hip.so: MI355X (gfx950) kernels for PQ3D's promptable query decoder.
 *
 * The reference (PQ3D, /root/reference) is pure Python on stock PyTorch ops; it defines no FFI for this
 * path.  Each entry point below therefore cites the reference *function* whose arithmetic it replaces
 * (file:line relative to the reference root).  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; every pointer is a DEVICE pointer unless noted;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); nothing synchronises the device,
 *     nothing allocates: workspaces are caller-provided; the library holds no thread-local device state and
 *     is re-entrant (PyTorch calls backward from its autograd thread);
 *   - return 0 on success, PQ3D_ERR_ARG (<0) for argument errors, >0 = hipError_t; pq3d_last_error() gives text;
 *   - dtype enum: PQ3D_F32 / PQ3D_BF16 (raw bfloat16 bits); "compute type" (ct) selects the MFMA path:
 *     PQ3D_BF16 -> v_mfma_f32_16x16x32_bf16 (fp32 accumulate), PQ3D_F32 -> v_mfma_f32_16x16x4_f32 (exact f32);
 *   - masks are uint8 (torch.bool storage), PyTorch convention: nonzero = ignore / padded.
 */
#ifndef PQ3D_HIP_H
#define PQ3D_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PQ3D_F32 0
#define PQ3D_BF16 1
/* compute type only (pq3d_gemm): split-bf16.  Each fp32 operand element x is staged as hi = bf16(x), lo = bf16(x - hi)
 * and the product is formed as hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16 (fp32 accumulate): ~2^-17 relative
 * per term instead of 2^-9, i.e. fp32-grade results from the bf16 matrix cores at 3x the MFMA work.  Used by the
 * 'bf16' compute mode for the query-side projections / FFN / heads (M = B*N_q rows: launch-latency-bound, the extra
 * MFMAs are free), which is what holds the per-sublayer parity of that mode at <= 1e-3 (tests/test_gpu_sublayer_parity.py).
 * Operands must be fp32; layouts the split kernel does not cover run on the exact-f32 MFMA path (same accuracy). */
#define PQ3D_BF16X3 2
#define PQ3D_ERR_ARG (-1)

#define PQ3D_ACT_NONE 0
#define PQ3D_ACT_RELU 1
#define PQ3D_ACT_GELU 2

#define PQ3D_MAX_GROUPS 32
#define PQ3D_ACT_ADD 3 /* act_grad mode: C = acc + aux (fused residual / gradient accumulation) */
#define PQ3D_ACT_PLANES 4 /* act_grad mode: v = act(alpha acc + bias) leaves as two bf16 planes, C = bf16(v), C2 = bf16(v - C)
                             (operand form of the split-bf16 attention, compute mode 'bf16x3'); plain bf16 NT products on the
                             128-row-tile kernel only (gemm128.hip): any other call is refused.  With ct = PQ3D_BF16X3, bf16 A / B
                             and A2 / B2 = their RESIDUAL planes the product itself is split-bf16, (A + A2)(B + B2)^T without the
                             A2 B2 term, 3 MFMAs per term pair on one staging of the four planes (gemm_x3p.hip: N % 128 == 0,
                             K % 32 == 0) -- the hoisted key/value projection of compute mode 'bf16x3' */

const char* pq3d_last_error(void);
int pq3d_version(void);

/* ------------------------------------------------------------------------------------------------
 * Dropout (train mode of the reference: nn.Dropout in every sublayer, query_encoder.py:198,224,273,304,366,385-386,
 * transformers.py:239-241, and nn.MultiheadAttention(dropout=p), query_encoder.py:195,268; utils.py:23;
 * object_encoder.py:72-73).  Counter-based, so the backward pass regenerates the forward's mask instead of
 * storing it.  A dropout *site* is a 2-D array [rows, cols]; element (r, c) is KEPT iff
 *     half16(word(r, c >> 1), c & 1) >= round(p * 65536)
 *     word(r, j) = mix(r * ceil(cols / 2) + j; k0, k1)        (rows * ceil(cols/2) must be < 2^32)
 *     mix(i; k0, k1): h = i ^ k0; h ^= h>>16; h *= 0x7feb352d; h ^= h>>15; h += k1; h *= 0x846ca68b; h ^= h>>16
 *     k0 = fin(lo32(seed) ^ (site * 0x9E3779B1)),  k1 = fin(k0 + hi32(seed) + 0x85ebca6b)
 *     fin(h):  h ^= h>>16; h *= 0x7feb352d; h ^= h>>15; h *= 0x846ca68b; h ^= h>>16
 * and kept values are multiplied by 1/(1-p) (torch semantics).  `seed` is a DEVICE pointer to one 64-bit word: a
 * captured HIP graph draws fresh masks on every replay once the host (or a captured kernel) bumps the word.
 * p == 0 or seed == NULL turns the site off.  Kernels that fuse dropout document which array is the site.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  float p;
  uint32_t site;
  const uint64_t* seed;
} pq3d_dropout;

/* keep[r*cols + c] = 1 if element (r, c) of the site is kept (tests / debugging; same hash as the fused sites). */
int pq3d_dropout_mask(uint8_t* keep, int64_t rows, int64_t cols, const pq3d_dropout* dr, void* stream);
/* y = dropout(x) elementwise over a [rows, cols] site (also its own backward: dx = dropout(dy) with the same
 * descriptor).  Used where the reference applies nn.Dropout to a tensor no kernel of ours produces last
 * (get_mlp_head, utils.py:23; ObjectEncoder, object_encoder.py:72-73). */
int pq3d_dropout_apply(const void* x, int32_t dt_x, void* y, int32_t dt_y, int64_t rows, int64_t cols,
                       const pq3d_dropout* dr, void* stream);
/* the same times a constant: y = alpha * dropout(x) (the caption decoder's "dropout(final norm) * d_model^-0.5" in front of
 * the tied LM head, one launch instead of two in either direction) */
int pq3d_dropout_apply_scaled(const void* x, int32_t dt_x, void* y, int32_t dt_y, int64_t rows, int64_t cols,
                              const pq3d_dropout* dr, float alpha, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Grouped / batched GEMM with fused prologue + epilogue (nn.Linear forward and both backward GEMMs,
 * MaskPredictionLayer einsum).  Replaces: F.linear calls inside nn.MultiheadAttention
 * (query_encoder.py:268-270,194), MultiHeadAttentionSpatial w_qs/w_ks/w_vs/fc (transformers.py:190-193,239),
 * FFNLayer.linear1/2 (query_encoder.py:384-388), get_mlp_head (utils.py:18-25), ObjectEncoder.input_feat_proj
 * (object_encoder.py:71), MaskPredictionLayer (mask_head.py:53-57).
 *
 *   for g < groups, z < batch:
 *     C_g[z][m][n] = epi( sum_k (A_g[z](m,k) + A2_g[z](m,k)) * (B_g[z](n,k) + B2_g[z](n,k)) )
 *   kconcat = c > 0: every c consecutive groups are concatenated along K into one output (groups/c outputs, taken
 *     from C[0], C[c], C[2c], ...):  C_o[z][m][n] = epi( sum_{g in [o*c,(o+1)*c)} sum_k A_g(m,k) B_g(n,k) ) --
 *     the multi-memory sum of mask_head.py:30-37 and the "sum over consumers" of input gradients.
 *   accumulate != 0 (split-K only): C is NOT zeroed by the call -- the atomics add onto its current contents
 *     (parameter-gradient accumulation across layers / shared-weight blocks into a pre-zeroed arena).
 *   A(m,k) is at A[m*lda + k] (transA=0) or A[k*lda + m] (transA=1); B(n,k) at B[n*ldb + k] (transB=0)
 *   or B[k*ldb + n] (transB=1).  epi: + bias_g[n]; optional C2 <- pre-activation; act; if act_grad: multiply by
 *   act'(aux) (aux has C's layout: ReLU uses aux>0, GELU uses the saved pre-activation); row_mask_g[z*M+m]==0
 *   zeroes the output row; row_scale/row_fill implement mask_head.py:37-38 (C = row_fill_flag ? fill : acc*scale)
 *   and mask_out (uint8 [batch][N][M]) receives sigmoid(C) < 0.5 (mask_head.py:43).
 *   splitk > 1: K is split over blockIdx; C must be fp32 and is zeroed by this call, then accumulated with
 *   atomics (no epilogue options besides alpha).
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t M, N, K;
  int32_t groups, batch;
  int32_t ct;            /* compute type */
  int32_t dtA, dtA2, dtB, dtC, dtC2, dtAux, dtBias;
  int32_t transA, transB;
  int32_t act, act_grad;
  int32_t splitk;
  int32_t kconcat;
  int32_t accumulate;
  int32_t dtB2;
  float alpha;           /* applied to the accumulator before bias */
  float row_fill;        /* value written to rows whose row_fill_flag != 0 */
  int64_t lda, ldb, ldc; /* A2 shares lda; C2 / aux share ldc */
  int64_t strideA, strideB, strideC; /* per-batch element strides (A2: strideA; C2/aux: strideC) */
  const void* A[PQ3D_MAX_GROUPS];
  const void* A2[PQ3D_MAX_GROUPS];
  const void* B[PQ3D_MAX_GROUPS];
  const void* B2[PQ3D_MAX_GROUPS];
  const void* bias[PQ3D_MAX_GROUPS];
  void* C[PQ3D_MAX_GROUPS];
  void* C2[PQ3D_MAX_GROUPS];
  const void* aux[PQ3D_MAX_GROUPS];
  const uint8_t* row_mask[PQ3D_MAX_GROUPS];       /* [batch*M], 0 = zero the row */
  float* colsum[PQ3D_MAX_GROUPS];                 /* transA GEMMs only, or NULL: colsum_g[m] += sum_k A_g(m,k) -- the
                                                     bias gradient sum_r dY[r][m] of the same linear (split-K launches
                                                     only; zeroed by the call unless `accumulate`, then atomics) */
  const float* row_scale;                         /* [batch*M] or NULL (group 0 only) */
  const uint8_t* row_fill_flag;                   /* [batch*M] or NULL */
  uint8_t* mask_out;                              /* [batch][N][M] or NULL */
  /* dropout applied to the activated output (site = C viewed as [batch*M, N], site id drop.site + group):
   * FFNLayer's self.dropout(self.activation(self.linear1(x))), query_encoder.py:385.  Not with split-K. */
  pq3d_dropout drop;
} pq3d_gemm_desc;

int pq3d_gemm(const pq3d_gemm_desc* d, void* stream);
/* pq3d_gemm serves the small-M launches of the query side (M = B*N_q rows; batch 1, non-transposed A, fp32 B, bf16 /
 * split-bf16 compute) with whole-K tiles (csrc/gemm_wk.hip): 32- or 64-row x 64-column tiles on 8 waves, all operand
 * loads of a 256- (or 128-) wide k chunk in flight at once, chosen so that every workgroup of the launch is resident in
 * one round -- same MFMA order and epilogue as the 64x64-tile kernel, identical bits.  This process-wide switch sets the
 * option word (bit 0: on -- the default; bits 4-5: force 32- / 64-row tiles
 * (1 / 2); bits 6-7: force 128- / 256-wide chunks (1 / 2); bit 8: also take launches that need several rounds) and the
 * largest M they take (max_m <= 0: keep; default 2048).  For A/B measurements and tests. */
int pq3d_gemm_set_wk(int options, int max_m);


/* A heterogeneous batch of short-reduction weight-gradient products in ONE launch (pq3d_amd/csrc/gemm_ttmulti.hip): for every
 * problem p < n (n <= PQ3D_TT_MAX_PROBLEMS)
 *     C[M, N] (fp32, ld = N) += A[K, M]^T (B[K, N] + B2[K, N]),      colsum[M] += column sums of A   (NULL: skipped)
 * -- the dW = g^T (x [+ x2]) (and bias gradient) of every nn.Linear of a backward pass over K = B * N_q query rows, whatever
 * their shapes and operand dtypes (f32 / bf16 per operand; B2 fp32, only with an fp32 B), instead of one pq3d_gemm launch per
 * (shape, dtype) bucket.  Operands are rounded to bf16 once, fp32 accumulation, fp32 atomics into C / colsum, which must hold
 * valid numbers (zeroed or running gradient slots).  M, N, lda, ldb multiples of 8; A, B, B2 16-byte aligned. */
#define PQ3D_TT_MAX_PROBLEMS 56
typedef struct {
  int32_t M, N, K, dtA, dtB;
  int64_t lda, ldb;
  const void* A;
  const void* B;
  const float* B2;
  float* C;
  float* colsum;
} pq3d_tt_problem;
int pq3d_gemm_tt_multi(const pq3d_tt_problem* probs, int32_t n, void* stream);
/* process-wide switch for A/B measurements: 0 = every problem on the 64 x 64 tile (default 1: problems with M % 256 == 0 and
 * N % 128 == 0 take 256 x 128 tiles, which read their operands 2.5-3x less often from L2) */
int pq3d_gemm_tt_multi_wide(int32_t on);

/* ------------------------------------------------------------------------------------------------
 * Fused masked multi-head attention (flash-style online softmax, never materialises [B*H,Lq,Lk]).
 * Replaces the need_weights branch of F.multi_head_attention_forward as driven by CrossAttentionLayer
 * (query_encoder.py:288-307, add_zero_attn=True :268-270), SelfAttentionLayer (:213-227) and, with `bias`,
 * MultiHeadAttentionSpatial's softmax(log(clamp(loc,1e-6)) + qk/sqrt(dh)).v (transformers.py:192-237).
 *
 *   P[b,h,i,:] = softmax_j( scale * q[b,i,h,:].k[b,j,h,:] + bias[b,h,i,j] + mask terms  [, 0 for the zero key] )
 *   o[b,i,h,:] = sum_j P[b,h,i,j] v[b,j,h,:]          lse[b,h,i] = log sum exp
 *   mask terms: kpm[b,j] != 0 -> -inf; mask[b,i,j] != 0 && !(row_open && row_open[b,i]) -> -inf  (row_open
 *   reproduces `attn_mask[attn_mask.all(-1)] = False`, query_encoder.py:83, without rewriting the mask);
 *   zero_attn: an extra never-masked key with logit exactly 0 and value 0 (torch add_zero_attn).
 * Tensors are addressed as base + b*sb + l*sl + h*sh + c (element strides), so both [B,L,H*dh] views of
 * projection outputs and head-major layouts work.  dh in {16, 32, 64}.
 * Backward (pq3d_attn_bwd) recomputes P from lse: needs o, do; writes dq/dk/dv with the strides of q/k/v,
 * `delta` is a [B,H,Lq] fp32 workspace, dbias ([B,H,Lq,Lk] fp32, may be NULL) receives dL/dbias.
 * ------------------------------------------------------------------------------------------------ */
/* Projection folded into an attention launch (the split-bf16 self-attention kernels only: compute type PQ3D_BF16X3,
 * d_h = 32, at most 240 queries / keys; any other call with mode != 0 is refused with an error, nothing is launched).
 *   PQ3D_ATTN_PROJ_DOUT (backward): the out-projection's input gradient is formed in the kernel,
 *     dO[b, i, h, :] = x[b, i, :] . w[0][:, h*dh : (h+1)*dh]      (x = dL/d(out-proj output) [B, Lq, dm] fp32,
 *     w[0] = out-proj weight [dm, dm] fp32; bf16 operands, fp32 accumulate -- the arithmetic of the backward products),
 *     `dout` is ignored: one dependent launch and the [B, Lq, dm] round trip of dO less per layer
 *     (query_encoder.py:213-227 backward; MultiHeadAttentionSpatial.fc, transformers.py:239). */
#define PQ3D_ATTN_PROJ_NONE 0
#define PQ3D_ATTN_PROJ_DOUT 2
typedef struct {
  int32_t mode, dm;
  const float* x;
  const float* x2;     /* reserved */
  const float* w[3];
  const float* b[3];   /* reserved */
} pq3d_attn_proj;

typedef struct {
  int32_t B, H, Lq, Lk, dh;
  int32_t ct;   /* compute type: PQ3D_F32 / PQ3D_BF16 (= storage dtype), or PQ3D_BF16X3 with fp32 storage */
  int32_t dt;   /* storage dtype of q,k,v,o,do,dq,dk,dv */
  int32_t zero_attn;
  int32_t mask_bmod; /* > 0: `mask` / `row_open` are shared by groups of scenes (memories stacked along B) */
  float scale;
  int64_t q_sb, q_sl, q_sh;
  int64_t k_sb, k_sl, k_sh;
  int64_t v_sb, v_sl, v_sh;
  int64_t o_sb, o_sl, o_sh;   /* o and do */
  const void* q; const void* k; const void* v;
  void* o;                    /* fwd: output; bwd: forward output (input) */
  float* lse;                 /* [B,H,Lq] fwd: output; bwd: input */
  const uint8_t* kpm;         /* [B,Lk] or NULL */
  const uint8_t* mask;        /* [B,Lq,Lk] or NULL; with mask_bmod > 0: [mask_bmod,Lq,Lk] indexed by b % mask_bmod */
  const uint8_t* row_open;    /* [B,Lq] or NULL */
  const float* bias;          /* [B,H,Lq,Lk] or NULL */
  /* backward only */
  const void* dout;
  void* dq; void* dk; void* dv;
  float* delta;
  float* dbias;
  /* key split: ksplit > 1 runs `ksplit` workgroups per (scene, head, query chunk), each over a slice of the key
   * blocks, and a small combine kernel merges the partial softmax states / dQ sums.  `ws` is a caller-provided fp32
   * workspace of at least ksplit * B * H * Lq * (dh + 2) elements.  Raises occupancy when B*H*ceil(Lq/128) is far
   * below the CU count and shortens the serial key loop; results are deterministic.  dbias requires ksplit == 1. */
  int32_t ksplit;
  float* ws;
  /* attention-probability dropout of nn.MultiheadAttention (functional.py: dropout(softmax(..)) before @ v): the
   * site is the probability matrix viewed as [B*H*Lq, Lk] (the zero key of add_zero_attn has no value row, so
   * dropping it is a no-op and it draws nothing).  drop_bmod > 0: scenes are stacked groups of drop_bmod along the
   * batch (memories of one layer): batch entry b uses site drop.site + b / drop_bmod and row index of scene
   * b % drop_bmod, so a stacked launch draws exactly the masks the per-memory launches would. */
  pq3d_dropout drop;
  int32_t drop_bmod;
  pq3d_attn_proj proj;   /* zero-initialised = none */
  /* optional bit form of `mask` with `row_open` folded in (pq3d_mask_pack): [B or mask_bmod, Lq, ceil(Lk / 32)] words, bit j
   * of word w of a row = key 32 w + j is masked for that query (an open row is all zero).  1/8 of the mask bytes and no
   * per-element byte loads: the all-queries-resident backward uses it when present (config 4: 8 launches per step whose
   * 3-D mask handling was +60 % of their time); `mask` / `row_open` must still be set (the other kernels read them). */
  const uint32_t* mask_bits;
  /* split-bf16 cross-attention FORWARD (compute mode 'bf16x3', csrc/attn_x3.hip): ct = PQ3D_BF16X3, q and o fp32 (dt = PQ3D_F32),
   * the keys / values as bf16 hi / lo PLANES (k, v = hi planes, k_lo, v_lo = residual planes with the same strides; written by
   * pq3d_gemm's PQ3D_ACT_PLANES epilogue).  Scores and the value contraction are 3 bf16 MFMAs per term pair (q, P split in
   * registers): fp32-grade results at bf16 MFMA rate, K / V bytes = an fp32 tensor's.  q_bf / o_bf (optional, strides of q / o):
   * bf16 copies of q and of the output for the (single-bf16) backward, which then reads exactly a 'bf16'-mode forward's tensors.
   * Shape: d_h = 32 or 64, Lq <= 256 (two query halves above 128), key padding / zero key / mask_bits, at most 1024 keys per key
   * split at d_h = 32 and 512 at d_h = 64 (ksplit >= ceil(Lk / 1024) resp. ceil(Lk / 512)); any other call with k_lo set is refused. */
  const void* k_lo;
  const void* v_lo;
  void* q_bf;
  void* o_bf;
} pq3d_attn_desc;

/* row_open[r] = every byte of mask row r is non-zero (query_encoder.py:83: such rows attend everywhere) and, when bits != NULL,
 * bits[r, w] = the row's mask as bits (bit j of word w = mask[r, 32 w + j] != 0; all zero for an open row; keys past Lk: 0).
 * One pass over the mask bytes; mask [rows, Lk] uint8, bits [rows, ceil(Lk / 32)] uint32. */
int pq3d_mask_pack(const uint8_t* mask, uint8_t* row_open, uint32_t* bits, int64_t rows, int64_t Lk, void* stream);
int pq3d_attn_fwd(const pq3d_attn_desc* d, void* stream);
int pq3d_attn_bwd(const pq3d_attn_desc* d, void* stream);
/* pq3d_attn_fwd / pq3d_attn_bwd pick specialised implementations where the call has their shape, and the general
 * streaming kernels otherwise:
 *   bit 0: the all-queries-resident single-pass backward (cross-attention shape: bf16, d_h 32 / 64, Lq <= 256
 *          (129..256 queries: two passes over the query halves), Lk >= 128, no additive bias);
 *   bit 1: the small-sequence fp32 kernels (self-attention shape: fp32, Lq, Lk <= 128, key padding / additive bias only,
 *          one workgroup per (scene, head));
 *   bit 2: the all-keys-resident forward (cross-attention shape: bf16, d_h 32, Lq <= 128, Lk >= 128, key-padding mask
 *          only, at most 1024 keys per key split: ksplit >= ceil(Lk / 1024)).
 *   bit 3: the split-bf16 MFMA self-attention kernels (csrc/attn_sa.hip) for compute type PQ3D_BF16X3 -- fp32 storage,
 *          fp32-grade arithmetic (hi/lo bf16 operand pairs, 3 MFMAs per product): d_h 32, Lq, Lk <= 240, key padding /
 *          additive bias only, one workgroup per (scene, head), no atomics.  PQ3D_BF16X3 calls of any other shape run the
 *          exact-fp32 kernels.
 *   bit 4: the small bf16 cross-attention kernels (csrc/attn_ca.hip): bf16 storage and compute, d_h 32 / 64, at most 128
 *          queries AND 128 keys per (scene, head), key padding / zero key only -- the shipped stage-2 decoder's attention
 *          over <= 80 objects per memory and its prompt tokens; one workgroup per (scene, head), no atomics.
 *   bit 5: the all-keys-resident forward also for key-split calls (a longer scene as slices of <= 1024 keys; config 5).
 * This process-wide switch sets which of them may be used (default 63 = all; for A/B measurements and tests) and returns
 * the previous value. */
int pq3d_attn_resident(int enable);

/* Small glue operations of the step, each ONE launch (they replace chains of framework elementwise / cat / reduce kernels
 * inside the captured step):
 *   pq3d_mask_not      : dst_g[i] = !src_g[i] for up to PQ3D_MAX_GROUPS byte masks of individual lengths (data_dict's
 *                        'True = valid' pad masks -> PyTorch's 'True = ignore', query3d_unified.py:113,139,143,148,155)
 *   pq3d_zero_many     : zero-fill n fp32 buffers (gradient arena, atomics targets)
 *   pq3d_sum_n         : out = sum_g src_g, fixed order (d query_pos over the layers)
 *   pq3d_mean_all      : out[0] = mean(x): per-block partial sums + last-arriver combine in block order (deterministic,
 *                        one launch; ws = counter + partials, self-resetting); pq3d_fill_scaled: dst[i] = scalar[0] * c
 *                        (its gradient) -- the synthetic 'mean(query)' loss of SURVEY 8d
 *   pq3d_cast_transpose: src_g [rows, cols] fp32 -> out_g bf16 (same layout) and outT_g bf16 with every cols x cols row
 *                        block transposed (outT[t][k][n] = src[t cols + n][k]): the K/V projection weights once per step
 *                        for the forward (NT) and the input-gradient (NT on W^T) products */
int pq3d_mask_not(const uint8_t* const* src, uint8_t* const* dst, const int64_t* counts, int32_t groups, void* stream);
int pq3d_zero_many(float* const* bufs, const int64_t* counts, int32_t n, void* stream);
/* dst_i[0 .. counts_i) = src_i[..] (fp32) for n (source, destination, length) triples in one launch per 64 triples: the
 * gradient pack of the data-parallel step (parameter gradients -> their slices of a flat bucket; DDP's bucket copy,
 * trainer/build.py:66-75). */
int pq3d_copy_many(const float* const* src, float* const* dst, const int64_t* counts, int32_t n, void* stream);
int pq3d_sum_n(const float* const* src, int32_t n, float* out, int64_t count, void* stream);
/* two such sums of different lengths in one launch (d query_pos and d pos at the end of the decoder backward, written
 * into the two halves of one buffer: the coordinate encoder ran on their row concatenation) */
int pq3d_sum_pair(const float* const* src_a, int32_t na, float* out_a, int64_t count_a, const float* const* src_b,
                  int32_t nb, float* out_b, int64_t count_b, void* stream);
#define PQ3D_MEAN_MAX_BLOCKS 256   /* ws: fp32[1 + PQ3D_MEAN_MAX_BLOCKS], element 0 (the arrival counter) zero before first use */
int pq3d_mean_all(const float* x, int64_t n, float* out, float* ws, void* stream);
/* out[0] = sum_g mean(f_g(x_g)) over n <= PQ3D_MAX_GROUPS fp32 tensors of counts[g] elements, deterministic: one launch of per-block
 * partial sums + a one-block combine launch (an in-kernel last-arriver combine was 4x slower: every block's agent-scope
 * release fence writes the L2 back) (modes[g]: 0 identity, 1 clamp(min = clamp_min[g]), 2 non-finite elements count as 0), and its gradient
 * dx_g[i] = gout[0] / counts[g] * f_g'(x_g[i]) in one more -- the synthetic loss of the mask configurations (SURVEY 8d:
 * sum over prediction layers of mean(clamp(mask_logits, -50)) + mean(class logits, filtered -inf columns dropped)).
 * ws: fp32[1 + 130 * PQ3D_MAX_GROUPS] scratch (the partial sums). */
int pq3d_mean_many(const float* const* x, const int64_t* counts, const int32_t* modes, const float* clamp_min, int32_t n,
                   float* out, float* ws, void* stream);
int pq3d_mean_many_bwd(const float* const* x, float* const* dx, const int64_t* counts, const int32_t* modes,
                       const float* clamp_min, int32_t n, const float* gout, void* stream);
int pq3d_fill_scaled(float* dst, int64_t n, const float* scalar, float c, void* stream);
int pq3d_cast_transpose(const float* const* src, void* const* out, void* const* outT, int32_t groups, int32_t rows,
                        int32_t cols, void* stream);

/* For every (b,i): row_open[b,i] = all_j(mask[b,i,j] != 0)   (query_encoder.py:83) */
int pq3d_mask_row_all(const uint8_t* mask, uint8_t* row_open, int64_t rows, int64_t Lk, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Residual-add + LayerNorm, optionally merged over M branches (post-norm sublayers, query_encoder.py:304-305,
 * :224-225, :386-387, :449-450 and the parallel cross-attention mean :152 / memory-dropout mean :145-151):
 *     y[r,:] = sum_m coef[m][r / rows_per_scene] * LN_m( x[r,:] + o_m[r,:] )        (coef NULL -> 1/M)
 * x may be NULL (plain LN of o_m: nn.Sequential(Linear, LayerNorm) encoders, get_mlp_head's LN eps=1e-12).
 * mean/rstd ([M,R] fp32) are saved for backward.  Backward: dx = sum_m d(x+o_m), d_o[m], dgamma[m], dbeta[m]
 * (dgamma/dbeta are zeroed by the call and accumulated with atomics).  One 64-lane wave per row, row kept in
 * registers (d <= 2048), two-pass statistics in fp32.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t R, d, M, rows_per_scene;
  int32_t dt_x, dt_o, dt_y;
  float eps;
  const void* x;
  const void* o[PQ3D_MAX_GROUPS];
  const float* gamma[PQ3D_MAX_GROUPS];
  const float* beta[PQ3D_MAX_GROUPS];
  const float* coef;
  void* y;
  float* mean;
  float* rstd;
  /* backward */
  const float* dy;
  float* dx;                      /* [R,d] fp32 or NULL */
  float* d_o[PQ3D_MAX_GROUPS];    /* [R,d] fp32 each */
  float* dgamma[PQ3D_MAX_GROUPS];
  float* dbeta[PQ3D_MAX_GROUPS];
  int32_t accumulate;             /* != 0: dgamma/dbeta are accumulated onto (not zeroed by the call) */
  int32_t independent;            /* != 0: the M branches are independent problems sharing one launch:
                                     ys[m] = LN_m(x + o_m) (no sum, coef ignored); backward reads dys[m], dx unused */
  void* ys[PQ3D_MAX_GROUPS];
  const float* dys[PQ3D_MAX_GROUPS];
  /* sum_branches != 0: o[0..M) are PARTIAL SUMS of a single branch (the producing GEMM split its K range over M
   * groups without atomics): y = LN(x + dropout(sum_m o_m)) with gamma[0] / beta[0], statistics at index 0; backward
   * writes d_o[0] = gradient w.r.t. the sum, dx, dgamma[0], dbeta[0]. */
  int32_t sum_branches;
  float* osum;   /* forward, sum_branches only, optional: receives sum_m o_m [R, d] (fp32) */
  /* residual dropout: y = LN(x + dropout(o_m)) (tgt + self.dropout(tgt2), query_encoder.py:224,304,386); the site
   * is o_m viewed as [R, d], site id drop.site + m.  The backward regenerates the mask for d_o. */
  pq3d_dropout drop;
  int32_t dx_zeroed;   /* backward, M > 1 summed branches: dx (accumulated with atomics by the M branch blocks) was
                          already zeroed by the caller -- one zero-fill for a whole backward pass instead of one per call */
  const float* dy2;    /* backward, optional: the upstream gradient is (dy + dy2) + dy3 -- the producers' outputs are summed */
  const float* dy3;    /* here instead of by one more launch (or a dependent "+ aux" chain) in front of this one */
} pq3d_ln_desc;

int pq3d_add_ln_fwd(const pq3d_ln_desc* d, void* stream);
int pq3d_add_ln_bwd(const pq3d_ln_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The row-local tail of a decoder layer's forward in ONE launch (bf16 mode's split-bf16 query side, ReLU, no dropout):
 *     f = o_s Wo^T + bo;  x2 = LN1(x1s + f);  h = relu(x2 W1^T + b1);  zp_k = h_k W2_k^T (+ b2, k = 0; K split in 4);
 *     z = ((zp_0 + zp_1) + zp_2) + zp_3;  x3 = LN2(x2 + z)
 * = self-attention out-projection + post-norm and FFNLayer (query_encoder.py:224-225, 384-388), bit for bit what
 * pq3d_gemm / pq3d_add_ln_fwd produce in five launches.  A group of 8 workgroups on one XCD owns a 32-row tile for all
 * five steps and hands rows over through that XCD's L2 (csrc/chain_ffn.hip).  R <= 2048 rows, d = 256, F = 2048.
 * flags: >= ceil(R / 32) * 128 uint32 words, zeroed ONCE when allocated, never touched by the caller afterwards, not
 * shared between call sites that can be in flight together.  err (optional): set to 1 if a hand-off wait gave up.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t R, d, F;
  float eps1, eps2;
  const float* o_s;             /* [R, d] self-attention output */
  const float* Wo;              /* [d, d] */
  const float* bo;              /* [d] */
  const float* x1s;             /* [R, d] residual of the first LayerNorm */
  const float* g1;
  const float* be1;
  float* f;                     /* [R, d] out: out-projection (saved for the backward) */
  float* x2;                    /* [R, d] out */
  float* mean1;                 /* [R] out */
  float* rstd1;                 /* [R] out */
  const float* W1;              /* [F, d] */
  const float* b1;              /* [F] */
  float* h;                     /* [R, F] out */
  const float* W2;              /* [d, F] */
  const float* b2;              /* [d] */
  float* zp;                    /* [4, R, d] out: partial sums of linear2 */
  float* z;                     /* [R, d] out: their sum */
  const float* g2;
  const float* be2;
  float* x3;                    /* [R, d] out */
  float* mean2;                 /* [R] out */
  float* rstd2;                 /* [R] out */
  uint32_t* flags;
  int32_t* err;
  /* optional sixth step: the NEXT layer application's cross-attention query projections q_m = (x3 + qpos) Wq_m^T + bq_m, m < nq <= 3
   * (split-bf16, bf16 output: what pq3d_gemm writes for it) -- one launch less per layer; nq = 0: none */
  int32_t nq;
  const float* qpos;            /* [R, d] */
  const float* Wq[3];           /* [d, d] */
  const float* bq[3];
  void* qout[3];                /* [R, d] out: bf16, or fp32 when qout_f32 != 0 */
  int32_t qout_f32;
  /* optional step 0 (round 6): the self-attention CORE of the layer inside this launch -- what pq3d_attn_fwd (ct PQ3D_BF16X3,
   * d_h = 32, 8 heads, additive bias, key padding) writes for q / k / v [R, d] of scenes of sa_nq rows each, bit for bit
   * (csrc/attn_sa_body.h's loop): o_s then is an OUTPUT of the launch, sa_lse [R / sa_nq, 8, sa_nq] too.  sa_q = NULL: o_s is an
   * input as before.  sa_bias [R / sa_nq, 8, sa_nq, sa_nq] or NULL, sa_kpm [R / sa_nq, sa_nq] (non-zero = padded key) or NULL. */
  const float* sa_q;
  const float* sa_k;
  const float* sa_v;
  const float* sa_bias;
  const uint8_t* sa_kpm;
  float* sa_lse;
  int32_t sa_nq;
  float sa_scale;
} pq3d_chain_ffn_desc;
int pq3d_chain_ffn_fwd(const pq3d_chain_ffn_desc* d, void* stream);
/* Device gate of every pq3d_chain_* entry point: 1 if the current device of `stream` is the one their in-launch hand-offs are valid
 * on -- gfx950, 256 CUs, 160 KB of LDS per workgroup, and (probe != 0: one 256-workgroup launch + a synchronous copy, cached per
 * device; do not call with probe != 0 while the stream is capturing) the measured placement rule "XCC_ID == (workgroup id + c) % 8, c constant" for
 * all 256 workgroups -- else 0: the caller must use the separate launches (same bits).  The err word of a chain descriptor reports
 * a hand-off that gave up at run time (a member that never became resident: CU-masked queue, a long-running neighbour kernel). */
int pq3d_chain_device_ok(int32_t probe, void* stream);
/* Test support (tests/test_gpu_chain.py): `workgroups` one-wave workgroups that each hold `lds_bytes` of a CU's LDS and spin for
 * `microseconds` on `stream` -- a stand-in for a collective of another stream (RCCL: one workgroup per channel) that keeps CUs busy
 * beside a chain launch (trainer/build.py:66-75: DDP's all-reduce overlaps the backward). */
int pq3d_test_occupy_cus(int32_t workgroups, int32_t lds_bytes, int64_t microseconds, void* stream);

/* The row-local steps between a layer's cross-attention and its self-attention in ONE launch (csrc/chain_ca.hip; bf16 mode, no
 * residual dropout): op_m = o_m Wo_m^T + bo_m (m < M <= 3, o_m bf16), x1 = sum_m c_m LN_m(x + op_m) (c_m = coef[m][row /
 * rows_per_scene] or 1 / M), q = (x1 + qpos) Wq^T + bq, k = (x1 + qpos) Wk^T + bk, v = x1 Wv^T + bv -- bit for bit what pq3d_gemm,
 * pq3d_add_ln_fwd, pq3d_gemm produce (CrossAttentionLayer out_proj + post-norm, query_encoder.py:145-152, 304-305; self-attention
 * projections, transformers.py:190-193).  R <= 2048 rows, d = 256.  flags / err: as for pq3d_chain_ffn_fwd (own words per site). */
typedef struct {
  int32_t R, d, M, rows_per_scene;
  float eps;
  const void* o[3];             /* [R, d] bf16 attention outputs (fp32 with o_f32) */
  const float* Wo[3];           /* [d, d] */
  const float* bo[3];
  const float* x;               /* [R, d] residual */
  const float* gamma[3];
  const float* beta[3];
  const float* coef;            /* [M, R / rows_per_scene] or NULL */
  float* op[3];                 /* [R, d] out */
  float* x1;                    /* [R, d] out */
  float* mean;                  /* [M, R] out */
  float* rstd;                  /* [M, R] out */
  const float* qpos;            /* [R, d] */
  const float* Wqkv[3];         /* [d, d] each: q, k, v */
  const float* bqkv[3];
  float* qkv[3];                /* [R, d] out each */
  uint32_t* flags;
  int32_t* err;
  int32_t o_f32;                /* != 0: o[] are fp32 and the out-projections split-bf16 products (compute mode 'bf16x3') */
} pq3d_chain_ca_desc;
int pq3d_chain_ca_fwd(const pq3d_chain_ca_desc* d, void* stream);

/* The row-local part of one MaskHeadSegLevel call in ONE launch (csrc/chain_mh.hip; bf16 mode, no head dropout):
 * h1 = relu(x W0^T + b0), qm_m = x Wq_m^T + bq_m (m < Mm <= 3), h2 = LN(h1), cls = h2 W4^T + b4 with the columns flagged in
 * colfill set to `fill` -- bit for bit what pq3d_gemm, pq3d_add_ln_fwd, pq3d_gemm, pq3d_fill_cols, pq3d_gemm produce (class MLP
 * of get_mlp_head, utils.py:17-26, called at mask_head.py:36; MaskPredictionLayer's query projection, mask_head.py:57-60).
 * R <= 2048 rows, d = hidden = 256, C <= 256 classes.  flags / err: as for pq3d_chain_ffn_fwd (own words per site). */
typedef struct {
  int32_t R, d, Mm, C;
  float eps, fill;
  const float* x;               /* [R, d] queries */
  const float* W0;              /* [d, d] */
  const float* b0;
  const float* gamma;
  const float* beta;
  const float* W4;              /* [C, d] */
  const float* b4;              /* [C] or NULL */
  const int32_t* colfill;       /* [C] (non-zero: filled column) or NULL */
  const float* Wq[3];           /* [d, d] */
  const float* bq[3];
  float* h1;                    /* [R, d] out */
  float* h2;                    /* [R, d] out */
  float* mean;                  /* [R] out */
  float* rstd;                  /* [R] out */
  float* cls;                   /* [R, C] out */
  float* qm[3];                 /* [R, d] out */
  uint32_t* flags;
  int32_t* err;
} pq3d_chain_mh_desc;
int pq3d_chain_mh_fwd(const pq3d_chain_mh_desc* d, void* stream);

/* Backward of the same part in ONE launch (csrc/chain_mh.hip): dcl = dc with the flagged columns zeroed, dh2 = dcl W4,
 * dh1 = LN'(h1; dh2) (dgamma / dbeta accumulated), dpre = [h1 > 0] dh1 (bf16), out = sum_m dq_m Wq_m + (dpre W0 + cur) -- what
 * pq3d_fill_cols, pq3d_gemm, pq3d_add_ln_bwd, pq3d_act_bwd, pq3d_gemm, pq3d_gemm produce (single-bf16 products, fp32
 * accumulation; the LayerNorm parameter gradients to summation order).  dq_m: the query-side gradient of the mask logits
 * (fp32 when dq_f32, else bf16); the weight gradients themselves stay with the caller (they read dcl, dpre, dq_m).
 * lnws: scratch of 32 * 8 * 512 floats. */
typedef struct {
  int32_t R, d, Mm, C, dq_f32;
  const float* dc;              /* [R, C] gradient of the class logits */
  const int32_t* colfill;       /* [C] or NULL */
  float* dcl;                   /* [R, C] out (NULL allowed without colfill: dcl = dc) */
  const float* W4;              /* [C, d] */
  const float* h1;              /* [R, d] */
  const float* mean;            /* [R] */
  const float* rstd;            /* [R] */
  const float* gamma;
  float* dgamma;                /* accumulated */
  float* dbeta;                 /* accumulated */
  float* dh2;                   /* [R, d] out (scratch of the launch) */
  void* dpre;                   /* [R, d] bf16 out */
  const float* W0;              /* [d, d] */
  const float* cur;             /* [R, d] gradient arriving at the queries from elsewhere */
  const void* dq[3];            /* [R, d] */
  const float* Wq[3];           /* [d, d] */
  float* out;                   /* [R, d] out */
  uint32_t* flags;
  int32_t* err;
  float* lnws;
  /* optional: cur = sum_m dqc_m Wqc_m + dxr formed in the launch (m < nq <= 3, dqc_m bf16; gq receives the sum without dxr) --
   * the input gradient of the cross-attention query projections that pq3d_gemm (kconcat, C2) forms otherwise; nq = 0: cur is read */
  int32_t nq;
  const void* dqc[3];           /* [R, d] bf16 */
  const float* Wqc[3];          /* [d, d] */
  const float* dxr;             /* [R, d] */
  float* gq;                    /* [R, d] out */
} pq3d_chain_mh_bwd_desc;
int pq3d_chain_mh_bwd(const pq3d_chain_mh_bwd_desc* d, void* stream);

/* The row-local head of a decoder layer's BACKWARD in one launch (csrc/chain_ffn_bwd.hip; bf16 mode, ReLU, no dropout):
 *     g2 = LN2'(x2 + z; dx);  dhp = [h > 0] (g2 W2) (bf16);  p_k = dhp_k W1_k (K = F in 4 partial sums);
 *     g1 = LN1'(x1s + f; g2 + p_0 + p_1 + p_2 + p_3)
 * = pq3d_add_ln_bwd, pq3d_gemm (transB, act_grad relu), pq3d_gemm (transB, split-K), pq3d_add_ln_bwd of FFNLayer and the
 * self-attention post-norm (query_encoder.py:384-388, 224-225).  dy (= g2: d z and the residual branch), dhp and df (= g1: d f and
 * the residual branch) are left in memory for the weight-gradient products; d gamma / d beta are ACCUMULATED (atomics) onto
 * dg2 / db2 / dg1 / db1.  R <= 2048, d = 256, F = 2048.  part: [4, R, d] fp32 workspace.  flags / err: as for pq3d_chain_ffn_fwd. */
typedef struct {
  int32_t R, d, F;
  const float* dx;              /* [R, d] upstream gradient */
  const float* x2;              /* [R, d] */
  const float* z;               /* [R, d] */
  const float* g2;              /* LN2 gamma */
  const float* mean2;           /* [R] */
  const float* rstd2;
  float* dg2;                   /* [d] accumulated */
  float* db2;
  float* dy;                    /* [R, d] out: g2 */
  const float* W2;              /* [d, F] */
  const float* h;               /* [R, F] forward hidden (post-ReLU) */
  void* dhp;                    /* [R, F] bf16 out */
  const float* W1;              /* [F, d] */
  float* part;                  /* [4, R, d] workspace */
  const float* x1s;             /* [R, d] */
  const float* f;               /* [R, d] */
  const float* g1;              /* LN1 gamma */
  const float* mean1;
  const float* rstd1;
  float* dg1;
  float* db1;
  float* df;                    /* [R, d] out: g1 */
  uint32_t* flags;
  int32_t* err;
  float* lnws;                  /* >= 32 * 8 * 1024 floats of scratch (the groups' LayerNorm parameter-gradient partials) */
  /* optional step 0 (nq > 0; dx is then ignored): the upstream gradient is formed here, dxo = sum_{m < nq} dq_m Wq_m + dxr and
   * gq = the same sum without dxr -- pq3d_gemm (transB, kconcat = nq, C2, "+ aux") of the cross-attention query projections of
   * the layer application that ran backward just before (its d query_pos term and this layer's input gradient) */
  int32_t nq;
  const void* dq[3];            /* [R, d] bf16 */
  const float* Wq[3];           /* [d, d] */
  const float* dxr;             /* [R, d] */
  float* gq;                    /* [R, d] out */
  float* dxo;                   /* [R, d] out */
} pq3d_chain_ffn_bwd_desc;
int pq3d_chain_ffn_bwd(const pq3d_chain_ffn_bwd_desc* d, void* stream);

/* The row-local steps between a layer's self-attention backward and its cross-attention backward in one launch
 * (csrc/chain_sa_bwd.hip; bf16 mode, no residual dropout, no prompt memory): g3 = [dq Wq, dk Wk, dv Wv + aux2]; for m < M <= 3:
 * dop_m = LN_m'(x + op_m; c_m ((g3_0 + g3_1) + g3_2)), dxr = sum_m dop_m, d gamma_m / d beta_m accumulated; do_m = dop_m Wo_m (bf16)
 * = pq3d_gemm (transB, "+ aux"), pq3d_add_ln_bwd (merged), pq3d_gemm (transB) -- the same bits at M = 3.  R <= 2048, d = 256. */
typedef struct {
  int32_t R, d, M, rows_per_scene;
  const float* dqkv[3];         /* [R, d] gradients of q, k, v */
  const float* Wl[3];           /* [d, d] Wq, Wk, Wv */
  const float* aux2;            /* [R, d] added to the v part */
  float* g3[3];                 /* [R, d] out */
  const float* x;               /* [R, d] layer input (residual of the merged LayerNorm) */
  const float* op[3];           /* [R, d] forward out-projections */
  const float* gamma[3];
  const float* mean;            /* [M, R] */
  const float* rstd;
  const float* coef;            /* [M, R / rows_per_scene] or NULL (1 / M) */
  float* dop[3];                /* [R, d] out */
  float* dxr;                   /* [R, d] out */
  float* dgamma[3];             /* [d] accumulated */
  float* dbeta[3];
  const float* Wo[3];           /* [d, d] */
  void* do_all[3];              /* [R, d] bf16 out */
  uint32_t* flags;
  int32_t* err;
  float* lnws;                  /* >= 32 * 8 * 1536 floats of scratch */
} pq3d_chain_sa_bwd_desc;
int pq3d_chain_sa_bwd(const pq3d_chain_sa_bwd_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Small memory-bound kernels.
 * ------------------------------------------------------------------------------------------------ */
/* out[n] = sum_r x[r*ld + n]  (bias gradients). */
int pq3d_colsum(const void* x, int32_t dt, int64_t R, int64_t N, int64_t ld, float* out, void* stream);
/* Grouped form: out_g[n] (+)= sum_r x_g[r*ld + n] for g < groups (host arrays of device pointers).
 * accumulate == 0: every out_g[n] is WRITTEN by exactly one workgroup, fixed summation order (deterministic).
 * accumulate == 1: added onto out_g, which must hold valid numbers (a pre-zeroed or running gradient slot).  Up to 2047 rows
 *   one writer per element; from 2048 rows on the rows are cut into slices that add with fp32 atomics (bias gradients of the
 *   10 240-row shipped shapes: 12-36 workgroups walking every row otherwise) -- the sum is then order-dependent in the last
 *   bit.  accumulate == 2: as 1 but ALWAYS one writer per element (bit-reproducible; what fingerprint / bit-exactness tests
 *   and a deterministic training mode ask for). */
int pq3d_colsum_grouped(const void* const* x, float* const* out, int32_t groups, int32_t dt, int64_t R, int64_t N,
                        int64_t ld, int32_t accumulate, void* stream);

/* y[r,c] = keep(r) ? x[r,c] * scale[r] : 0, keep(r) = (!zero_flag || !zero_flag[r]) && (!keep_mask || keep_mask[r]);
 * scale NULL -> 1.  (backward of mask_head.py:35-38 and of row-masked projections) */
int pq3d_scale_rows(const void* x, int32_t dtx, void* y, int32_t dty, int64_t R, int64_t C, const float* scale,
                    const uint8_t* zero_flag, const uint8_t* keep_mask, void* stream);
/* the same for `groups` <= PQ3D_MAX_GROUPS fp32 tensors of one shape sharing scale / flags (C % 8 == 0, 16-byte aligned): one launch */
int pq3d_scale_rows_grouped(const float* const* x, void* const* y, int32_t groups, int32_t dty, int64_t R, int64_t C,
                            const float* scale, const uint8_t* zero_flag, const uint8_t* keep_mask, void* stream);

/* Grouped elementwise add + cast: out_g[i] = (dt_out)(a_g[i] + b_g[i]) for g < groups (b_g may be NULL); fp32 inputs.
 * Used once per forward to materialise the layer-invariant bf16 MFMA operands (feat + pos) and (feat) of every scene
 * memory, so the hoisted K/V projection GEMMs read 2 B instead of 8 B per element. */
int pq3d_add_cast(const float* const* a, const float* const* b, void* const* out, int32_t groups, int32_t dt_out,
                  int64_t n, void* stream);

/* Operand planes of the split-bf16 products (compute mode 'bf16x3'): for g < groups, v = a_g[i] + b_g[i] (b_g may be NULL),
 * hi_g[i] = bf16(v) (round to nearest even), lo_g[i] = bf16(v - hi_g[i]); hi_g may be NULL (only the residual plane is written).
 * counts[g] elements (multiples of 8), 16-byte aligned pointers.  Written once per step for tensors that many launches read as
 * MFMA operands (the memories' tokens, the K / V rows of in_proj_weight: query_encoder.py:288-307's key / value projections). */
int pq3d_split_planes(const float* const* a, const float* const* b, void* const* hi, void* const* lo, const int64_t* counts,
                      int32_t groups, void* stream);

/* out[r,:] = x[r,:] + bias[:]  (fp32): seeds the output of a split-K GEMM with residual + bias so that
 * LayerNorm(x + h W2^T + b2) (query_encoder.py:385-387) needs no separate epilogue pass. */
int pq3d_bias_add_rows(const float* x, const float* bias, float* out, int64_t R, int64_t N, void* stream);

/* dpre = dy * act'(saved): ReLU: saved = activation output (or pre-activation), GELU: saved = pre-activation. */
int pq3d_act_bwd(const void* dy, int32_t dt_dy, const void* saved, int32_t dt_saved, void* dpre, int32_t dt_dpre,
                 int32_t act, int64_t n, void* stream);

/* y = x with columns cols[0..ncols) set to `value` (mask_head.py:28 `cls_logits[..., filter] = -inf`). */
int pq3d_fill_cols(const float* x, float* y, int64_t R, int64_t C, const int32_t* cols, int32_t ncols, float value,
                   void* stream);

/* mask_head.py:33-37 denominators: inv_den[b,s] = 1 / (sum_m !mask_m[b,s] + 1e-8) over M uint8 masks. */
int pq3d_mask_inv_den(const uint8_t* const* masks, int32_t M, int64_t n, float* inv_den, void* stream);

/* calc_pairwise_locs (modules/utils.py:38-87; 'center', spatial_dist_norm, spatial_dim=5):
 * centers [B,L,3] fp32 -> out [B,L,L,5] fp32. */
int pq3d_pairwise_locs(const float* centers, int64_t center_stride, float* out, int32_t B, int32_t L, float eps,
                       void* stream);

/* Fourier positional features (position_embedding.py:127-156 + shift_scale_points :13-43):
 * xyz [B,N,*] (row stride xyz_stride) , cmin/cmax [B,3], gauss_B [3,half] -> out [B,N,2*half] = [sin | cos]. */
int pq3d_fourier(const float* xyz, int64_t xyz_stride, const float* cmin, const float* cmax, const float* gauss_B,
                 float* out, int32_t B, int32_t N, int32_t half, void* stream);
/* the same for two point sets of the same B scenes (queries [B,Na,*], segments [B,Nb,*]; cmin / cmax / gauss_B shared) in
 * one launch: out = [B*Na rows of set a | B*Nb rows of set b] x 2*half -- the row-concatenated input of the one
 * Linear + LayerNorm pass both sets share (query3d_unified.py:117-128 calls the CoordinateEncoder once per set). */
int pq3d_fourier_pair(const float* xyz_a, int64_t stride_a, int32_t Na, const float* xyz_b, int64_t stride_b, int32_t Nb,
                      const float* cmin, const float* cmax, const float* gauss_B, float* out, int32_t B, int32_t half,
                      void* stream);

/* MultiHeadAttentionSpatial 'mul' fusion bias (transformers.py:196-200,226):
 * bias[b,h,i,j] = log(max(relu(W[h,:].pl[b,i,j,:] + bw[h]), 1e-6));  backward accumulates dW [H,5], dbw [H]
 * (zeroed by the call) from dbias. */
int pq3d_spatial_bias_fwd(const float* pl, const float* W, const float* bw, float* bias, int32_t B, int32_t H,
                          int32_t L, void* stream);
/* `groups` layers sharing pairwise_locs in one launch (host pointer arrays of device pointers) */
int pq3d_spatial_bias_fwd_grouped(const float* pl, const float* const* W, const float* const* bw, float* const* bias,
                                  int32_t groups, int32_t B, int32_t H, int32_t L, void* stream);
int pq3d_spatial_bias_bwd(const float* pl, const float* W, const float* bw, const float* dbias, float* dW, float* dbw,
                          int32_t B, int32_t H, int32_t L, void* stream);
/* same, accumulating onto dW/dbw instead of zeroing them first */
int pq3d_spatial_bias_bwd_acc(const float* pl, const float* W, const float* bw, const float* dbias, float* dW,
                              float* dbw, int32_t B, int32_t H, int32_t L, void* stream);
/* the same for `groups` (<= PQ3D_MAX_GROUPS) layers sharing pairwise_locs in ONE launch (accumulating; host pointer
 * arrays of device pointers): the fused executor defers the layers' spatial-bias parameter gradients to the end of the
 * backward pass */
int pq3d_spatial_bias_bwd_grouped(const float* pl, const float* const* W, const float* const* bw,
                                  const float* const* dbias, float* const* dW, float* const* dbw, int32_t groups,
                                  int32_t B, int32_t H, int32_t L, void* stream);

/* gate structure (query_encoder.py:166-170): y = (1-s)*q + s*u, s = sigmoid(g);  bwd: dq, du, dg. */
int pq3d_gate_mix_fwd(const float* q, const float* u, const float* g, float* y, int64_t n, void* stream);
int pq3d_gate_mix_bwd(const float* q, const float* u, const float* g, const float* dy, float* dq, float* du,
                      float* dg, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Segment pooling: torch_scatter.scatter_mean(feat, point2segment, dim=0, dim_size=max_seg)
 * (modules/vision/pcd_mask3d_encoder.py:149; on coordinates data/datasets/sceneverse_instseg.py:183,186), the
 * MinkowskiPoolingTranspose up-sampling in front of it (pcd_mask3d_encoder.py:127-131,147) as an index composition, and the
 * inverse gathers mask[voxel2segment] (evaluator/instseg_eval.py:101,272-281).  Bandwidth kernels: no atomics, no zero
 * fill, every result bit-identical run to run (pq3d_amd/csrc/segment.hip).
 *
 * pq3d_segment_plan sorts the voxel -> segment ids of a batch ONCE (stable: ascending voxel id inside a segment; ids outside
 * [0, S) are dropped) into the caller's plan buffer of pq3d_segment_plan_bytes(N, S) bytes (16-byte aligned device memory;
 * opaque: sorted voxel order, per-segment offsets, work items).  The plan serves every reduction over that grouping: the 5
 * feature levels of a batch and, keyed by the coarse parent instead, the gradient of an up-sampled level.
 *
 * pq3d_segment_reduce: out[s,:] = sum or mean over the voxels v of segment s of  w(v) * src[row(v),:]   (out [S,C] fp32,
 *   every row written, empty segments zero), with row(v) = v (gather NULL: src has a row per voxel, torch_scatter's
 *   scatter_mean) or gather[v] (src [Nsrc,C] = a COARSE level, gather = the composed fine -> coarse row index: the
 *   up-sampled [N,C] intermediate of the reference never exists); rows outside [0, Nsrc) contribute nothing and are not
 *   counted.  w(v) = 1 / max(row_div[row(v)], 1) or 1 (NULL).  count (optional, [S] fp32) receives the number of rows summed; mean
 *   divides by max(count, 1).  ws: pq3d_segment_ws_bytes(N, S, C) bytes of scratch (partial rows of segments longer than one
 *   128-voxel piece), 16-byte aligned.
 *   Gradient of the up-sampled mean w.r.t. the coarse level = the same call with the plan of `parent`, src = dout [S,C],
 *   gather = index, row_div = count (per segment), mean = 0.
 *
 * pq3d_segment_gather: out[v,:] = table[index[v],:] (* 1 / max(count[index[v]], 1) when count is given) for v < N, zero rows
 *   for ids outside [0, S): the gradient of the plain segment mean (table = dout) and the evaluator's voxel <- segment
 *   gathers.  16-byte vector rows when C % 4 == 0 and the pointers are 16-byte aligned, scalar rows otherwise. */
int64_t pq3d_segment_plan_bytes(int64_t N, int64_t S);
int64_t pq3d_segment_ws_bytes(int64_t N, int64_t S, int64_t C);
int pq3d_segment_plan(const int64_t* index, int64_t N, int64_t S, void* plan, int64_t plan_bytes, void* stream);
int pq3d_segment_reduce(const float* src, int64_t Nsrc, const int64_t* gather, const float* row_div, const void* plan,
                        int64_t N, int64_t S, int64_t C, int32_t mean, float* out, float* count, void* ws, int64_t ws_bytes,
                        void* stream);
int pq3d_segment_gather(const float* table, const int64_t* index, const float* count, float* out, int64_t N, int64_t S,
                        int64_t C, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Optimizer side of the training step around the path: what Query3DTrainer.backward does after
 * loss.backward() (trainer/query3d_trainer.py:18-28): clip_grad_norm_(grad_norm) (trainer/build.py:144-145),
 * torch.optim.AdamW.step() with the parameter groups of optim/utils.py:1-18 (weight decay 0.01 except biases /
 * LayerNorm), LambdaLR.step() with optim/scheduler.py:5-17.  All on ONE flat fp32 buffer holding the parameter groups
 * back to back (pq3d_opt_segments: end offset, lr multiplier vs hp.lr -- get_opt_params' per-module lr,
 * query3d_unified.py:224-238 -- and weight decay per group); step-dependent scalars stay in device memory
 * (HIP-graph capturable):
 *   pq3d_sumsq_partials : partials[1024] <- partial sums of g^2 (two-pass, deterministic)
 *   pq3d_train_scalars  : t = ++(*step); lr = hp.lr * lambda(t-1); scalars[0..5] <- lr, lr/(1-b1^t), 1/sqrt(1-b2^t),
 *                         clip = min(1, max_grad_norm/(||g||+1e-6)) (1 if max_grad_norm <= 0), ||g||
 *   pq3d_adamw          : per element of group s: g' = clip*g; p *= 1 - lr*lr_mul_s*wd_s; m += (1-b1)(g'-m);
 *                         v = b2 v + (1-b2) g'^2; p -= lr*lr_mul_s/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 *                         (torch.optim.AdamW, amsgrad=False, maximize=False)
 * ------------------------------------------------------------------------------------------------ */
#define PQ3D_SCHED_CONSTANT 0
#define PQ3D_SCHED_WARMUP_COSINE 1
#define PQ3D_SCHED_WARMUP_EXP 2
#define PQ3D_SUMSQ_PARTIALS 1024
#define PQ3D_MAX_OPT_SEGMENTS 16
typedef struct {
  int32_t n;                                 /* groups in use */
  int64_t end[PQ3D_MAX_OPT_SEGMENTS];        /* exclusive end offset (elements) of group s in the flat buffer */
  float lr_mul[PQ3D_MAX_OPT_SEGMENTS];       /* group lr / hp.lr; < 0: SKIP the segment (p, m, v untouched: a parameter
                                              * without a gradient this step, as torch.optim.AdamW treats grad None) */
  float weight_decay[PQ3D_MAX_OPT_SEGMENTS];
} pq3d_opt_segments;
typedef struct {
  float lr, beta1, beta2, eps;
  float max_grad_norm;   /* <= 0: no clipping */
  int32_t sched;         /* PQ3D_SCHED_* */
  int32_t warmup_steps, total_steps;
  float sched_gamma;     /* warmup_exp only */
  int32_t sched_stride;  /* scheduler steps per optimizer step (<= 0: 1).  The reference prepares its LambdaLR through
                            accelerate (trainer/build.py:123), whose AcceleratedScheduler.step() advances the wrapped
                            scheduler num_processes times per optimizer step: the factor in effect for optimizer step k
                            is lambda(k * num_gpu, warmup * num_gpu, total) -- which is why optim/scheduler.py:20
                            multiplies the warm-up by num_gpu.  Adam's bias correction still counts optimizer steps. */
} pq3d_adamw_hp;
int pq3d_sumsq_partials(const float* g, int64_t n, float* partials, void* stream);
int pq3d_train_scalars(const pq3d_adamw_hp* hp, int64_t* step, const float* partials, float* scalars, void* stream);
int pq3d_adamw(float* p, const float* g, float* m, float* v, int64_t n, const pq3d_adamw_hp* hp,
               const pq3d_opt_segments* segs, const float* scalars, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Instance-segmentation set criterion on the path's outputs (configs/instseg_sceneverse.yaml:160-175):
 * HungarianMatcher cost matrix (modules/third_party/mask3d/matcher.py:104-184, num_points = -1) and the matched
 * losses of SetCriterion (criterion.py:136-206).  Per prediction layer, with X = mask logits [B, Ns, Nq] (segments
 * first), T = padded target masks [B, Nt, Ns] (0/1 floats), seg_len[b] = segments of scene b (matcher.py:139-148 uses
 * the first tgt_mask.shape[1] columns), n_inst[b] = targets of scene b:
 *   pq3d_mask_cost_prep : sig = sigma(X) (0 past seg_len), partial column sums of softplus(X) and sigma(X)
 *                         ([B, nsplit, Nq], nsplit = pq3d_mask_cost_nsplit(Ns))
 *   pq3d_gemm (ct F32, groups 2, batch B, A = T, B = X | sig with transB): TX = T X, TS = T sigma(X)   [B, Nt, Nq]
 *   pq3d_match_cost     : cost_mask = (sum_s softplus(x) - TX)/S  (== batch_sigmoid_ce_loss: pos - neg = -x),
 *                         cost_dice = 1 - (2 TS + 1)/(sum sigma + sum T + 1), cost_class = -softmax(cls)[label]
 *                         (-1 for ignore_label), cost = w_mask cm + w_class cc + w_dice cd     [B, Nq, Nt]
 *   (host)              : scipy.optimize.linear_sum_assignment per scene, as the reference
 *   losses              : sigmoid_ce_loss / dice_loss of a matched pair (criterion.py:27-70) ARE cost_mask / cost_dice at
 *                         that pair -> gathers; pq3d_matched_mask_grad writes d loss / d X (zero off the matched columns);
 *   pq3d_cross_entropy_fwd/bwd : loss_labels (criterion.py:136-163), F.cross_entropy with ignore_index, per-row
 *                         loss + logsumexp; dlogits = scale[layer] * (softmax - onehot) on non-ignored rows.
 * ------------------------------------------------------------------------------------------------ */
/* All entry points below process `layers` (<= PQ3D_MAX_GROUPS) prediction layers of one step in ONE launch: the targets
 * (T, t_sum, seg_len, n_inst, labels) are shared, per-layer inputs arrive as pointer arrays, per-layer intermediates
 * are stacked along a leading `layers` dimension. */
typedef struct {
  int32_t layers, B, Ns, Nq, nsplit;
  const float* X[PQ3D_MAX_GROUPS];   /* mask logits [B, Ns, Nq] of each layer */
  const int32_t* seg_len;            /* [B] */
  float* sig;                        /* [layers, B, Ns, Nq] */
  float* sp_part;                    /* [layers, B, nsplit, Nq] */
  float* sg_part;                    /* [layers, B, nsplit, Nq] */
} pq3d_mask_prep_desc;
typedef struct {
  int32_t layers, B, Nq, Nt, Ns, C, nsplit;
  float w_class, w_mask, w_dice;
  int64_t ignore_label;
  const float* TXS;         /* [layers, 2, B, Nt, Nq]: T X and T sigma(X) (pq3d_gemm outputs) */
  const float* sp_part;     /* [layers, B, nsplit, Nq] */
  const float* sg_part;     /* [layers, B, nsplit, Nq] */
  const float* t_sum;       /* [B, Nt] = sum_s T */
  const int32_t* seg_len;   /* [B] */
  const int32_t* n_inst;    /* [B] */
  const float* cls_logits[PQ3D_MAX_GROUPS];  /* [B, Nq, C] per layer (may hold -inf at filtered classes) */
  const int64_t* labels;    /* [B, Nt] */
  float* cost;              /* [layers, 3, B, Nq, Nt]: total, mask term, dice term; entries t >= n_inst[b] are 0 */
} pq3d_match_cost_desc;
typedef struct {
  int32_t layers, B, Ns, Nq, Nt, Nm;   /* Nm = row length of q_idx / t_idx */
  const float* sig;            /* [layers, B, Ns, Nq] from pq3d_mask_cost_prep */
  const float* T;              /* [B, Nt, Ns] */
  const float* TXS;            /* [layers, 2, B, Nt, Nq] */
  const float* sig_sum;        /* [layers, B, Nq] (sg_part summed over splits) */
  const float* t_sum;          /* [B, Nt] */
  const int32_t* seg_len;      /* [B] */
  const int32_t* q_idx;        /* [layers, B, Nm] matched queries */
  const int32_t* t_idx;        /* [layers, B, Nm] matched targets */
  const int32_t* n_match;      /* [layers, B] */
  const float* g;              /* [layers, 2, B]: upstream gradient x 1/(n_match_b * B) of the BCE term and of the dice
                                  term (the BCE term's 1/S_b is applied inside) */
  float* dX[PQ3D_MAX_GROUPS];  /* [B, Ns, Nq] per layer, fully written */
} pq3d_mask_grad_desc;
typedef struct {
  int32_t layers, C;
  int64_t R, ignore_index;
  const float* logits[PQ3D_MAX_GROUPS];  /* [R, C] per layer */
  const int64_t* target;                 /* [layers, R] */
  float* row_loss;                       /* [layers, R] (forward) */
  float* lse;                            /* [layers, R] written by forward, read by backward */
  const float* scale;                    /* [layers] (backward) */
  float* dlogits[PQ3D_MAX_GROUPS];       /* [R, C] per layer (backward) */
  const float* scale_mul;                /* optional [layers] (backward): a second device-side factor of scale -- the
                                            1 / kept-row count pq3d_cross_entropy_mean leaves, so that neither it nor the
                                            upstream gradient ever visits the host */
} pq3d_ce_desc;
int pq3d_mask_cost_prep(const pq3d_mask_prep_desc* d, void* stream);
int32_t pq3d_mask_cost_nsplit(int32_t Ns);
int pq3d_match_cost(const pq3d_match_cost_desc* d, void* stream);
int pq3d_matched_mask_grad(const pq3d_mask_grad_desc* d, void* stream);
int pq3d_cross_entropy_fwd(const pq3d_ce_desc* d, void* stream);
/* loss[layer] = sum_r row_loss / kept rows (F.cross_entropy reduction='mean' with ignore_index) [+ addend[layer]: the other
 * terms of a total loss, so that "loss = loss + ce" costs no launch], inv_count[layer] = 1 / kept rows; one launch, fixed
 * summation order.  Rows of >= 1024 classes (the caption head's 32128-way LM head,
 * generation_head.py:24-27) take one workgroup per row in both directions. */
int pq3d_cross_entropy_mean(const pq3d_ce_desc* d, float* loss, float* inv_count, const float* addend, void* stream);
int pq3d_cross_entropy_bwd(const pq3d_ce_desc* d, void* stream);

/* Padded mask losses without matching: batch_mask_loss / batch_dice_loss (optim/loss/instseg_loss.py:54-85), used by
 * DirectCriterion (:88-133) and the stage-2 mask_loss (optim/loss/query3d_loss.py:28-39).  X [B, S, N] mask logits
 * (segments first = pred_masks before the reference's permute(0, 2, 1)), T [B, N, S] float targets, P [B, N, S] uint8
 * padding mask (0 = padding pixel / instance).
 *   pq3d_padded_mask_sums : part[B, ceil(S/64), N, 4] partial sums over 64-segment tiles of
 *                           (bce(x,t) p, p, sigma(x) t p, (sigma(x) + t) p); summing over tiles gives sums[B, N, 4]
 *   pq3d_padded_mask_grad : dX = p (gm[b,n] (sigma - t) + gd[b,n] sigma (1 - sigma) d dice_score-term), with gm / gd the
 *                           upstream gradients already divided by the caller's normalisers. */
int pq3d_padded_mask_sums(const float* X, const float* T, const uint8_t* P, float* part, int32_t B, int32_t S, int32_t N,
                          void* stream);
int pq3d_padded_mask_grad(const float* X, const float* T, const uint8_t* P, const float* sums, const float* gm,
                          const float* gd, float* dX, int32_t B, int32_t S, int32_t N, void* stream);

/* Ragged -> padded collate on the device: pad_sequence / pad_sequence_2d (data/data_utils.py:337-382) as
 * used by InstSegDatasetWrapper.collate_fn (data/datasets/instseg_wrapper.py:27-81) for segment features, centres, labels
 * and target masks.  The ragged batch is packed: sample b occupies rows offsets[b] .. offsets[b+1] of src [total, D]
 * (1-D) or h[b] x w[b] x D elements starting at element offset offsets[b] (2-D).  out = [B, L, D] / [B, H, W, D] filled
 * with the element at *pad_value (host pointer, elem_size bytes) outside the samples; mask (optional) = 1 where padded
 * ("True as masked").  Elements are copied as opaque 1/2/4/8-byte units (any dtype). */
int pq3d_pad_sequence(const void* src, const int64_t* offsets, void* out, uint8_t* mask, int32_t B, int64_t L, int64_t D,
                      int32_t elem_size, const void* pad_value, void* stream);
int pq3d_pad_sequence_2d(const void* src, const int64_t* offsets, const int32_t* heights, const int32_t* widths, void* out,
                         uint8_t* mask, int32_t B, int64_t H, int64_t W, int64_t D, int32_t elem_size, const void* pad_value,
                         void* stream);

/* ------------------------------------------------------------------------------------------------
 * PointNet++ point-set operators (the reference's only native code:
 * modules/third_party/pointnet2/_ext_src/src, bound in bindings.cpp:6-19 and wrapped by pointnet2_utils.py).  fp32
 * coordinates / features, int32 indices, tensors laid out as those wrappers pass them.
 *   pq3d_furthest_point_sampling : xyz [B,N,3] -> idx [B,M]; starts at point 0, skips points with |p|^2 <= 1e-3, running
 *                                  min squared distance initialised to 1e10 (sampling_gpu.cu:70-172, sampling.cpp);
 *                                  ties go to the smaller index; N <= 8192
 *   pq3d_ball_query              : first `nsample` point indices (in index order) with squared distance < radius^2 from
 *                                  each centre new_xyz [B,M,3]; unfilled slots repeat the first hit
 *                                  (ball_query_gpu.cu:9-44) -> idx [B,M,nsample]
 *   pq3d_gather_points(_grad)    : out[b,c,i] = points[b,c,idx[b,i]], idx [B,L]: gather_points (L = m,
 *                                  sampling_gpu.cu:8-20) and group_points (L = npoint*nsample, group_points_gpu.cu:8-30);
 *                                  the gradient scatter-adds into a zeroed [B,C,N]
 *   pq3d_three_nn                : 3 nearest known [B,M,3] of every unknown [B,N,3]: SQUARED distances + indices
 *                                  (interpolate_gpu.cu:9-58; the Python wrapper takes the sqrt)
 *   pq3d_three_interpolate(_grad): out[b,c,i] = sum_k points[b,c,idx[b,i,k]] weight[b,i,k] (interpolate_gpu.cu:72-143)
 * ------------------------------------------------------------------------------------------------ */
int pq3d_furthest_point_sampling(const float* xyz, int32_t* idx, int32_t B, int32_t N, int32_t M, void* stream);
int pq3d_ball_query(const float* new_xyz, const float* xyz, int32_t* idx, int32_t B, int32_t N, int32_t M, float radius,
                    int32_t nsample, void* stream);
int pq3d_gather_points(const float* points, const int32_t* idx, float* out, int32_t B, int32_t C, int32_t N, int64_t L,
                       void* stream);
int pq3d_gather_points_grad(const float* grad_out, const int32_t* idx, float* grad_points, int32_t B, int32_t C, int32_t N,
                            int64_t L, void* stream);
int pq3d_three_nn(const float* unknown, const float* known, float* dist2, int32_t* idx, int32_t B, int32_t N, int32_t M,
                  void* stream);
int pq3d_three_interpolate(const float* points, const int32_t* idx, const float* weight, float* out, int32_t B, int32_t C,
                           int32_t M, int32_t N, void* stream);
int pq3d_three_interpolate_grad(const float* grad_out, const int32_t* idx, const float* weight, float* grad_points, int32_t B,
                                int32_t C, int32_t M, int32_t N, void* stream);
/* Row form of the set-abstraction grouping for the frozen PointNet++ backbone (modules/layers/pointnet.py:22-63,
 * pointnet2_modules.py:23-70; QueryAndGroup / GroupAll of pointnet2_utils.py:291-419 with use_xyz): the SharedMLP's 1x1
 * convolutions become row GEMMs (pq3d_gemm), so grouping writes channels-last rows.
 *   pq3d_group_rows    : out[(b,p,s), :] = [xyz[b,j] - new_xyz[b,p] (3), feats[b,j,0:C], 0 ... (to Kp)], j = idx[b,p,s];
 *                        new_xyz == NULL: no centring; idx == NULL (GroupAll): j = s, ns == N.  xyz [B,N,3] fp32,
 *                        feats rows [B,N,feat_stride] (dt_f), out [B*np*ns, Kp] (dt_o)
 *   pq3d_group_maxpool : out[g, c] = max_s rows[g, s, c]  (F.max_pool2d over the samples, pointnet2_modules.py:62-65)
 */
int pq3d_group_rows(const float* xyz, const float* new_xyz, const void* feats, int32_t dt_f, int64_t feat_stride,
                    const int32_t* idx, void* out, int32_t dt_o, int32_t B, int32_t N, int32_t C, int32_t np, int32_t ns,
                    int32_t Kp, void* stream);
int pq3d_group_maxpool(const void* rows, void* out, int32_t dt, int64_t G, int32_t ns, int32_t C, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Pieces of the generation head's decoder body (third-party arithmetic: HF transformers
 * T5ForConditionalGeneration as driven by modules/heads/generation_head.py:20-30) that the kernels above do not cover:
 *   pq3d_rmsnorm_fwd/bwd  : T5LayerNorm  y = x rsqrt(mean(x^2) + eps) w  (saves rstd [R]); backward writes dx and adds
 *                           (accumulate != 0) or stores the weight gradient
 *   pq3d_embedding_fwd / _bwd_acc : token embedding rows and their scatter-add gradient (dtable is accumulated into)
 * Everything else of the body is pq3d_gemm (bias-free projections, LM head), pq3d_attn_fwd/bwd (scale 1, additive
 * relative-position + causal bias, key padding) and pq3d_cross_entropy_*.
 * ------------------------------------------------------------------------------------------------ */
int pq3d_rmsnorm_fwd(const float* x, const float* w, float* y, float* rstd, int64_t R, int32_t d, float eps, void* stream);
int pq3d_rmsnorm_bwd(const float* x, const float* w, const float* rstd, const float* dy, float* dx, float* dw, int64_t R,
                     int32_t d, int32_t accumulate, void* stream);
/* the same with the gradient of the sublayer's residual branch added to dx (x feeds the norm AND the residual add of a
 * pre-norm sublayer, modeling_t5.py T5LayerSelfAttention / CrossAttention / FF: hidden + dropout(f(norm(hidden)))) -- the
 * two-way gradient junction without an add launch; dres may be NULL */
int pq3d_rmsnorm_bwd_res(const float* x, const float* w, const float* rstd, const float* dy, const float* dres, float* dx,
                         float* dw, int64_t R, int32_t d, int32_t accumulate, void* stream);
/* ... and with a second output dxm = dropout_mask(drop) * dx / (1 - p) over the [R, d] site `drop`: the gradient the
 * PRECEDING sublayer's output projection needs (hidden = residual + dropout(o W^T)), written by this kernel instead of by a
 * dropout launch of its own (18 launches per caption-body backward at config 5).  drop / dxm may be NULL. */
int pq3d_rmsnorm_bwd_res_drop(const float* x, const float* w, const float* rstd, const float* dy, const float* dres, float* dx,
                              float* dw, int64_t R, int32_t d, int32_t accumulate, const pq3d_dropout* drop, float* dxm,
                              void* stream);
int pq3d_embedding_fwd(const float* table, const int64_t* ids, float* out, int64_t R, int32_t d, void* stream);
int pq3d_embedding_bwd_acc(const float* dout, const int64_t* ids, float* dtable, int64_t R, int32_t d, void* stream);
/* token embedding with the embedding dropout of the T5 stack fused in (out = dropout(table[ids]) over the [R, d] site), and
 * its gradient (dtable += scatter of dropout(dout)) */
int pq3d_embedding_drop_fwd(const float* table, const int64_t* ids, float* out, int64_t R, int32_t d, const pq3d_dropout* dr,
                            void* stream);
int pq3d_embedding_drop_bwd_acc(const float* dout, const int64_t* ids, float* dtable, int64_t R, int32_t d,
                                const pq3d_dropout* dr, void* stream);
/* One launch for what the HF model does with framework elementwise ops before its first layer (pq3d_amd/csrc/t5glue.hip):
 * ids [B,T] = shift_right(labels) ([start, labels[:-1]], -100 -> pad); bias [B,H,T,T] = rel[buckets[q,k], h] with -inf for
 * k > q (relative-position bias + causal mask, rel [NB,H] = relative_attention_bias.weight, buckets [T,T] int64); kpm [B,N]
 * = !enc_valid (NULL: skipped).  pq3d_t5_bias_bwd: drel[nb,h] (+)= sum_b sum_{(q,k) causal, bucket nb} dbias[b,h,q,k]. */
int pq3d_t5_prep(const int64_t* labels, int64_t start_id, int64_t pad_id, const float* rel, const int64_t* buckets,
                 const uint8_t* enc_valid, int64_t* ids, float* bias, uint8_t* kpm, int32_t B, int32_t T, int32_t H, int64_t N,
                 void* stream);
int pq3d_t5_bias_bwd(const float* dbias, const int64_t* buckets, float* drel, int32_t B, int32_t T, int32_t H, int32_t NB,
                     int32_t accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Data-parallel gradient exchange over RCCL / xGMI (SURVEY 8b's export list: pq3d_comm_init, pq3d_allreduce_grads; 8e).
 * Replaces, for a host that does not go through torch.distributed: DistributedDataParallel's bucketed all-reduce(mean) of
 * the parameter gradients every step (reference trainer/build.py:66-75 via accelerate; the only collective on the path).
 * One communicator per process (= per GPU).  librccl is bound at the first call (dlopen; a copy the process already loaded --
 * torch's -- is reused), so the library itself has no link-time dependency on it; a missing librccl is an error text, not a
 * load failure.  Everything is stream-ordered: no host synchronisation, capturable where RCCL's collectives are.
 *   pq3d_comm_unique_id : rank 0 makes the 128-byte rendezvous id (ncclGetUniqueId); the host carries it to the other ranks
 *   pq3d_comm_init      : collective over all ranks (ncclCommInitRank on the CURRENT device); *comm = opaque handle
 *   pq3d_allreduce_grads: in-place sum (mean != 0: mean, inside the collective -- ncclAvg) of count elements, dtype PQ3D_F32
 *                         or PQ3D_BF16 (a bf16 buffer is summed in bf16 by the ring: use the wire form below instead)
 *   pq3d_allreduce_grads_wire : fp32 gradients, bf16 on the links, FP32 ACCUMULATION: cast -> all-to-all (rank r receives
 *                         piece r of every rank) -> fp32 sum in rank order (/ world) -> bf16 shard -> all-gather -> fp32.  Half the
 *                         bytes of the fp32 all-reduce per link, two roundings per element, bit-identical results on every rank.
 *                         scratch: pq3d_allreduce_wire_scratch_bytes(world, count) bytes of device memory, 16-byte aligned.
 * ------------------------------------------------------------------------------------------------ */
#define PQ3D_COMM_ID_BYTES 128
int pq3d_comm_unique_id(void* id);
int pq3d_comm_init(int32_t rank, int32_t world, const void* unique_id, void** comm);
int pq3d_comm_destroy(void* comm);
int pq3d_comm_info(void* comm, int32_t* rank, int32_t* world, int32_t* rccl_version);
int pq3d_allreduce_grads(void* comm, void* grads, int64_t count, int32_t dtype, int32_t mean, void* stream);
int64_t pq3d_allreduce_wire_scratch_bytes(int32_t world, int64_t count);
int pq3d_allreduce_grads_wire(void* comm, float* grads, int64_t count, void* scratch, int64_t scratch_bytes, int32_t mean,
                              void* stream);
/* test hook: the wire form's reduce stream alone -- shard[i] = bf16((sum_r float(recv[r * per + i])) / (mean ? world : 1)), bf16 in / out,
 * per % 8 == 0 -- so that the W > 1 arithmetic is checked on a one-GPU box (tests/test_gpu_comm.py) */
int pq3d_test_wire_reduce(const void* recv, void* shard, int64_t per, int32_t world, int32_t mean, void* stream);

#ifdef __cplusplus
}
#endif
#endif
