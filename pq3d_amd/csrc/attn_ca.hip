// Small cross-attention on bf16 operands: FEW queries AND few keys per (scene, head) -- the shipped stage-2 decoder's
// cross-attention over <= 80 objects per memory (unified_tasks_sceneverse.yaml: 128 scenes x 3 memories x 12 heads of 64 =
// 4608 (scene, head) problems of 80 x 80 scores), its prompt tokens (80 x 32), the caption decoder's attention to the
// query tokens.  The general kernels of attention.hip walk such a call as 64-key tiles through their two-stage pipeline
// and form S / dP in two kernels (342 us backward at the stage-2 shape: 0.96 TB/s, 44 TFLOP/s); the all-queries-resident
// backward of attn_resident.hip still spends a workgroup's prologue and dQ reduction on 2.5 key groups (262 us).  Here
// one workgroup owns a (scene, head) outright, as in attn_sa.hip: Q, K, V (and dO) go into LDS once as bf16 planes
// (a straight copy: the operands ARE bf16), one wave per 16-row block of queries / keys, no staging pipeline, no barrier
// after the first, no atomics, two workgroups per CU:
//   forward : S^T = K Q^T per pair of 16-key tiles (lane = query column), online softmax over the pairs starting from the
//             zero key of add_zero_attn (m = 0, l = 1), O^T += V^T P^T with P^T straight from the softmax registers;
//   backward: phase A (wave = query block) S^T, P = exp(S - lse), dP^T - delta (accumulators start at -delta), dS,
//             dQ^T += K^T dS^T; phase B (wave = key block) the same tiles in the S orientation for dV^T += dO^T P and
//             dK^T += Q^T dS.  S and dP are formed twice (a few dozen MFMAs per wave); every output has one writer.
// Same rounding points as the general bf16 kernels (P and dS rounded to bf16 for the second products, fp32 accumulate,
// fp32 softmax); key padding through an additive 0 / -inf row; attention dropout from the counter-based generator (the
// masks of the general kernels: same site rows / columns); 3-D masks, additive bias and key splits stay on the general kernels.  The launch is bound by HBM: 82 KB of operands and results per (scene, head) at 80 x 80 x 64.
#include <atomic>
#include <cstdlib>

#include "attn_common.h"
#include "gemm_common.h"

namespace {

typedef unsigned u32pair_c __attribute__((ext_vector_type(2)));

namespace ca32 {
#define CA_DH 32
#include "attn_ca_body.h"
#undef CA_DH
}  // namespace ca32
namespace ca64 {
#define CA_DH 64
#include "attn_ca_body.h"
#undef CA_DH
}  // namespace ca64

bool ca_al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

// bf16 attention with at most 128 queries and 128 keys per (scene, head), d_h = 32 / 64, key-padding mask only.  Returns
// false when the call is not of that shape (the general / resident kernels run instead).
bool pq3d_attn_ca_try(const pq3d_attn_desc& d, hipStream_t s, bool bwd) {
  if (d.ct != PQ3D_BF16 || d.dt != PQ3D_BF16 || (d.dh != 32 && d.dh != 64)) return false;
  if (d.mask || d.bias || d.dbias || d.ksplit > 1 || d.proj.mode != PQ3D_ATTN_PROJ_NONE) return false;
  const bool dr = d.drop.p > 0.f && d.drop.seed;
  if (d.Lq < 1 || d.Lk < 1 || d.Lq > 128 || d.Lk > 128) return false;
  // 16-byte row accesses: strides and bases of q / k / v / o (/ their gradients)
  if ((d.q_sl | d.k_sl | d.v_sl | d.o_sl | d.q_sb | d.k_sb | d.v_sb | d.o_sb | d.q_sh | d.k_sh | d.v_sh | d.o_sh) & 7) return false;
  if (!ca_al16(d.q) || !ca_al16(d.k) || !ca_al16(d.v) || !ca_al16(d.o)) return false;
  if (bwd && !(d.dout && d.dq && d.dk && d.dv && d.delta && ca_al16(d.dout) && ca_al16(d.dq) && ca_al16(d.dk) && ca_al16(d.dv))) return false;
  const size_t lds = d.dh == 32 ? ca32::ca_lds_bytes(d.Lq, d.Lk, bwd) : ca64::ca_lds_bytes(d.Lq, d.Lk, bwd);
  const int blocks = (max(d.Lq, d.Lk) + 15) / 16;
#define CA_LAUNCH(KERN)                                                                              \
  do {                                                                                               \
    static std::atomic<unsigned> done{0};                                                            \
    if (pq3d_enable_big_lds(KERN, 160 * 1024, done)) { (void)hipGetLastError(); return false; }      \
    hipLaunchKernelGGL(KERN, dim3(d.H, d.B), dim3(blocks * 64), lds, s, d);                          \
  } while (0)
  if (d.dh == 32) {
    if (bwd) { if (dr) CA_LAUNCH(ca32::attn_ca_bwd_kernel<true>); else CA_LAUNCH(ca32::attn_ca_bwd_kernel<false>); }
    else { if (dr) CA_LAUNCH(ca32::attn_ca_fwd_kernel<true>); else CA_LAUNCH(ca32::attn_ca_fwd_kernel<false>); }
  } else {
    if (bwd) { if (dr) CA_LAUNCH(ca64::attn_ca_bwd_kernel<true>); else CA_LAUNCH(ca64::attn_ca_bwd_kernel<false>); }
    else { if (dr) CA_LAUNCH(ca64::attn_ca_fwd_kernel<true>); else CA_LAUNCH(ca64::attn_ca_fwd_kernel<false>); }
  }
#undef CA_LAUNCH
  return true;
}
