// Self-attention of the decoder on the matrix cores at fp32 grade: fp32 q / k / v / o in HBM, SPLIT-bf16 MFMA products
// (every fp32 operand x staged as hi = bf16(x), lo = bf16(x - hi); a product is lo*hi + hi*lo + hi*hi, fp32 accumulate --
// the arithmetic of PQ3D_BF16X3 in gemm.hip: ~2^-17 relative per term), additive fp32 bias [B,H,Lq,Lk] (spatial
// self-attention: log(clamp(loc)) term, transformers.py:229-237) and key padding.  Replaces the vector-ALU kernels of
// attn_small.hip for d_h = 32 and -- round 4 -- d_h = 64, the head width of the reference's SHIPPED decoders (hidden 768, 12 heads:
// configs/instseg_sceneverse.yaml:95,141; at most 128 queries / keys there: the planes of a head are 72 KB + 72 KB of LDS;
// config s2: self-attention backward 830 -> us on the vector-ALU kernels before) (12 / 26 us forward / backward at N_q = 100 for 1.3 / 3.2 MFLOP per head: the FMA chains
// and LDS sweeps of 64 x 4 workgroups) and the general streaming fp32 kernels for 128 < N_q <= 240 (config 4: 31 + 63 us).
//
// One workgroup per (scene, head), one wave per 16-row block of queries / keys, everything resident:
//   * Q, K, V (and dO) of the head are converted once into hi / lo bf16 planes in LDS, row-major [token][d_h + 8];
//   * scores are formed TRANSPOSED per 16-key tile, S^T = K_tile Q_blk^T, so that the MFMA C layout (lane = query column,
//     4 consecutive keys per lane) IS the B-operand layout of the second product: P^T goes from the softmax registers
//     straight into O^T = V^T P^T (two key tiles = one 32-wide k step; the matching A fragments of V^T come from the
//     row-major V plane through the transposing LDS read at the same permuted key order) -- no P tile in LDS, no barrier
//     inside the key loop; online softmax over key-tile pairs;
//   * backward, no atomics: phase A (wave = query block) recomputes S^T, P = exp(S - lse), dP^T = V dO^T,
//     dS = P (dP - delta), writes dbias = dS and accumulates dQ^T = K^T dS^T; phase B (wave = key block) recomputes the
//     same tiles in the S orientation (lane = key column) for dV^T = dO^T P and dK^T = Q^T dS.  Every output element has
//     one writer; S and dP are formed twice (a few hundred MFMAs per wave).  Round 6: with few (scene, head) units the two phases
//     (and, below a quarter of a chip, two halves of each phase's 16-row blocks) run on separate workgroups that stage the
//     same planes: grid.z = 2 / 4, the same bits.
// zero_attn, 3-D masks and attention dropout stay on the general kernels.
#include <atomic>
#include <cstdlib>

#include "attn_common.h"
#include "gemm_common.h"

namespace {

namespace sa32 {   // d_h = 32: up to 240 queries / keys, 15 waves, optional folded out-projection backward
#define SA_DH 32
#define SA_MAXT 1024
#include "attn_sa_body.h"
#undef SA_DH
#undef SA_MAXT
}  // namespace sa32
namespace sa64 {   // d_h = 64: two k steps per product over d_h, four 16-row tiles per [d_h][tokens] result, up to 128 queries / keys
#define SA_DH 64
#define SA_MAXT 512
#include "attn_sa_body.h"
#undef SA_DH
#undef SA_MAXT
}  // namespace sa64

bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

// fp32 self-attention at d_h = 32 (at most 240 queries and keys) or d_h = 64 (at most 128) on the split-bf16 MFMA kernels:
// returns false when the call is not of that shape (the vector-ALU kernels of attn_small.hip / the general kernels run instead).
bool pq3d_attn_sa_try(const pq3d_attn_desc& d, hipStream_t s, bool bwd) {
  if (d.ct != PQ3D_F32 || d.dt != PQ3D_F32 || (d.dh != 32 && d.dh != 64) || d.mask || d.zero_attn || (d.drop.p > 0.f && d.drop.seed) || d.ksplit > 1)
    return false;
  const int lmax = d.dh == 32 ? 240 : 128;
  if (d.Lq < 1 || d.Lk < 1 || d.Lq > lmax || d.Lk > lmax) return false;
  // 16-byte row accesses: strides and bases of q / k / v / o (/ their gradients)
  if ((d.q_sl | d.k_sl | d.v_sl | d.o_sl | d.q_sb | d.k_sb | d.v_sb | d.o_sb | d.q_sh | d.k_sh | d.v_sh | d.o_sh) & 3) return false;
  if (!al16(d.q) || !al16(d.k) || !al16(d.v) || !al16(d.o)) return false;
  const bool fold = bwd && d.proj.mode == PQ3D_ATTN_PROJ_DOUT;
  if (bwd && !((d.dout || fold) && d.dq && d.dk && d.dv && d.delta && al16(d.dout) && al16(d.dq) && al16(d.dk) && al16(d.dv))) return false;
  if (fold && (d.dh != 32 || d.proj.dm != d.H * 32 || (d.o_sl & 3))) return false;
  const int lqp = bwd ? ((d.Lq + 31) & ~31) : d.Lq;
  size_t lds = d.dh == 32 ? sa32::sa_lds_bytes(lqp, d.Lk, bwd, fold ? d.proj.dm : 0) : sa64::sa_lds_bytes(lqp, d.Lk, bwd, 0);
  if (lds > 160 * 1024) return false;
  const int blocks = (max(d.Lq, d.Lk) + 15) / 16;
  // backward with at most half a chip of (scene, head) units: phase A and phase B on separate workgroups, and with at most a
  // quarter of a chip each phase on two (attn_sa_body.h; PQ3D_SA_BWD_SPLIT=0 / 1 / 2 for an A/B).  Same-box, rocprofv3, per launch:
  // config 2 (64 units, 100 queries) 22.4 -> 17.9 -> 16.8 us, config 4 (32 units, 200 queries) 41.2 -> 25.9 -> 22.2 us
  static const int split = [] { const char* e = getenv("PQ3D_SA_BWD_SPLIT"); return e ? atoi(e) : 2; }();
  // (the forward gains nothing from the same split: 9.4 -> 9.0 us at config 2, 15.7 -> 15.5 at config 4 -- its time is the staging)
  const int parts = (bwd && split > 0 && (long)d.H * d.B <= 128) ? ((split > 1 && (long)d.H * d.B <= 64 && blocks >= 4) ? 4 : 2) : 1;
#define SA_LAUNCH(KERN)                                                                              \
  do {                                                                                               \
    static std::atomic<unsigned> done{0};                                                            \
    if (pq3d_enable_big_lds(KERN, 160 * 1024, done)) { (void)hipGetLastError(); return false; }      \
    hipLaunchKernelGGL(KERN, dim3(d.H, d.B, parts), dim3(blocks * 64), lds, s, d);                   \
  } while (0)
  if (d.dh == 32) { if (bwd) SA_LAUNCH(sa32::attn_sa_bwd_kernel); else SA_LAUNCH(sa32::attn_sa_fwd_kernel); }
  else { if (bwd) SA_LAUNCH(sa64::attn_sa_bwd_kernel); else SA_LAUNCH(sa64::attn_sa_fwd_kernel); }
#undef SA_LAUNCH
  return true;
}
