// Shared device helpers of the attention kernels (attention.hip: streaming forward / two-kernel recompute backward;
// attn_resident.hip: all-queries-resident single-pass backward).  Tile conventions: see attention.hip's header.
#pragma once
#include "common.h"

namespace {

// ---- XCD-aware workgroup coordinates of a (X, H, B) attention grid ---------------------------------------------------------
// Heads are packed in the last dimension of Q / K / V (h * d_h + c): at d_h = 32 a bf16 key row of one head
// is a 64-byte slice at a 512-byte stride, i.e. HALF of a 128-byte line whose other half belongs to the neighbouring head; a
// 3-D self-mask tile is read by all H heads of its scene.  The dispatcher deals workgroups to the 8 XCDs round-robin in
// linear-id order (x fastest, then y = head): with the plain blockIdx mapping the H heads of a scene run on H DIFFERENT
// XCDs (id = x + X * (h + H * b): for X = 1, H = 8 head h is always on XCD h), each with its own L2, and every K / V line is
// fetched from memory twice (measured, rocprofv3 FETCH_SIZE calibrated: 53 MB per forward launch at config 2 for 25 MB of
// K + V, 212 MB for 101 MB at config 5), every mask tile 8 times.
// Here the workgroups ONE XCD receives in consecutive dispatch rounds (ids i, i + 8, i + 16, ...) are the members of ONE
// group = the workgroups that read the same K / V lines: every head -- and every query chunk -- of one (key slice, scene).
// They run at the same time on the same L2 and stream the same lines in step, so all but the first reader of a line hit (or
// merge with its miss).  `xinner` = how many DISTINCT key slices the x dimension enumerates in its low part (x = xo * xinner
// + xi: the streaming kernels' x = qchunk * KS + split -> xinner = KS; the resident kernels' x = split and the dK / dV kernel's
// x = key chunk -> xinner = gridDim.x).  The last partial block of 8 groups keeps the plain logical order.  Pure placement:
// every workgroup still owns exactly one (x, h, b); results do not depend on it.
#ifndef PQ3D_ATTN_XCD
#define PQ3D_ATTN_XCD 1
#endif
struct WgXyz { int x, h, b; };
PQ_DEV WgXyz attn_wg_xyz(int xinner) {
#if PQ3D_ATTN_XCD
  const unsigned X = gridDim.x, H = gridDim.y, B = gridDim.z, XI = (unsigned)xinner, XO = X / XI;
  const unsigned id = blockIdx.x + X * (blockIdx.y + H * blockIdx.z);
  const unsigned G = XO * H;                       // members of a group: (xo, h)
  const unsigned blk = 8u * G, T = (X * H * B / blk) * blk;
  unsigned unit, m;                                // unit = (xi, b), xi fastest
  if (id < T) {
    const unsigned xcd = id & 7u, j = id >> 3;
    unit = (j / G) * 8u + xcd;
    m = j % G;
  } else {                                         // tail: members fastest, then units
    unit = id / G;
    m = id % G;
  }
  return WgXyz{(int)((m / H) * XI + unit % XI), (int)(m % H), (int)(unit / XI)};
#else
  (void)xinner;
  return WgXyz{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
#endif
}

constexpr int KB = 64;  // keys per main-loop iteration (fwd, dQ)
constexpr int QB = 32;  // queries per main-loop iteration (dK/dV)
// dK/dV: 16 keys per wave; 4 waves (64 keys) per workgroup for short key sequences, 8 (128 keys) for long ones

template <typename CT, int DH> struct AT {
  static constexpr int EPL = Mma<CT>::EPL, KSTEP = Mma<CT>::KSTEP;
  static constexpr int DHK = ((DH + KSTEP - 1) / KSTEP) * KSTEP;  // dh rounded up to MFMA k-steps
  static constexpr int NS = DHK / KSTEP;                          // k-steps over dh
  static constexpr int PADE = 16 / (int)sizeof(CT);
  static constexpr int LDR = DHK + PADE;   // row stride of a row-major [rows][dh] tile
  static constexpr int LDT = KB + PADE;    // row stride of a transposed [dh][64] tile
  static constexpr int LDQ = QB + PADE;    // row stride of a transposed [dh][32] tile
  static constexpr int MT = DH / 16;       // 16-row tiles over dh
  static constexpr int CPR = DH / EPL;     // 16-byte chunks per row
};

template <typename CT> PQ_DEV float fexp(float x);
template <> PQ_DEV float fexp<float>(float x) { return expf(x); }
template <> PQ_DEV float fexp<bf16_t>(float x) { return __expf(x); }

// A/B fragment from a row-major tile row (k contiguous)
template <typename CT> PQ_DEV u32x4 rfrag(const CT* row, int step, int g) {
  return *(const u32x4*)&row[step * Mma<CT>::KSTEP + g * Mma<CT>::EPL];
}
// A fragment from a transposed tile row whose k index enumerates the columns of C-layout tiles:
// bf16 step = two 16-wide tiles -> k slots {4g..4g+3} of tile 2*step and of tile 2*step+1; f32 step = one tile.
template <typename CT> PQ_DEV u32x4 tfrag(const CT* row, int step, int g);
template <> PQ_DEV u32x4 tfrag<bf16_t>(const bf16_t* row, int step, int g) {
  const u32x2 lo = *(const u32x2*)&row[step * 32 + 4 * g];
  const u32x2 hi = *(const u32x2*)&row[step * 32 + 16 + 4 * g];
  return (u32x4){lo.x, lo.y, hi.x, hi.y};
}
template <> PQ_DEV u32x4 tfrag<float>(const float* row, int step, int g) {
  return *(const u32x4*)&row[step * 16 + 4 * g];
}
// bf16 only: the same fragment straight from a ROW-MAJOR tile (row = k index, e.g. key; column = the 16-wide output
// chunk col0) through the gfx950 hardware transposing LDS read.  Measured semantics (tools/probes/tr_probe.hip):
// a lane (i, g) pointing at row base + i/4, columns 4*(i%4).. receives tile[base + j][i], j = 0..3.  Two reads
// give the 8 k-slots {4g+j} and {16+4g+j} of step `row0/32`, so transposed LDS copies (2-byte scattered stores
// with 8-way bank conflicts) are not needed at all.
typedef short v4i16_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16_t lds_v4i16_t;
PQ_DEV u32x4 tfrag_tr(const bf16_t* tile, int ldr, int row0, int col0, int li, int lg) {
  const bf16_t* p0 = tile + (row0 + 4 * lg + (li >> 2)) * ldr + col0 + 4 * (li & 3);
  const v4i16_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_t*)p0);
  const v4i16_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16_t*)(p0 + 16 * ldr));
  const u32x2 lo = __builtin_bit_cast(u32x2, a), hi = __builtin_bit_cast(u32x2, b);
  return (u32x4){lo.x, lo.y, hi.x, hi.y};
}
// A fragment of the transposed operand for step u, output chunk mt: bf16 -> tr read of the row-major tile `rm`
// (row stride ldr); f32 -> plain read of the transposed copy `tr` (row stride ldt).
template <typename CT> PQ_DEV u32x4 tfrag_any(const CT* rm, int ldr, const CT* tr, int ldt, int u, int mt, int li, int lg) {
  if constexpr (sizeof(CT) == 2) return tfrag_tr((const bf16_t*)rm, ldr, u * 32, mt * 16, li, lg);
  else return tfrag<CT>(&tr[(mt * 16 + li) * ldt], u, lg);
}

// C-layout tiles (lane: rows 4g+r of tile t, column i) -> B fragments whose k index = tile rows.
template <typename CT, int NTILES> struct PackP;
template <int NTILES> struct PackP<bf16_t, NTILES> {
  static constexpr int STEPS = NTILES / 2;
  static PQ_DEV void run(const float (&p)[NTILES][4], u32x4 (&out)[STEPS]) {
#pragma unroll
    for (int u = 0; u < STEPS; ++u)
      out[u] = (u32x4){pack_bf2(p[2 * u][0], p[2 * u][1]), pack_bf2(p[2 * u][2], p[2 * u][3]),
                       pack_bf2(p[2 * u + 1][0], p[2 * u + 1][1]), pack_bf2(p[2 * u + 1][2], p[2 * u + 1][3])};
  }
};
template <int NTILES> struct PackP<float, NTILES> {
  static constexpr int STEPS = NTILES;
  static PQ_DEV void run(const float (&p)[NTILES][4], u32x4 (&out)[STEPS]) {
#pragma unroll
    for (int t = 0; t < NTILES; ++t)
      out[t] = (u32x4){__float_as_uint(p[t][0]), __float_as_uint(p[t][1]), __float_as_uint(p[t][2]),
                       __float_as_uint(p[t][3])};
  }
};

// Cooperative global -> register -> LDS staging of ROWS x DH elements of type CT (rows r0.., element stride sl
// between rows).  load() issues branch-free 16-byte loads (row index clamped to R-1: out-of-range rows hold finite
// duplicates and are neutralised by the masks); store() writes a row-major copy (stride LDR) and/or a transposed
// copy (stride ldt).  Splitting the two lets the next tile's loads fly during the current tile's MFMAs.
template <typename CT, int DH, int ROWS, int NTHREADS>
struct TileRegs {
  typedef AT<CT, DH> A;
  static constexpr int TOTAL = ROWS * A::CPR;
  static constexpr int MAXC = (TOTAL + NTHREADS - 1) / NTHREADS;
  u32x4 reg[MAXC];
  PQ_DEV void load(const void* base, long off, long sl, int r0, int R, int tid) {
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = tid + i * NTHREADS;
      if (c < TOTAL) {
        const int row = c / A::CPR, kc = c % A::CPR;
        const int gr = min(r0 + row, R - 1);
        reg[i] = *(const u32x4*)((const CT*)base + off + (long)gr * sl + kc * A::EPL);
      }
    }
  }
  PQ_DEV void store(CT* rm, CT* tr, int ldt, int tid) const {
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = tid + i * NTHREADS;
      if (c < TOTAL) {
        const int row = c / A::CPR, kc = c % A::CPR;
        if (rm) *(u32x4*)&rm[row * A::LDR + kc * A::EPL] = reg[i];
        if (tr) {
#pragma unroll
          for (int j = 0; j < A::EPL; ++j) {
            if constexpr (sizeof(CT) == 2)
              tr[(kc * A::EPL + j) * ldt + row] = (CT)((reg[i][j >> 1] >> (16 * (j & 1))) & 0xffffu);
            else
              tr[(kc * A::EPL + j) * ldt + row] = __uint_as_float(reg[i][j]);
          }
        }
      }
    }
  }
};

// B-operand fragments of one row (lane i = row index, g = dh slice), straight from global (typed 16-byte loads).
template <typename CT, int DH>
PQ_DEV void row_frags(u32x4* f, const void* base, long rowoff, int g) {
  typedef AT<CT, DH> A;
#pragma unroll
  for (int s = 0; s < A::NS; ++s) {
    const int c0 = s * A::KSTEP + g * A::EPL;
    f[s] = (c0 < DH) ? *(const u32x4*)((const CT*)base + rowoff + c0) : (u32x4){0, 0, 0, 0};
  }
}

template <typename CT, int DH> PQ_DEV void zero_lds(CT* p, int n, int tid, int nthreads) {
  for (int i = tid; i < n; i += nthreads) p[i] = Cvt<CT>::from(0.f);
}

// Cross-lane reductions over the 4 lane groups (lanes l, l^16, l^32, l^48) with the gfx950 VALU lane swaps instead of
// LDS-crossbar shuffles: permlaneN_swap(x, x) leaves {x[l], x[l^N]} in the two results on every lane (measured,
// tools/probes/permlane_probe.hip), so one swap + one max/add is a complete xor-N step.
typedef unsigned u32pair __attribute__((ext_vector_type(2)));
PQ_DEV float group_max(float v) {
  u32pair a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  u32pair b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
PQ_DEV float group_sum(float v) {
  u32pair a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  u32pair b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// Key blocks (64 keys) that contain at least one non-padded key, in order; fully padded blocks contribute nothing
// to softmax / gradients and are skipped (loads and MFMAs).  Built once per workgroup from the key-padding mask.
// Returns the number of active blocks; act[] (LDS) holds their indices.  All threads must call it.
PQ_DEV int active_key_blocks(const uint8_t* kpm_row, int Lk, uint8_t* flag /*[nkb]*/, uint16_t* act /*[nkb]*/, int* cnt,
                             int tid, int nthreads) {
  const int nkb = (Lk + KB - 1) / KB;
  if (!kpm_row) return -nkb;   // no key-padding mask: every block is active (negative = identity mapping)
  for (int i = tid; i < nkb; i += nthreads) flag[i] = 0;
  __syncthreads();
  for (int j = tid; j < Lk; j += nthreads)
    if (!kpm_row[j]) flag[j / KB] = 1;   // benign race: all writers store 1
  __syncthreads();
  if (tid == 0) {
    int n = 0;
    for (int i = 0; i < nkb; ++i)
      if (flag[i]) act[n++] = (uint16_t)i;
    *cnt = n;
  }
  __syncthreads();
  return *cnt;
}
constexpr int MAXKB = 256;  // key blocks per scene supported by the skip list (Lk <= 16384)

}  // namespace
