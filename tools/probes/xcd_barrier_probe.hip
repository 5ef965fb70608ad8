// Cost of a hand-off between the workgroups of ONE XCD inside a persistent kernel (no cross-XCD coherence needed):
//   producer: stores, s_waitcnt vmcnt(0) (write-through L1: acknowledged = in the XCD's L2), one relaxed L2 atomic;
//   consumer: polls the counter at L2, buffer_inv sc1 (drop stale L1 lines), plain loads.
// A kernel boundary in a replayed graph costs ~4.7 us launch-to-launch; an agent-scope release (buffer_wbl2) 1.7-6.5 us.
// Workgroup id % 8 = XCD (observed placement, checked here through XCC_ID).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/xcd_barrier_probe tools/probes/xcd_barrier_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
constexpr int WPX = 32;         // workgroups per XCD (one per CU)
constexpr int N = 2048;         // floats written per workgroup per round (8 KB)
__device__ inline unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }
__device__ inline void xcd_barrier(unsigned* counter, unsigned target) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's stores are in L2
  __syncthreads();                                            // ... and every other wave's of the workgroup
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // buffer_inv sc1: stale L1 lines of other CUs' data
}
// flag form: every workgroup publishes its round in its own slot (plain store), lanes 0..31 of wave 0 each watch one slot
// with L1-bypassing loads: no read-modify-write at all
__device__ inline void xcd_barrier_flags(unsigned* flags, int l, unsigned round, int variant) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x < 64) {
    if (threadIdx.x == 0) __hip_atomic_store(flags + l * 16, round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int w = threadIdx.x & 31;
    while (true) {
      const unsigned v = __hip_atomic_load(flags + w * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__all((int)(v >= round))) break;
      if (variant & 1) __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  if (!(variant & 2)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
__global__ __launch_bounds__(256) void probe_flags(float* buf, unsigned* flags_all, unsigned* bad, int rounds, int variant) {
  const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
  float* mine = buf + ((long)xcd * WPX + l) * N;
  const float* next = buf + ((long)xcd * WPX + (l + 1) % WPX) * N;
  unsigned* flags = flags_all + xcd * WPX * 16;
  unsigned errs = 0, t = 0;
  for (int r = 0; r < rounds; ++r) {
    for (int i = threadIdx.x; i < N; i += blockDim.x) mine[i] = (float)(r * 131 + l * 7 + i);
    xcd_barrier_flags(flags, l, ++t, variant);
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
      float v;
      if (variant & 4) {   // L1-bypassing load (sc1): reads the XCD's L2, where the producer's write-through store is
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)next, 0, N * 4, 0x00020000);
        v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, i * 4, 0, 16));
      } else v = (variant & 2) ? __builtin_nontemporal_load(next + i) : next[i];
      errs += v != (float)(r * 131 + ((l + 1) % WPX) * 7 + i);
    }
    xcd_barrier_flags(flags, l, ++t, variant);
  }
  if (errs) atomicAdd(bad, errs);
}
__global__ __launch_bounds__(256) void probe(float* buf, unsigned* counters, unsigned* bad, unsigned* xcc_seen, int rounds, int mode) {
  const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
  if (threadIdx.x == 0) xcc_seen[blockIdx.x] = xcc_id();
  float* mine = buf + ((long)xcd * WPX + l) * N;
  const float* next = buf + ((long)xcd * WPX + (l + 1) % WPX) * N;
  unsigned* ctr = counters + xcd * 64;   // one counter per XCD, own cache line
  unsigned errs = 0, t = 0;
  for (int r = 0; r < rounds; ++r) {
    for (int i = threadIdx.x; i < N; i += blockDim.x) mine[i] = (float)(r * 131 + l * 7 + i);
    t += WPX; xcd_barrier(ctr, t);
    if (mode) for (int i = threadIdx.x; i < N; i += blockDim.x) errs += next[i] != (float)(r * 131 + ((l + 1) % WPX) * 7 + i);
    t += WPX; xcd_barrier(ctr, t);       // nobody overwrites before everybody has read
  }
  if (errs) atomicAdd(bad, errs);
}
int main() {
  float* buf; unsigned *ctr, *bad, *seen;
  hipMalloc(&buf, 8L * WPX * N * 4); hipMalloc(&ctr, 8 * 64 * 4); hipMalloc(&bad, 4); hipMalloc(&seen, 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) for (int rounds : {100, 1100}) {
    hipMemset(ctr, 0, 8 * 64 * 4); hipMemset(bad, 0, 4);
    hipEventRecord(e0);
    probe<<<8 * WPX, 256>>>(buf, ctr, bad, seen, rounds, mode);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned b; hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost);
    printf("mode %d (%s) rounds %4d: %8.1f us total, mismatches %u\n", mode, mode ? "write + barrier + read + barrier" : "write + 2 barriers", rounds, ms * 1e3, b);
  }
  unsigned* flags; hipMalloc(&flags, 8 * WPX * 16 * 4);
  for (int variant : {0, 2, 6, 7}) for (int rounds : {100, 1100}) {
    hipMemset(flags, 0, 8 * WPX * 16 * 4); hipMemset(bad, 0, 4);
    hipEventRecord(e0);
    probe_flags<<<8 * WPX, 256>>>(buf, flags, bad, rounds, variant);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned b; hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost);
    printf("flags variant %d (sleep %d, %s) rounds %4d: %8.1f us total, mismatches %u\n", variant, variant & 1, (variant & 4) ? "no fence, sc1 buffer loads" : (variant & 2) ? "no fence, nontemporal reads" : "acquire fence", rounds, ms * 1e3, b);
  }
  std::vector<unsigned> s(256); hipMemcpy(s.data(), seen, 256 * 4, hipMemcpyDeviceToHost);
  int okx = 0; for (int i = 0; i < 256; ++i) okx += (s[i] == (unsigned)(i & 7));
  printf("workgroups with XCC_ID == id %% 8: %d / 256\n", okx);
  return 0;
}
