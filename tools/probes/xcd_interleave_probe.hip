// Probe (round 6): is the workgroup -> XCD placement of ONE dispatch a pure rotation (XCC_ID == (id + c) % 8, c constant within the
// dispatch) also while OTHER queues dispatch kernels at the same time?  The row-local chains need exactly that: the 8 members of a
// group (ids with equal id % 8) on one XCD.  Stream A: N launches of a 200-workgroup one-wave kernel that records XCC_ID per
// workgroup; streams B, C: a flood of tiny kernels with odd workgroup counts (3, 5, 25) at the same time.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/xcd_interleave_probe tools/probes/xcd_interleave_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void rec(unsigned* seen) {
  unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  if (threadIdx.x == 0) seen[blockIdx.x] = v & 0xf;
}
__global__ void tiny(float* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1.f; }
__global__ void hold(long long ticks) { long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8); }
int main(int argc, char** argv) {
  const int N = 2000, WG = 200;
  unsigned* seen; hipMalloc(&seen, (size_t)N * WG * 4); hipMemset(seen, 0xff, (size_t)N * WG * 4);
  float* junk; hipMalloc(&junk, 4096); hipMemset(junk, 0, 4096);
  hipStream_t a, b, c; hipStreamCreate(&a); hipStreamCreate(&b); hipStreamCreate(&c);
  for (int mode = 0; mode < 3; ++mode) {   // 0: alone, 1: flood of tiny odd-sized kernels, 2: flood of short CU-holding kernels (25 workgroups)
    hipMemset(seen, 0xff, (size_t)N * WG * 4);
    hipDeviceSynchronize();
    for (int i = 0; i < N; ++i) {
      if (mode == 1) { hipLaunchKernelGGL(tiny, dim3(3 + 2 * (i % 3)), dim3(64), 0, b, junk); hipLaunchKernelGGL(tiny, dim3(25), dim3(64), 0, c, junk); }
      if (mode == 2) { hipLaunchKernelGGL(hold, dim3(25), dim3(64), 0, b, 300LL); hipLaunchKernelGGL(hold, dim3(7), dim3(64), 0, c, 500LL); }
      hipLaunchKernelGGL(rec, dim3(WG), dim3(64), 0, a, seen + (size_t)i * WG);
    }
    hipDeviceSynchronize();
    std::vector<unsigned> h((size_t)N * WG); hipMemcpy(h.data(), seen, h.size() * 4, hipMemcpyDeviceToHost);
    int rot[8] = {0}, broken = 0, groups_broken = 0;
    for (int i = 0; i < N; ++i) {
      const unsigned* s = &h[(size_t)i * WG];
      unsigned c0 = (s[0] + 8 - 0) % 8; bool ok = true;
      for (int w = 0; w < WG; ++w) ok = ok && s[w] == (unsigned)((w + c0) % 8);
      if (ok) rot[c0]++; else {
        broken++;
        // the property the chains need: equal id % 8 within each block of 64 consecutive ids -> one XCD
        bool g = true;
        for (int w = 0; w < WG; ++w) g = g && s[w] == s[(w / 64) * 64 + (w % 8)];
        groups_broken += !g;
        if (broken <= 3) { printf("  launch %d:", i); for (int w = 0; w < 40; ++w) printf(" %u", s[w]); printf("\n"); }
      }
    }
    printf("mode %d: %d launches; pure rotations by offset:", mode, N);
    for (int k = 0; k < 8; ++k) printf(" %d", rot[k]);
    printf("; not a pure rotation: %d (of which a 64-id group split over XCDs: %d)\n", broken, groups_broken);
  }
  return 0;
}
