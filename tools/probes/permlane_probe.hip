// Semantics of v_permlane16_swap / v_permlane32_swap on gfx950 used as xor-16 / xor-32 reductions.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* y) {
  unsigned x = threadIdx.x;
  u32x2 a = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  u32x2 b = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  y[threadIdx.x * 4 + 0] = a[0]; y[threadIdx.x * 4 + 1] = a[1];
  y[threadIdx.x * 4 + 2] = b[0]; y[threadIdx.x * 4 + 3] = b[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 64 * 4 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad16 = 0, bad32 = 0;
  for (int l = 0; l < 64; ++l) {
    // hypothesis: {a0,a1} == {l, l^16} as a set; {b0,b1} == {l, l^32} as a set
    unsigned a0 = h[l*4], a1 = h[l*4+1], b0 = h[l*4+2], b1 = h[l*4+3];
    if (!((a0 == (unsigned)l && a1 == (unsigned)(l ^ 16)) || (a1 == (unsigned)l && a0 == (unsigned)(l ^ 16)))) ++bad16;
    if (!((b0 == (unsigned)l && b1 == (unsigned)(l ^ 32)) || (b1 == (unsigned)l && b0 == (unsigned)(l ^ 32)))) ++bad32;
    if (l % 8 == 0) printf("lane %2d: p16 = (%u,%u)  p32 = (%u,%u)\n", l, a0, a1, b0, b1);
  }
  printf("permlane16_swap(x,x) gives {x[l], x[l^16]}: %s; permlane32_swap(x,x) gives {x[l], x[l^32]}: %s\n", bad16 ? "FALSE" : "TRUE", bad32 ? "FALSE" : "TRUE");
  return 0;
}
