// Empirical lane semantics of ds_read_b64_tr_b16 on gfx950 (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O2 tr_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16 lds_v4i16;
__global__ void k(v4i16* y) {
  __shared__ short tile[64 * 40];  // [row][col], row stride 40; value = row*100 + col
  for (int i = threadIdx.x; i < 64 * 40; i += blockDim.x) tile[i] = (short)((i / 40) * 100 + (i % 40));
  __syncthreads();
  int l = threadIdx.x & 63, li = l & 15, lg = l >> 4;
  short* p = &tile[(4 * lg + (li >> 2)) * 40 + 4 * (li & 3)];   // lane points at row 4*lg + li/4, cols 4*(li%4)..+3
  y[threadIdx.x] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)p);
}
int main() {
  v4i16* d; hipMalloc(&d, 64 * sizeof(v4i16));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  v4i16 h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    int li = l & 15, lg = l >> 4;
    for (int j = 0; j < 4; ++j) { int expect = (4 * lg + j) * 100 + li; if (h[l][j] != expect) ++bad; }
    if (l < 20 || l % 16 == 0) printf("lane %2d: %d %d %d %d\n", l, h[l][0], h[l][1], h[l][2], h[l][3]);
  }
  printf("hypothesis lane(li,lg) elem j == tile[4*lg + j][li]: %s (%d mismatches)\n", bad ? "FALSE" : "TRUE", bad);
  return 0;
}
