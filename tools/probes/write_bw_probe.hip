// Achievable HBM write bandwidth by data and by store shape (537 MB = the hoisted K/V tensor of config 5).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/write_bw_probe tools/probes/write_bw_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
// mode 0: constant, 1: hashed data; coalesced 1 KB per wave instruction
__global__ void stream_write(u32x4* out, long n16, int mode) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) {
    u32x4 v = (u32x4){0x3fa03fa0u, 0x3fa03fa0u, 0x3fa03fa0u, 0x3fa03fa0u};
    if (mode) { const uint32_t h = mix((uint32_t)i); v = (u32x4){h, h * 3u, h * 5u, h * 7u}; }
    out[i] = v;
  }
}
// GEMM-epilogue shape: a workgroup (256 threads) owns a contiguous run of rows of a [R][256] bf16 matrix; a wave instruction
// writes 16 rows x 64 B (seg = 4) or 4 rows x 256 B (seg = 16) or 1 KB contiguous (seg = 64)
__global__ void tile_write(u32x4* out, long rows, int rows_per_wg, int seg, int mode) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long r_lo = (long)blockIdx.x * rows_per_wg;
  const int rows_per_inst = 64 / seg;                 // rows covered by one wave instruction
  for (int r0 = wave * 16; r0 < rows_per_wg; r0 += 4 * 16) {      // each wave: 16-row blocks
    for (int c0 = 0; c0 < 32; c0 += seg)                           // 32 16-byte chunks per 512-byte row
      for (int rr = 0; rr < 16; rr += rows_per_inst) {
        const long row = r_lo + r0 + rr + lane / seg;
        const int ch = c0 + lane % seg;
        if (row < rows) {
          const uint32_t h = mode ? mix((uint32_t)(row * 32 + ch)) : 0x3fa03fa0u;
          out[row * 32 + ch] = (u32x4){h, h * 3u, h * 5u, h * 7u};
        }
      }
  }
}
int main() {
  const long bytes = 537L * 1000 * 1000, n16 = bytes / 16, rows = bytes / 512;
  u32x4* d; hipMalloc(&d, bytes + 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-52s %8.1f us  %.2f TB/s\n", name, ms * 100, bytes / (ms * 1e-4) / 1e6 / 1e6);
  };
  timeit("stream constant (grid 2048 x 256)", [&] { stream_write<<<2048, 256>>>(d, n16, 0); });
  timeit("stream hashed   (grid 2048 x 256)", [&] { stream_write<<<2048, 256>>>(d, n16, 1); });
  timeit("stream hashed   (grid 256 x 256)", [&] { stream_write<<<256, 256>>>(d, n16, 1); });
  timeit("stream hashed   (grid 256 x 1024)", [&] { stream_write<<<256, 1024>>>(d, n16, 1); });
  for (int seg : {4, 16, 64}) for (int rpw : {128, 4096}) for (int mode : {0, 1}) {
    char nm[96]; snprintf(nm, 96, "tile seg %2d rows/wg %4d %s", seg, rpw, mode ? "hashed" : "constant");
    const int grid = (int)((rows + rpw - 1) / rpw);
    timeit(nm, [&] { tile_write<<<grid, 256>>>(d, rows, rpw, seg, mode); });
  }
  return 0;
}
