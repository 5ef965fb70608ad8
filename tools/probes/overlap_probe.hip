// Probe kernels for the "background stream gated by device-side flags" experiment (round 4).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/overlap_probe.hip -o tools/probes/liboverlap_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>

// FMA loop: `iters` dependent fused multiply-adds per thread (~4 cycles each) -> a kernel of predictable duration
__global__ void busy_kernel(float* out, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.000001f;
  for (int i = 0; i < iters; ++i) a = a * b + 1e-7f;
  if (a == 12345.678f) out[0] = a;
}
// writes val into buf[0..n): the "producer" data the consumer checks after the flag
__global__ void produce_kernel(uint32_t* buf, long n, const uint32_t* epoch) {
  const uint32_t v = *epoch;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) buf[i] = v;
}
// counts elements != *epoch into err (visibility check)
__global__ void consume_kernel(const uint32_t* buf, long n, const uint32_t* epoch, uint32_t* err) {
  const uint32_t v = *epoch;
  uint32_t bad = 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) bad += buf[i] != v;
  if (bad) atomicAdd(err, bad);
}
__global__ void bump_kernel(uint32_t* epoch) { *epoch += 1; }
// flag := *epoch (release, agent scope).  Stream order makes everything launched before this kernel complete and visible.
__global__ void signal_kernel(uint32_t* flag, const uint32_t* epoch) {
  __hip_atomic_store(flag, *epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// spin until *flag >= *epoch (acquire, agent scope) or `timeout` polls elapsed (then *err += 1<<20)
__global__ void wait_kernel(const uint32_t* flag, const uint32_t* epoch, long timeout, uint32_t* err) {
  const uint32_t want = *epoch;
  long n = 0;
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
    __builtin_amdgcn_s_sleep(8);
    if (++n > timeout) { atomicAdd(err, 1u << 20); break; }
  }
}

extern "C" {
int op_busy(float* out, int grid, int block, int iters, void* s) {
  busy_kernel<<<grid, block, 0, (hipStream_t)s>>>(out, iters);
  return (int)hipGetLastError();
}
int op_produce(uint32_t* buf, long n, const uint32_t* epoch, int grid, void* s) {
  produce_kernel<<<grid, 256, 0, (hipStream_t)s>>>(buf, n, epoch);
  return (int)hipGetLastError();
}
int op_consume(const uint32_t* buf, long n, const uint32_t* epoch, uint32_t* err, int grid, void* s) {
  consume_kernel<<<grid, 256, 0, (hipStream_t)s>>>(buf, n, epoch, err);
  return (int)hipGetLastError();
}
int op_bump(uint32_t* epoch, void* s) { bump_kernel<<<1, 1, 0, (hipStream_t)s>>>(epoch); return (int)hipGetLastError(); }
int op_signal(uint32_t* flag, const uint32_t* epoch, void* s) {
  signal_kernel<<<1, 1, 0, (hipStream_t)s>>>(flag, epoch);
  return (int)hipGetLastError();
}
int op_wait(const uint32_t* flag, const uint32_t* epoch, long timeout, uint32_t* err, void* s) {
  wait_kernel<<<1, 1, 0, (hipStream_t)s>>>(flag, epoch, timeout, err);
  return (int)hipGetLastError();
}
// stream memory operations (command-processor waits: no wave occupied)
int op_stream_wait32(void* s, void* ptr, uint32_t value) {
  return (int)hipStreamWaitValue32((hipStream_t)s, ptr, value, hipStreamWaitValueGte, 0xffffffffu);
}
int op_stream_write32(void* s, void* ptr, uint32_t value) { return (int)hipStreamWriteValue32((hipStream_t)s, ptr, value, 0); }
}

// CU-masked stream: every kernel launched on it (any grid) runs only on the CUs whose mask bit is set
extern "C" void* op_stream_masked(const uint32_t* mask, uint32_t words) {
  hipStream_t s = nullptr;
  if (hipExtStreamCreateWithCUMask(&s, words, mask) != hipSuccess) return nullptr;
  return (void*)s;
}

// where does a workgroup run?  out[block] = XCC_ID | HW_ID << 4  (HW_ID: cu_id bits 8-11, sh_id 12, se_id 13-15 on gfx9)
__global__ void where_kernel(uint32_t* out) {
  if (threadIdx.x == 0) {
    const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);       // HW_REG_XCC_ID[3:0]
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID
    out[blockIdx.x] = (xcc & 15) | (hw << 4);
  }
  float a = threadIdx.x;
  for (int i = 0; i < 2000; ++i) a = a * 1.0001f + 1e-7f;   // keep the block alive so that the grid spreads
  if (a == 1.2345f) out[0] = 0;
}
extern "C" int op_where(uint32_t* out, int grid, void* s) {
  where_kernel<<<grid, 64, 0, (hipStream_t)s>>>(out);
  return (int)hipGetLastError();
}
