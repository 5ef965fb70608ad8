// Background lane (round 4): the primitives that let TWO captured HIP graphs, replayed on two streams, run as one step.
//
// The decoder backward is a chain of ~40 small dependent launches (52-192 workgroups each on a 256-CU chip) followed by
// work that depends only on per-layer intermediates: weight gradients, K/V weight gradients, the spatial-bias projection
// gradient.  On this runtime forked branches of ONE captured graph do not run concurrently (tools/probes/
// graph_branch_probe.py), two graphs on two streams do (tools/probes/overlap_probe.py, cumask_probe.py) -- provided the
// background stream cannot take every CU: a full-width background launch in front of each small chain launch made the
// pair slower than serial, a stream restricted to 128 CUs by a CU mask (16 per XCD: mask bit i selects a CU of XCD
// i % 8, tools/probes/cumask_map_probe.py) overlapped cleanly.  A captured graph cannot wait on an event of another
// graph, so the hand-offs are device-side flags: the main graph publishes "layer a's backward is complete" with a
// one-thread kernel (stream order makes everything launched before it complete and visible at agent scope: the
// end-of-kernel release of its predecessors), the background graph holds a one-thread poller in front of the work that
// needs it.  Flags carry an epoch that both graphs bump once per replay, so nothing is ever reset and a replay cannot
// see the previous step's flags.  A poller gives up after `timeout_us` (serialising profilers, a background graph
// replayed alone) and raises the error word instead of hanging the device.
#include "common.h"

namespace {
// `ts` (optional): three 100 MHz wall-clock stamps per kernel -- a profiler that serialises queues cannot show the overlap
// of the two graphs, these stamps can (BackgroundLane.timeline())
__global__ void lane_bump_kernel(uint32_t* counter, long* ts) {
  *counter = *counter + 1u;
  if (ts) ts[0] = (long)wall_clock64();
}

__global__ void lane_signal_kernel(uint32_t* flag, const uint32_t* epoch, long* ts) {
  __hip_atomic_store(flag, *epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  if (ts) ts[0] = (long)wall_clock64();
}

// one lane polls with relaxed agent-scope loads (L2-served, no L1 hit) and sleeps in between; one acquire at the end.
// The kernels behind it in the stream start after it retires and begin with their own kernel-start acquire.
__global__ void lane_wait_kernel(const uint32_t* flag, const uint32_t* epoch, long timeout_ticks, uint32_t* err, long* ts) {
  const uint32_t want = *epoch;
  const long t0 = (long)wall_clock64();   // constant 100 MHz counter
  if (ts) ts[1] = t0;
  bool ok = true;
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) {   // exact: desynchronised epochs time out
    __builtin_amdgcn_s_sleep(32);
    if ((long)wall_clock64() - t0 > timeout_ticks) { ok = false; break; }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (ts) ts[2] = (long)wall_clock64();
  if (!ok) atomicAdd(err, 1u);
}
}  // namespace

extern "C" int pq3d_lane_stream_create(const uint32_t* cu_mask, int32_t words, void** stream) {
  PQ_CHECK_ARG(cu_mask && words >= 1 && words <= 32 && stream, "pq3d_lane_stream_create: bad args");
  hipStream_t s = nullptr;
  const hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, cu_mask);
  if (e != hipSuccess) { pq3d_set_error(hipGetErrorString(e)); return (int)e; }
  *stream = (void*)s;
  return 0;
}

extern "C" int pq3d_lane_stream_destroy(void* stream) {
  PQ_CHECK_ARG(stream, "pq3d_lane_stream_destroy: null stream");
  const hipError_t e = hipStreamDestroy((hipStream_t)stream);
  if (e != hipSuccess) { pq3d_set_error(hipGetErrorString(e)); return (int)e; }
  return 0;
}

extern "C" int pq3d_lane_bump(uint32_t* counter, int64_t* ts, void* stream) {
  PQ_DEVICE_GUARD(stream, counter);
  PQ_CHECK_ARG(counter, "pq3d_lane_bump: null counter");
  hipLaunchKernelGGL(lane_bump_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter, (long*)ts);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_lane_signal(uint32_t* flag, const uint32_t* epoch, int64_t* ts, void* stream) {
  PQ_DEVICE_GUARD(stream, flag);
  PQ_CHECK_ARG(flag && epoch, "pq3d_lane_signal: null pointer");
  hipLaunchKernelGGL(lane_signal_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, flag, epoch, (long*)ts);
  PQ_LAUNCH_CHECK();
  return 0;
}

extern "C" int pq3d_lane_wait(const uint32_t* flag, const uint32_t* epoch, int64_t timeout_us, uint32_t* err, int64_t* ts,
                              void* stream) {
  PQ_DEVICE_GUARD(stream, flag);
  PQ_CHECK_ARG(flag && epoch && err && timeout_us > 0, "pq3d_lane_wait: null pointer / non-positive timeout");
  hipLaunchKernelGGL(lane_wait_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, flag, epoch, (long)timeout_us * 100, err,
                     (long*)ts);
  PQ_LAUNCH_CHECK();
  return 0;
}
