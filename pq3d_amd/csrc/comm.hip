// Data-parallel gradient exchange over RCCL / xGMI behind the C-ABI (pq3d_comm_init / pq3d_allreduce_grads; 8e).
// Reference behaviour replaced: DistributedDataParallel's all-reduce(mean) of the parameter gradients every step
// (trainer/build.py:66-75; accelerate wraps the model in DDP, backend 'nccl').  One communicator per process = per GPU.
//
// librccl is bound at the FIRST comm call with dlopen / dlsym: a copy the process already mapped (torch ships its own
// librccl.so) is reused so that one process never runs two RCCL runtimes; otherwise librccl.so.1 of the ROCm install is
// loaded.  libpq3d_hip.so therefore has no link-time dependency on RCCL, and every other entry point works without it.
//
// pq3d_allreduce_grads_wire is the xGMI-minded form: the links are point-to-point (7 x ~153 GB/s per GPU), a ring all-reduce is
// per-link bound, so the fp32 gradients cross them as bf16 -- but are never SUMMED in bf16:
//     pack   : send[r * per + i] = bf16(grads[r * per + i])                                 (one pass over the bucket)
//     a2a    : rank r receives piece r of every rank (grouped ncclSend / ncclRecv)
//     reduce : shard[i] = bf16((sum_r float(recv[r * per + i])) / world)   fp32, rank order  (every rank owns 1 / world)
//     gather : ncclAllGather of the bf16 shards
//     unpack : grads[i] = float(gathered[i])                                                (bit-identical on every rank)
// Bytes on a rank's links: 2 x 2 B x count x (world - 1) / world against 2 x 4 B x ... for the fp32 ring.  The three kernels
// are streams over the bucket (HBM-bound: 6 + 2 (world + 1) / world + 6 bytes per element).
#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <string>

#include <rccl/rccl.h>

#include "common.h"

namespace {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string why;
};

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    // a copy that is already mapped (matched by soname) first: never two RCCL runtimes in one process
    for (const char* n : names)
      if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char* n : names)
      if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!r.lib) { r.why = std::string("librccl not found: ") + (dlerror() ? dlerror() : "dlopen failed"); return; }
    bool ok = true;
    auto sym = [&](auto& fn, const char* name) {
      fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(r.lib, name));
      if (!fn) { ok = false; r.why = std::string("librccl lacks ") + name; }
    };
    sym(r.GetVersion, "ncclGetVersion"); sym(r.GetUniqueId, "ncclGetUniqueId"); sym(r.CommInitRank, "ncclCommInitRank");
    sym(r.CommDestroy, "ncclCommDestroy"); sym(r.AllReduce, "ncclAllReduce"); sym(r.AllGather, "ncclAllGather");
    sym(r.Send, "ncclSend"); sym(r.Recv, "ncclRecv"); sym(r.GroupStart, "ncclGroupStart"); sym(r.GroupEnd, "ncclGroupEnd");
    sym(r.GetErrorString, "ncclGetErrorString");
    if (!ok) r.lib = nullptr;
  });
  return r.lib ? &r : nullptr;
}

constexpr unsigned COMM_MAGIC = 0x70713364u;   // "pq3d"
struct Comm {
  unsigned magic;
  int rank, world, device;
  ncclComm_t nccl;
};

#define PQ_RCCL(call, what)                                                        \
  do {                                                                             \
    const ncclResult_t r_ = (call);                                                \
    if (r_ != ncclSuccess) {                                                       \
      pq3d_set_error((std::string(what ": ") + R->GetErrorString(r_)).c_str());    \
      return 1000 + (int)r_;                                                       \
    }                                                                              \
  } while (0)

// ---- the wire form's three streams (8 elements per thread, 16-byte accesses on the bf16 side) ------------------------------------
__global__ __launch_bounds__(256) void wire_pack_kernel(const float* __restrict__ g, bf16_t* __restrict__ send, long n, long total) {
  long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
  const long stride = (long)gridDim.x * 2048;
  for (; i < total; i += stride) {
    float v[8];
    if (i + 7 < n) {
      const float4 a = *(const float4*)(g + i), b = *(const float4*)(g + i + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = i + j < n ? g[i + j] : 0.f;   // the padding travels as zeros
    }
    *(u32x4*)(send + i) = (u32x4){pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
  }
}

__global__ __launch_bounds__(256) void wire_reduce_kernel(const bf16_t* __restrict__ recv, bf16_t* __restrict__ shard, long per, int world,
                                                          float div) {
  long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
  const long stride = (long)gridDim.x * 2048;
  for (; i < per; i += stride) {
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = 0.f;
    for (int r = 0; r < world; ++r) {   // rank order: the same sum on every rank that owns a copy of this arithmetic
      const u32x4 w = *(const u32x4*)(recv + (long)r * per + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s[2 * j] += __uint_as_float(w[j] << 16);
        s[2 * j + 1] += __uint_as_float(w[j] & 0xffff0000u);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = s[j] / div;
    *(u32x4*)(shard + i) = (u32x4){pack_bf2(s[0], s[1]), pack_bf2(s[2], s[3]), pack_bf2(s[4], s[5]), pack_bf2(s[6], s[7])};
  }
}

__global__ __launch_bounds__(256) void wire_unpack_kernel(const bf16_t* __restrict__ gathered, float* __restrict__ g, long n) {
  long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
  const long stride = (long)gridDim.x * 2048;
  for (; i < n; i += stride) {
    const u32x4 w = *(const u32x4*)(gathered + i);
    float v[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(w[j] << 16); v[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u); }
    if (i + 7 < n) {
      *(float4*)(g + i) = (float4){v[0], v[1], v[2], v[3]};
      *(float4*)(g + i + 4) = (float4){v[4], v[5], v[6], v[7]};
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (i + j < n) g[i + j] = v[j];
    }
  }
}

unsigned grid1d(long threads, long cap) {   // a grid-stride launch: at most `cap` workgroups of 256
  const long g = (threads + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

long wire_per(int world, long count) {
  long per = (count + world - 1) / world;
  return per + ((8 - per % 8) % 8);   // 16-byte pieces
}

Comm* as_comm(void* h) {
  Comm* c = (Comm*)h;
  return c && c->magic == COMM_MAGIC ? c : nullptr;
}

}  // namespace

extern "C" int pq3d_comm_unique_id(void* id) {
  PQ_CHECK_ARG(id != nullptr, "pq3d_comm_unique_id: null buffer (PQ3D_COMM_ID_BYTES bytes)");
  Rccl* R = rccl();
  PQ_CHECK_ARG(R != nullptr, "pq3d_comm_unique_id: librccl could not be bound (dlopen librccl.so / librccl.so.1)");
  static_assert(sizeof(ncclUniqueId) == PQ3D_COMM_ID_BYTES, "unique id size");
  ncclUniqueId u;
  PQ_RCCL(R->GetUniqueId(&u), "ncclGetUniqueId");
  std::memcpy(id, &u, sizeof(u));
  return 0;
}

extern "C" int pq3d_comm_init(int32_t rank, int32_t world, const void* unique_id, void** comm) {
  PQ_CHECK_ARG(comm != nullptr && unique_id != nullptr, "pq3d_comm_init: null argument");
  *comm = nullptr;
  PQ_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "pq3d_comm_init: need 0 <= rank < world");
  Rccl* R = rccl();
  PQ_CHECK_ARG(R != nullptr, "pq3d_comm_init: librccl could not be bound (dlopen librccl.so / librccl.so.1)");
  ncclUniqueId u;
  std::memcpy(&u, unique_id, sizeof(u));
  Comm* c = new Comm{COMM_MAGIC, rank, world, 0, nullptr};
  (void)hipGetDevice(&c->device);
  const ncclResult_t r = R->CommInitRank(&c->nccl, world, u, rank);
  if (r != ncclSuccess) {
    pq3d_set_error((std::string("ncclCommInitRank: ") + R->GetErrorString(r)).c_str());
    delete c;
    return 1000 + (int)r;
  }
  *comm = c;
  return 0;
}

extern "C" int pq3d_comm_destroy(void* comm) {
  Comm* c = as_comm(comm);
  PQ_CHECK_ARG(c != nullptr, "pq3d_comm_destroy: not a communicator handle");
  Rccl* R = rccl();
  PQ_CHECK_ARG(R != nullptr, "pq3d_comm_destroy: librccl is not bound");
  const ncclResult_t r = R->CommDestroy(c->nccl);
  c->magic = 0;
  delete c;
  if (r != ncclSuccess) { pq3d_set_error((std::string("ncclCommDestroy: ") + R->GetErrorString(r)).c_str()); return 1000 + (int)r; }
  return 0;
}

extern "C" int pq3d_comm_info(void* comm, int32_t* rank, int32_t* world, int32_t* rccl_version) {
  Comm* c = as_comm(comm);
  PQ_CHECK_ARG(c != nullptr, "pq3d_comm_info: not a communicator handle");
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  if (rccl_version) {
    Rccl* R = rccl();
    int v = 0;
    if (R) (void)R->GetVersion(&v);
    *rccl_version = v;
  }
  return 0;
}

extern "C" int pq3d_allreduce_grads(void* comm, void* grads, int64_t count, int32_t dtype, int32_t mean, void* stream) {
  Comm* c = as_comm(comm);
  PQ_CHECK_ARG(c != nullptr, "pq3d_allreduce_grads: not a communicator handle");
  PQ_CHECK_ARG(count >= 0 && (grads != nullptr || count == 0), "pq3d_allreduce_grads: null buffer");
  PQ_CHECK_ARG(dtype == PQ3D_F32 || dtype == PQ3D_BF16, "pq3d_allreduce_grads: dtype PQ3D_F32 / PQ3D_BF16");
  if (count == 0) return 0;
  PQ_DEVICE_GUARD(stream, grads);
  Rccl* R = rccl();
  PQ_CHECK_ARG(R != nullptr, "pq3d_allreduce_grads: librccl is not bound");
  PQ_RCCL(R->AllReduce(grads, grads, (size_t)count, dtype == PQ3D_F32 ? ncclFloat32 : ncclBfloat16, mean ? ncclAvg : ncclSum, c->nccl,
                       (hipStream_t)stream), "ncclAllReduce");
  return 0;
}

extern "C" int64_t pq3d_allreduce_wire_scratch_bytes(int32_t world, int64_t count) {
  if (world < 1 || count < 0) return -1;
  const long per = wire_per(world, count);
  return (3L * world + 1) * per * 2;   // send, recv, gathered [world * per] + shard [per], bf16
}

extern "C" int pq3d_allreduce_grads_wire(void* comm, float* grads, int64_t count, void* scratch, int64_t scratch_bytes, int32_t mean,
                                         void* stream) {
  Comm* c = as_comm(comm);
  PQ_CHECK_ARG(c != nullptr, "pq3d_allreduce_grads_wire: not a communicator handle");
  PQ_CHECK_ARG(count >= 0 && (grads != nullptr || count == 0), "pq3d_allreduce_grads_wire: null buffer");
  if (count == 0) return 0;
  const int W = c->world;
  const long per = wire_per(W, count), total = per * W;
  PQ_CHECK_ARG(scratch != nullptr && scratch_bytes >= pq3d_allreduce_wire_scratch_bytes(W, count),
               "pq3d_allreduce_grads_wire: scratch smaller than pq3d_allreduce_wire_scratch_bytes(world, count)");
  PQ_CHECK_ARG(((((uintptr_t)grads) | ((uintptr_t)scratch)) & 15) == 0, "pq3d_allreduce_grads_wire: buffers must be 16-byte aligned");
  PQ_DEVICE_GUARD(stream, grads);
  Rccl* R = rccl();
  PQ_CHECK_ARG(R != nullptr, "pq3d_allreduce_grads_wire: librccl is not bound");
  hipStream_t s = (hipStream_t)stream;
  bf16_t* const send = (bf16_t*)scratch;
  bf16_t* const recv = send + total;
  bf16_t* const gathered = recv + total;
  bf16_t* const shard = gathered + total;
  hipLaunchKernelGGL(wire_pack_kernel, dim3(grid1d((total + 7) / 8, 2048)), dim3(256), 0, s, (const float*)grads, send, (long)count, total);
  PQ_LAUNCH_CHECK();
  // (a one-rank communicator runs the same calls: the self send / receive and the one-rank gather are local copies inside RCCL,
  // so a one-GPU box executes the whole code path)
  PQ_RCCL(R->GroupStart(), "ncclGroupStart");
  for (int r = 0; r < W; ++r) {
    PQ_RCCL(R->Send(send + (long)r * per, (size_t)per, ncclBfloat16, r, c->nccl, s), "ncclSend");
    PQ_RCCL(R->Recv(recv + (long)r * per, (size_t)per, ncclBfloat16, r, c->nccl, s), "ncclRecv");
  }
  PQ_RCCL(R->GroupEnd(), "ncclGroupEnd");
  hipLaunchKernelGGL(wire_reduce_kernel, dim3(grid1d((per + 7) / 8, 2048)), dim3(256), 0, s, (const bf16_t*)recv, shard, per, W, mean ? (float)W : 1.f);
  PQ_LAUNCH_CHECK();
  PQ_RCCL(R->AllGather(shard, gathered, (size_t)per, ncclBfloat16, c->nccl, s), "ncclAllGather");
  hipLaunchKernelGGL(wire_unpack_kernel, dim3(grid1d((count + 7) / 8, 2048)), dim3(256), 0, s, (const bf16_t*)gathered, grads, (long)count);
  PQ_LAUNCH_CHECK();
  return 0;
}

// test hook: the reduce stream of the wire form alone (the W > 1 arithmetic on a one-GPU box): shard[i] = bf16((sum_r float(recv[r * per + i])) / (mean ? world : 1))
extern "C" int pq3d_test_wire_reduce(const void* recv, void* shard, int64_t per, int32_t world, int32_t mean, void* stream) {
  PQ_CHECK_ARG(recv && shard && per >= 8 && per % 8 == 0 && world >= 1, "pq3d_test_wire_reduce: per % 8 == 0, world >= 1");
  PQ_CHECK_ARG(((((uintptr_t)recv) | ((uintptr_t)shard)) & 15) == 0, "pq3d_test_wire_reduce: 16-byte aligned buffers");
  PQ_DEVICE_GUARD(stream, recv);
  hipLaunchKernelGGL(wire_reduce_kernel, dim3(grid1d((per + 7) / 8, 2048)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)recv, (bf16_t*)shard,
                     (long)per, world, mean ? (float)world : 1.f);
  PQ_LAUNCH_CHECK();
  return 0;
}
