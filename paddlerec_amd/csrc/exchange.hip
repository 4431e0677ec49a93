// Row-sharded table exchange behind the C-ABI (SURVEY.md §8(b) `alltoall_exchange (takes ncclComm_t)`).
//
// Reference counterpart: the inter-GPU pull / push shuffles of core.PSGPU inside exe.train_from_dataset
// (/root/reference/tools/static_gpubox_trainer.py:152-160,256) [EXT HeterPS].  Here: the three all-to-all(v) rounds of
// a sharded step (ids to owners, rows back, row-gradients to owners) and the dense-gradient all-reduce, issued on the
// caller's stream over an RCCL communicator — all-to-all drives the 7 xGMI links of a GPU at once.
//
// RCCL is bound at RUN time (dlsym): a process that already hosts an RCCL (PyTorch-ROCm bundles one) must not get a
// second copy through this library's link line, and a process without any GPU work can still load librecengine.so.
// The communicator is either created here (rec_comm_unique_id / rec_comm_init: one per process, ranks = GPUs) or handed
// in by the framework (a Paddle binder passes the ncclComm_t of its own collective context).
#include <dlfcn.h>
#include <stdint.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "rec_common.h"

namespace rec {
namespace {

// the slice of rccl.h this file needs (ABI-stable NCCL 2 signatures)
typedef struct { char internal[128]; } NcclUniqueId;
typedef int (*FnGetUniqueId)(NcclUniqueId*);
typedef int (*FnCommInitRank)(void**, int, NcclUniqueId, int);
typedef int (*FnCommDestroy)(void*);
typedef int (*FnCommCount)(void*, int*);
typedef int (*FnAllToAllv)(const void*, const size_t*, const size_t*, void*, const size_t*, const size_t*, int,
                           void*, hipStream_t);
typedef int (*FnAllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*FnGetErrorString)(int);
constexpr int kNcclInt8 = 0, kNcclFloat32 = 7, kNcclSum = 0;

struct Rccl {
  FnGetUniqueId get_unique_id = nullptr;
  FnCommInitRank comm_init_rank = nullptr;
  FnCommDestroy comm_destroy = nullptr;
  FnCommCount comm_count = nullptr;
  FnAllToAllv alltoallv = nullptr;
  FnAllReduce allreduce = nullptr;
  FnGetErrorString err = nullptr;
  bool ok = false;
};

const Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    void* h = RTLD_DEFAULT;
    if (!dlsym(h, "ncclAllToAllv")) {   // no RCCL in the process yet: load the system one
      for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
      }
      if (!h) return;
    }
    r.get_unique_id = (FnGetUniqueId)dlsym(h, "ncclGetUniqueId");
    r.comm_init_rank = (FnCommInitRank)dlsym(h, "ncclCommInitRank");
    r.comm_destroy = (FnCommDestroy)dlsym(h, "ncclCommDestroy");
    r.comm_count = (FnCommCount)dlsym(h, "ncclCommCount");
    r.alltoallv = (FnAllToAllv)dlsym(h, "ncclAllToAllv");
    r.allreduce = (FnAllReduce)dlsym(h, "ncclAllReduce");
    r.err = (FnGetErrorString)dlsym(h, "ncclGetErrorString");
    r.ok = r.get_unique_id && r.comm_init_rank && r.comm_destroy && r.comm_count && r.alltoallv && r.allreduce;
  });
  return r;
}

int nccl_fail(const char* what, int rc) {
  const Rccl& r = rccl();
  set_error("%s: RCCL error %d (%s)", what, rc, r.err ? r.err(rc) : "?");
  return REC_ENCCL;
}

}  // namespace
}  // namespace rec

using namespace rec;

extern "C" int rec_comm_unique_id(void* id128) {
  REC_REQUIRE(id128, REC_EINVAL, "id buffer is NULL");
  const Rccl& r = rccl();
  REC_REQUIRE(r.ok, REC_ENCCL, "RCCL is not available in this process");
  NcclUniqueId id;
  if (int rc = r.get_unique_id(&id)) return nccl_fail("ncclGetUniqueId", rc);
  memcpy(id128, &id, sizeof(id));
  return REC_OK;
}

extern "C" int rec_comm_init(const void* id128, int32_t world, int32_t rank, void** comm) {
  REC_REQUIRE(id128 && comm && world >= 1 && rank >= 0 && rank < world, REC_EINVAL, "bad arguments");
  const Rccl& r = rccl();
  REC_REQUIRE(r.ok, REC_ENCCL, "RCCL is not available in this process");
  NcclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  void* c = nullptr;
  if (int rc = r.comm_init_rank(&c, world, id, rank)) return nccl_fail("ncclCommInitRank", rc);
  *comm = c;
  return REC_OK;
}

extern "C" int rec_comm_destroy(void* comm) {
  if (!comm) return REC_OK;
  const Rccl& r = rccl();
  REC_REQUIRE(r.ok, REC_ENCCL, "RCCL is not available in this process");
  if (int rc = r.comm_destroy(comm)) return nccl_fail("ncclCommDestroy", rc);
  return REC_OK;
}

// Number of ranks of the communicator, as RCCL itself reports it (ncclCommCount).
extern "C" int rec_comm_size(void* comm, int32_t* world) {
  REC_REQUIRE(comm && world, REC_EINVAL, "bad arguments");
  const Rccl& r = rccl();
  REC_REQUIRE(r.ok, REC_ENCCL, "RCCL is not available in this process");
  int w = 0;
  if (int rc = r.comm_count(comm, &w)) return nccl_fail("ncclCommCount", rc);
  *world = w;
  return REC_OK;
}

// 1 when RCCL's entry points resolve in this process (rec_comm_* usable), else 0.  Never fails.
extern "C" int rec_comm_available(void) { return rccl().ok ? 1 : 0; }

// send: rows grouped by destination rank (send_counts[d] rows to rank d, ascending d); recv: rows grouped by source.
// Counts are HOST arrays of `world` entries, in rows of row_bytes bytes.
extern "C" int rec_alltoall_exchange(void* comm, const void* send, const int64_t* send_counts, void* recv,
                                     const int64_t* recv_counts, int32_t row_bytes, void* stream) {
  REC_REQUIRE(comm && send_counts && recv_counts && row_bytes > 0, REC_EINVAL, "bad arguments");
  const Rccl& r = rccl();
  REC_REQUIRE(r.ok, REC_ENCCL, "RCCL is not available in this process");
  int world = 0;
  if (int rc = r.comm_count(comm, &world)) return nccl_fail("ncclCommCount", rc);
  std::vector<size_t> sc(world), sd(world), rc_(world), rd(world);
  size_t so = 0, ro = 0;
  for (int i = 0; i < world; ++i) {
    REC_REQUIRE(send_counts[i] >= 0 && recv_counts[i] >= 0, REC_EINVAL, "negative count");
    sc[i] = (size_t)send_counts[i] * row_bytes;
    rc_[i] = (size_t)recv_counts[i] * row_bytes;
    sd[i] = so;
    rd[i] = ro;
    so += sc[i];
    ro += rc_[i];
  }
  REC_REQUIRE((so == 0 || send) && (ro == 0 || recv), REC_EINVAL, "null buffer with a non-zero count");
  if (so == 0 && ro == 0 && world == 1) return REC_OK;
  if (int rc = r.alltoallv(send, sc.data(), sd.data(), recv, rc_.data(), rd.data(), kNcclInt8, comm,
                           (hipStream_t)stream))
    return nccl_fail("ncclAllToAllv", rc);
  return REC_OK;
}

extern "C" int rec_allreduce_sum_f32(void* comm, float* buf, int64_t n, void* stream) {
  REC_REQUIRE(comm && n >= 0 && (n == 0 || buf), REC_EINVAL, "bad arguments");
  if (n == 0) return REC_OK;
  const Rccl& r = rccl();
  REC_REQUIRE(r.ok, REC_ENCCL, "RCCL is not available in this process");
  if (int rc = r.allreduce(buf, buf, (size_t)n, kNcclFloat32, kNcclSum, comm, (hipStream_t)stream))
    return nccl_fail("ncclAllReduce", rc);
  return REC_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// A stand-in link for ONE-GPU measurements of the row-sharded step: what an RCCL all-to-all looks
// like to the rest of the chip — a handful of resident workgroups (RCCL's channels) that move bytes at the LINK's rate,
// not HBM's.  `blocks` workgroups of 256 threads copy `bytes` from src to dst (both device; may alias rings of any
// size >= ring_bytes: the copy wraps) in 64 KB slices and pace themselves with the 100 MHz wall clock so that the launch
// lasts bytes / gbytes_per_s + fixed_us.  It holds its wave slots for that long, as the real collective would.
namespace rec {
__global__ __launch_bounds__(256) void link_emulate_kernel(const char* __restrict__ src, char* __restrict__ dst,
                                                           size_t bytes, unsigned ring_mask, double ticks_per_byte,
                                                           long long fixed_ticks) {
  const long long t0 = wall_clock64();
  const long long end = t0 + fixed_ticks + (long long)((double)bytes * ticks_per_byte);
  const size_t share = bytes / gridDim.x;
  const unsigned base = (unsigned)(((size_t)blockIdx.x * share) & ring_mask);
  constexpr unsigned kIter = 4 * 256 * 16;              // four 16-byte loads per thread in flight: 16 KB per iteration
  size_t moved = 0;
  // The launch lasts bytes / rate + fixed whatever the copy manages: a workgroup moves its share while it is ahead of
  // the link's schedule and sleeps otherwise (a few workgroups cannot outrun seven xGMI links; they hold their wave
  // slots and touch memory as the collective's channels would)
  while (true) {
    const long long now = wall_clock64();
    if (now >= end) break;
    const double sched = (double)(now - t0 - fixed_ticks) / ticks_per_byte / gridDim.x;      // bytes the link has taken
    if (moved < share && (double)moved <= sched) {
      uint4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        v[u] = *reinterpret_cast<const uint4*>(src + ((base + (unsigned)moved + (u * 256 + threadIdx.x) * 16) & ring_mask));
#pragma unroll
      for (int u = 0; u < 4; ++u)
        *reinterpret_cast<uint4*>(dst + ((base + (unsigned)moved + (u * 256 + threadIdx.x) * 16) & ring_mask)) = v[u];
      moved += kIter;
    } else {
      __builtin_amdgcn_s_sleep(16);
    }
  }
}
}  // namespace rec

extern "C" int rec_link_emulate(size_t bytes, float gbytes_per_s, float fixed_us, int32_t blocks, const void* src,
                                void* dst, size_t ring_bytes, void* stream) {
  REC_REQUIRE(gbytes_per_s > 0.f && fixed_us >= 0.f && blocks >= 1 && blocks <= 64, REC_EINVAL, "bad arguments");
  REC_REQUIRE(src && dst && ring_bytes >= 65536 && ring_bytes <= (1u << 31) && (ring_bytes & (ring_bytes - 1)) == 0 &&
                  ((uintptr_t)src) % 16 == 0 && ((uintptr_t)dst) % 16 == 0, REC_EINVAL,
              "src / dst: 16-byte aligned rings of a power of two >= 64 KB");
  int dev = 0, khz = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0)
    khz = 100000;   // gfx9: a constant 100 MHz counter
  const double ticks_per_us = khz / 1000.0;
  const double ticks_per_byte = ticks_per_us / ((double)gbytes_per_s * 1e3);      // GB/s = 1e3 bytes per us
  hipLaunchKernelGGL(link_emulate_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const char*)src,
                     (char*)dst, bytes, (unsigned)(ring_bytes - 1) & ~15u, ticks_per_byte, (long long)(fixed_us * ticks_per_us));
  return check_launch("rec_link_emulate");
}
