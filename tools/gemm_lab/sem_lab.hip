// LAB (not run in round 4 — written for the next round): the synchronisation primitive of a one-launch split-K, alone.
// nblk blocks per tile each write a partial vector of `len` floats, arrive at the tile's counter, and the block that arrives
// last sums the nblk partial vectors in ascending order.  Two forms of making the partial vectors visible across the eight
// XCD L2s of the part:
//   mode 0  plain stores, __threadfence() (agent-scope release: buffer_wbl2) / counter / __threadfence() (acquire:
//           buffer_inv), plain loads                    -- what profiles/r04_splitk_sem.txt measured inside the GEMM: ~50 us
//   mode 1  agent-scope relaxed atomic stores (write-through, sc1), s_waitcnt vmcnt(0), counter (relaxed), agent-scope
//           relaxed atomic loads (sc1)                   -- no cache-wide operation; relies on an sc1 store being visible at
//           agent scope once vmcnt has counted it
// tools/gemm_lab/sem_lab.py changes the data every launch (a stale partial vector of the previous launch is then a wrong sum)
// and counts mismatches over thousands of launches.
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ int g_sem[4096];

template <int MODE>
__global__ __launch_bounds__(256) void sem_kernel(int tiles, int nblk, int len, const float* __restrict__ src, float* partial,
                                                  float* __restrict__ out, int round) {
  const int tile = blockIdx.x % tiles, z = blockIdx.x / tiles;
  float* mine = partial + ((size_t)z * tiles + tile) * len;
  for (int i = threadIdx.x; i < len; i += 256) {
    const float v = src[((size_t)z * tiles + tile) * len + i] + (float)round;
    if (MODE == 0) mine[i] = v;
    else __hip_atomic_store(mine + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __shared__ int s_last;
  if (MODE == 0) __threadfence();
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const int old = MODE == 0 ? __hip_atomic_fetch_add(g_sem + tile, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT)
                              : __hip_atomic_fetch_add(g_sem + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = old == nblk - 1;
    if (s_last) __hip_atomic_store(g_sem + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!s_last) return;
  if (MODE == 0) __threadfence();
  for (int i = threadIdx.x; i < len; i += 256) {
    float t = 0.f;
    for (int q = 0; q < nblk; ++q) {
      const float* p = partial + ((size_t)q * tiles + tile) * len + i;
      t += MODE == 0 ? *p : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    out[(size_t)tile * len + i] = t;
  }
}

// the two-launch reference: partial vectors by one kernel, their sums by the next
__global__ __launch_bounds__(256) void part_kernel(int tiles, int len, const float* __restrict__ src, float* __restrict__ partial,
                                                   int round) {
  const size_t b = (size_t)blockIdx.x * len;
  for (int i = threadIdx.x; i < len; i += 256) partial[b + i] = src[b + i] + (float)round;
}
__global__ __launch_bounds__(256) void sum_kernel(int tiles, int nblk, int len, const float* __restrict__ partial,
                                                  float* __restrict__ out) {
  const int tile = blockIdx.x;
  for (int i = threadIdx.x; i < len; i += 256) {
    float t = 0.f;
    for (int q = 0; q < nblk; ++q) t += partial[((size_t)q * tiles + tile) * len + i];
    out[(size_t)tile * len + i] = t;
  }
}

extern "C" int lab_sem(int mode, int tiles, int nblk, int len, const float* src, float* partial, float* out, int round,
                       void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (tiles > 4096) return 1;
  if (mode == 0) hipLaunchKernelGGL(sem_kernel<0>, dim3(tiles * nblk), dim3(256), 0, st, tiles, nblk, len, src, partial, out, round);
  else if (mode == 1) hipLaunchKernelGGL(sem_kernel<1>, dim3(tiles * nblk), dim3(256), 0, st, tiles, nblk, len, src, partial, out, round);
  else {
    hipLaunchKernelGGL(part_kernel, dim3(tiles * nblk), dim3(256), 0, st, tiles, len, src, partial, round);
    hipLaunchKernelGGL(sum_kernel, dim3(tiles), dim3(256), 0, st, tiles, nblk, len, (const float*)partial, out);
  }
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
