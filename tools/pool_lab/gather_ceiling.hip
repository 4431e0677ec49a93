// What HBM delivers for row P's traffic MIX with none of its bookkeeping: a lane reads one 8-byte id, gathers the 36-byte
// row it names (a 64-byte-aligned record in a 10 M-row table: one 64-B sector per id), and the kernel writes the bytes
// the pooling kernel writes (pooled output, rows_out, seg_of_value) as plain streams.  No LoD, no hash, no segments, no
// LDS: the time of this kernel is the floor any multi-slot pooling kernel on this chip is measured against.
//   hipcc --offload-arch=gfx950 -O3 tools/pool_lab/gather_ceiling.hip -o tools/pool_lab/_build/gather_ceiling
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#pragma clang diagnostic ignored "-Wunused-value"

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}

// ids: n_ids int64 (live fraction `live`: the others are the padding id 0 and gather nothing)
template <int SIDE>
__global__ __launch_bounds__(256) void gather_kernel(int64_t n_ids, const int64_t* __restrict__ ids,
                                                     const float* __restrict__ table, int64_t rows,
                                                     float* __restrict__ out, int64_t out_floats,
                                                     int64_t* __restrict__ rows_out, int32_t* __restrict__ seg_out) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_ids; i += stride) {
    const int64_t id = ids[i];
    float4 a = make_float4(0, 0, 0, 0), b = a;
    float c = 0.f;
    int64_t row = 0;
    if (id != 0) {
      row = (int64_t)(mix64((uint64_t)id) % (uint64_t)rows);
      const float* r = table + row * 16;
      a = *reinterpret_cast<const float4*>(r);
      b = *reinterpret_cast<const float4*>(r + 4);
      c = r[8];
    }
    // the pooled output is 0.66 x as many 36-byte rows as there are ids (963 MB for 40.35 M ids): every lane writes 24
    // bytes of it, as a stream
    const int64_t o = i * 6;
    if (o + 6 <= out_floats) {
      float2* p = reinterpret_cast<float2*>(out + o);
      __builtin_nontemporal_store(a.x + b.x, &p[0].x); __builtin_nontemporal_store(a.y + b.y, &p[0].y);
      __builtin_nontemporal_store(a.z + b.z, &p[1].x); __builtin_nontemporal_store(a.w + b.w, &p[1].y);
      __builtin_nontemporal_store(c, &p[2].x);         __builtin_nontemporal_store(c + a.x, &p[2].y);
    }
    if (SIDE) {
      __builtin_nontemporal_store(row, rows_out + i);
      __builtin_nontemporal_store((int32_t)(i >> 1), seg_out + i);
    }
  }
}

// U ids per lane in flight: all ids loaded first, then all rows requested, then the stores
template <int U>
__global__ __launch_bounds__(256) void gather_unrolled_kernel(int64_t n_ids, const int64_t* __restrict__ ids,
                                                              const float* __restrict__ table, int64_t rows,
                                                              float* __restrict__ out, int64_t out_floats,
                                                              int64_t* __restrict__ rows_out, int32_t* __restrict__ seg_out) {
  const int64_t stride = (int64_t)gridDim.x * 256 * U;
  for (int64_t base = ((int64_t)blockIdx.x * 256 * U) + threadIdx.x; base < n_ids; base += stride) {
    int64_t id[U], row[U];
    float4 a[U], b[U];
    float c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const int64_t i = base + u * 256; id[u] = i < n_ids ? ids[i] : 0; }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      a[u] = make_float4(0, 0, 0, 0); b[u] = a[u]; c[u] = 0.f; row[u] = 0;
      if (id[u] != 0) {
        row[u] = (int64_t)(mix64((uint64_t)id[u]) % (uint64_t)rows);
        const float* r = table + row[u] * 16;
        a[u] = *reinterpret_cast<const float4*>(r);
        b[u] = *reinterpret_cast<const float4*>(r + 4);
        c[u] = r[8];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + u * 256;
      if (i >= n_ids) continue;
      const int64_t o = i * 6;
      if (o + 6 <= out_floats) {
        float2* p = reinterpret_cast<float2*>(out + o);
        __builtin_nontemporal_store(a[u].x + b[u].x, &p[0].x); __builtin_nontemporal_store(a[u].y + b[u].y, &p[0].y);
        __builtin_nontemporal_store(a[u].z + b[u].z, &p[1].x); __builtin_nontemporal_store(a[u].w + b[u].w, &p[1].y);
        __builtin_nontemporal_store(c[u], &p[2].x);            __builtin_nontemporal_store(c[u] + a[u].x, &p[2].y);
      }
      __builtin_nontemporal_store(row[u], rows_out + i);
      __builtin_nontemporal_store((int32_t)(i >> 1), seg_out + i);
    }
  }
}

int main(int argc, char** argv) {
  const int64_t n_ids = 40354718, rows = 10000019;
  const double live = 22706897.0 / n_ids;
  int64_t* h = (int64_t*)malloc(n_ids * 8);
  uint64_t s = 88172645463325252ULL;
  int64_t n_live = 0;
  for (int64_t i = 0; i < n_ids; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const bool l = (s >> 11) * (1.0 / 9007199254740992.0) < live;
    h[i] = l ? (int64_t)(s | 1) : 0;
    n_live += l;
  }
  int64_t *ids, *rows_out;
  int32_t* seg;
  float *table, *out;
  const int64_t out_floats = n_ids * 6;
  hipMalloc(&ids, n_ids * 8); hipMalloc(&rows_out, n_ids * 8); hipMalloc(&seg, n_ids * 4);
  hipMalloc(&table, rows * 64); hipMalloc(&out, out_floats * 4);
  hipMemcpy(ids, h, n_ids * 8, hipMemcpyHostToDevice);
  hipMemset(table, 0, rows * 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int side : {1, 0})
    for (int bpc : {4, 8, 16, 32}) {
      const int grid = 256 * bpc;
      float best = 1e9f;
      for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        if (side) hipLaunchKernelGGL(gather_kernel<1>, dim3(grid), dim3(256), 0, 0, n_ids, ids, table, rows, out, out_floats, rows_out, seg);
        else hipLaunchKernelGGL(gather_kernel<0>, dim3(grid), dim3(256), 0, 0, n_ids, ids, table, rows, out, out_floats, rows_out, seg);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
      }
      const double alg = n_ids * 8.0 + n_live * 36.0 + out_floats * 4.0 + (side ? n_ids * 12.0 : 0.0);
      const double act = n_ids * 8.0 + n_live * 64.0 + out_floats * 4.0 + (side ? n_ids * 12.0 : 0.0);
      printf("side outputs %d  blocks/CU %2d : %.3f ms   algorithmic %.2f GB -> %.2f TB/s   with 64-B sectors %.2f GB -> %.2f TB/s\n",
             side, bpc, best, alg / 1e9, alg / best / 1e9, act / 1e9, act / best / 1e9);
    }
  for (int bpc : {2, 4, 8}) {
    const int grid = 256 * bpc;
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(gather_unrolled_kernel<4>, dim3(grid), dim3(256), 0, 0, n_ids, ids, table, rows, out, out_floats, rows_out, seg);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best) best = ms;
    }
    const double act = n_ids * 8.0 + n_live * 64.0 + out_floats * 4.0 + n_ids * 12.0;
    printf("4 ids per lane in flight, side outputs 1, blocks/CU %d : %.3f ms   with 64-B sectors %.2f TB/s\n", bpc, best, act / best / 1e9);
  }
  return 0;
}
