// Shared host/device helpers for the recengine kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <type_traits>

#include "recengine.h"

namespace rec {

constexpr int kWave = 64;  // CDNA wavefront
constexpr int kBlock = 256;
constexpr int kNumCU = 256;  // MI355X

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return REC_EHIP;
  }
  return REC_OK;
}

#define REC_REQUIRE(cond, code, ...) \
  do {                               \
    if (!(cond)) {                   \
      rec::set_error(__VA_ARGS__);   \
      return (code);                 \
    }                                \
  } while (0)

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Grid of a persistent kernel: as many blocks as are co-resident on the chip (occupancy query x CU
// count), so that no block waits for a slot — a second, partly filled dispatch round is pure tail.
// Only speed depends on it (no inter-block waits anywhere), so an optimistic answer is harmless.
template <class Kernel>
inline int64_t resident_blocks(Kernel kernel, int block_threads, size_t dyn_lds) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block_threads, dyn_lds) !=
          hipSuccess || per_cu < 1) {
    (void)hipGetLastError();
    per_cu = 2;
  }
  int dev = 0, cus = kNumCU;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      cus = v;
  }
  return (int64_t)per_cu * cus;
}

// Lanes that cooperate on one embedding row: each lane owns VEC consecutive floats.
// D % 4 == 0 && stride % 4 == 0 -> float4 per lane; otherwise one float per lane.
inline int pow2_ceil(int x) {
  int p = 1;
  while (p < x) p <<= 1;
  return p;
}

template <int VEC>
struct Vec;
template <>
struct Vec<1> {
  using T = float;
};
template <>
struct Vec<4> {
  using T = float4;
};

template <int VEC>
__device__ __forceinline__ void vload(float (&r)[VEC], const float* p) {
  if constexpr (VEC == 4) {
    float4 t = *reinterpret_cast<const float4*>(p);
    r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
  } else {
    r[0] = *p;
  }
}
template <int VEC>
__device__ __forceinline__ void vstore(float* p, const float (&r)[VEC]) {
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(r[0], r[1], r[2], r[3]);
  } else {
    *p = r[0];
  }
}
// streaming (non-temporal) variants for data written once and read by a later kernel
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int VEC>
__device__ __forceinline__ void vstore_nt(float* p, const float (&r)[VEC]) {
  if constexpr (VEC == 4) {
    f32x4 t = {r[0], r[1], r[2], r[3]};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(p));
  } else {
    __builtin_nontemporal_store(r[0], p);
  }
}
template <int VEC>
__device__ __forceinline__ void vload_nt(float (&r)[VEC], const float* p) {
  if constexpr (VEC == 4) {
    f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
  } else {
    r[0] = __builtin_nontemporal_load(p);
  }
}

// sum over the LANES lanes of a row group (LANES is a power of two <= 64, groups are aligned)
template <int LANES>
__device__ __forceinline__ float group_sum(float x) {
#pragma unroll
  for (int o = LANES / 2; o > 0; o >>= 1) x += __shfl_xor(x, o, kWave);
  return x;
}

// murmur3 fmix64: the 64-bit finaliser (public domain, Austin Appleby) — the "64-bit mix" of SURVEY §8(d) cfg5
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return k;
}

// feasign -> row of a hashed table with N rows: 0 stays the padding row, everything else lands in [1, N)
__host__ __device__ __forceinline__ int64_t feasign_row(uint64_t f, int64_t N) {
  return f == 0 ? 0 : (int64_t)(1 + mix64(f) % (uint64_t)(N - 1));
}

// Value a PS feature is created with (SparseAdaGradSGDRule::InitValue [EXT]: uniform(-range, range)): a pure
// function of (seed, row, element) so that a row can be born lazily, on the rank that owns it, at its first
// pull OR push, with the same value — no init pass over a table that is mostly never touched.
__host__ __device__ __forceinline__ float ps_init_value(uint64_t seed, int64_t row, int d, float range) {
  const uint64_t h = mix64(seed ^ mix64((uint64_t)row * 0x9E3779B97F4A7C15ull + (uint64_t)(d + 1)));
  const float u = (float)(h >> 40) * (1.0f / 16777216.0f);   // 24 bits -> [0,1)
  return (2.f * u - 1.f) * range;
}

// Dispatch on (emb_dim,row_stride) -> <VEC,LANES>.  F is a generic lambda taking
// std::integral_constant<int,VEC>, std::integral_constant<int,LANES>.
template <class F>
inline int dispatch_row_shape(int D, int stride, F&& f) {
  const bool v4 = (D % 4 == 0) && (stride % 4 == 0);
  const int lanes = pow2_ceil(v4 ? D / 4 : D);
#define REC_CASE(V, L) \
  if (lanes == L) return f(std::integral_constant<int, V>{}, std::integral_constant<int, L>{});
  if (v4) {
    REC_CASE(4, 1) REC_CASE(4, 2) REC_CASE(4, 4) REC_CASE(4, 8) REC_CASE(4, 16) REC_CASE(4, 32)
    REC_CASE(4, 64)
  } else {
    REC_CASE(1, 1) REC_CASE(1, 2) REC_CASE(1, 4) REC_CASE(1, 8) REC_CASE(1, 16) REC_CASE(1, 32)
    REC_CASE(1, 64)
  }
#undef REC_CASE
  set_error("emb_dim %d (stride %d) unsupported: need D<=64 (or D<=256 with D%%4==0)", D, stride);
  return REC_ESHAPE;
}

// Fork / join events of a two-stream step schedule (csrc/deepfm_step.hip, csrc/din_step.hip): kStepEvents events per
// (device, stream, side stream), created on first use and kept for the life of the process (defined in deepfm_step.hip).
constexpr int kStepEvents = 4;
int step_events(void* stream, void* side_stream, hipEvent_t** out);

}  // namespace rec
