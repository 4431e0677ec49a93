// What the chip sustains on v_mfma_f32_16x16x4_f32 with NOTHING else in the way: every wave keeps its A / B fragments
// and NACC independent accumulators in registers and issues MFMAs back to back for `iters` trips (no LDS, no memory).
// The rate over a launch as long as a GEMM of the hot path (hundreds of microseconds, whole chip busy) is the ceiling a
// GEMM kernel can reach on this board under its power management — the number the 157.3 TF datasheet peak should be
// read against.  Operands: zeros, random dense, or ReLU-sparse random (half the A values zero), because the clock
// the chip holds depends on how many datapath bits toggle.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NACC = 8;

extern "C" __global__ __launch_bounds__(256) void mfma_probe_kernel(int iters, int mode, float* out) {
  const int lane = threadIdx.x;
  uint32_t s = 0x9E3779B9u * (blockIdx.x * 256 + lane + 1);
  auto rnd = [&]() {
    s ^= s << 13; s ^= s >> 17; s ^= s << 5;
    return ((float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.25f;
  };
  float a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = mode == 0 ? 0.f : rnd();
    b[i] = mode == 0 ? 0.f : rnd();
    if (mode == 2 && a[i] < 0.f) a[i] = 0.f;      // ReLU-sparse activations
  }
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], b[k], acc[i], 0, 0, 0);
  }
  float t = 0.f;
  for (int i = 0; i < NACC; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (t == 12345.678f) out[0] = t;     // keeps the loop alive
}

// The same MFMA stream fed the way a GEMM feeds it: every trip re-reads its A / B fragments from LDS (one ds_read_b128
// + four ds_read_b32 per 8 MFMAs, about the LDS traffic per MFMA of the engine's 256x80 tile) — does the rate the chip
// sustains change when the LDS is busy too, and does it then depend on the data?
extern "C" __global__ __launch_bounds__(256) void mfma_lds_probe_kernel(int iters, int mode, float* out) {
  __shared__ float lds[4096];
  const int lane = threadIdx.x;
  uint32_t s = 0x9E3779B9u * (blockIdx.x * 256 + lane + 1);
  auto rnd = [&]() {
    s ^= s << 13; s ^= s >> 17; s ^= s << 5;
    return ((float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.25f;
  };
  for (int i = lane; i < 4096; i += 256) {
    float v = mode == 0 ? 0.f : rnd();
    if (mode == 2 && (i & 1) && v < 0.f) v = 0.f;
    lds[i] = v;
  }
  __syncthreads();
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int w = lane & 63;
  for (int it = 0; it < iters; ++it) {
    const int base = (it & 7) * 512;
    const float4 a4 = *reinterpret_cast<const float4*>(&lds[base + w * 4]);   // ds_read_b128
    float b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) b[k] = lds[base + 256 + k * 64 + w];          // 4 x ds_read_b32
    const float a[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], b[k], acc[i], 0, 0, 0);
  }
  float t = 0.f;
  for (int i = 0; i < NACC; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (t == 12345.678f) out[0] = t;
}

// ... and with HBM traffic beside it: every trip each lane also streams 16 bytes from a buffer far larger than the
// caches (1 KB per wave per 32 MFMAs = ~2.3 TB/s over the chip at one wave per SIMD) — the third thing a GEMM keeps busy.
extern "C" __global__ __launch_bounds__(256) void mfma_mem_probe_kernel(int iters, int mode, float* out,
                                                                        const float4* __restrict__ big, long n4) {
  __shared__ float lds[4096];
  const int lane = threadIdx.x;
  uint32_t s = 0x9E3779B9u * (blockIdx.x * 256 + lane + 1);
  auto rnd = [&]() {
    s ^= s << 13; s ^= s >> 17; s ^= s << 5;
    return ((float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.25f;
  };
  for (int i = lane; i < 4096; i += 256) lds[i] = mode == 0 ? 0.f : rnd();
  __syncthreads();
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int w = lane & 63;
  long at = ((long)blockIdx.x * 256 + lane) % n4;
  const long stride = (long)gridDim.x * 256;
  float4 sink = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int it = 0; it < iters; ++it) {
    const float4 x = __builtin_nontemporal_load(&big[at]);
    at += stride;
    if (at >= n4) at -= n4;
    const int base = (it & 7) * 512;
    const float4 a4 = *reinterpret_cast<const float4*>(&lds[base + w * 4]);
    float b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) b[k] = lds[base + 256 + k * 64 + w];
    const float a[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], b[k], acc[i], 0, 0, 0);
    sink.x += x.x; sink.y += x.y; sink.z += x.z; sink.w += x.w;
  }
  float t = sink.x + sink.y + sink.z + sink.w;
  for (int i = 0; i < NACC; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (t == 12345.678f) out[0] = t;
}

// launches `blocks` blocks of 4 waves; returns milliseconds of the launch (HIP events)
extern "C" float mfma_probe(int blocks, int iters, int mode, float* out, void* stream, int with_lds,
                            const void* big, long n4) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipStream_t st = (hipStream_t)stream;
  (void)hipEventRecord(e0, st);
  if (with_lds == 2)
    hipLaunchKernelGGL(mfma_mem_probe_kernel, dim3(blocks), dim3(256), 0, st, iters, mode, out, (const float4*)big, n4);
  else if (with_lds) hipLaunchKernelGGL(mfma_lds_probe_kernel, dim3(blocks), dim3(256), 0, st, iters, mode, out);
  else hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(256), 0, st, iters, mode, out);
  (void)hipEventRecord(e1, st);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return ms;
}
