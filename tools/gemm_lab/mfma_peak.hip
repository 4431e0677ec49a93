// What the matrix pipes sustain on f32 with nothing else going on: every wave issues v_mfma_f32_16x16x4_f32 back to back
// from registers (16 independent accumulators), no memory traffic.  Prints TFLOP/s for a few occupancies and durations —
// the practical ceiling the GEMM kernels are measured against (the 157.3 TF datasheet figure assumes the peak clock).
//   hipcc --offload-arch=gfx950 -O3 tools/gemm_lab/mfma_peak.hip -o tools/gemm_lab/_build/mfma_peak && tools/gemm_lab/_build/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#pragma clang diagnostic ignored "-Wunused-value"
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(int iters, float a0, float b0, int random, float* out) {
  f32x4_t acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // operands: constants (a0 = b0 = 1: nothing toggles) or 16 + 16 per-lane pseudo-random registers in [-0.5, 0.5)
  // (random != 0: the GEMMs' data) — same instruction stream, different switching activity
  float av[NACC], bv[NACC];
  unsigned h = (threadIdx.x + blockIdx.x * 256u) * 2654435761u + 12345u;
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    h = h * 1664525u + 1013904223u;
    av[i] = random ? (float)(h >> 8) * (1.f / 16777216.f) - 0.5f : a0;
    h = h * 1664525u + 1013904223u;
    bv[i] = random ? (float)(h >> 8) * (1.f / 16777216.f) - 0.5f : b0;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[(i + 5) % NACC], acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[0] = s;
}

int main() {
  float* out;
  hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int cus = 256;
  for (int random : {0, 1})
  for (int bpc : {1, 2, 4}) {            // blocks of 4 waves per CU -> 1, 2, 4 waves per SIMD
    for (int iters : {2000, 20000, 200000}) {
      hipLaunchKernelGGL(mfma_loop<16>, dim3(cus * bpc), dim3(256), 0, 0, 100, 1.f, 1.f, random, out);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(mfma_loop<16>, dim3(cus * bpc), dim3(256), 0, 0, iters, 1.f, 1.f, random, out);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      const double flops = (double)cus * bpc * 4 * iters * 16 * (16.0 * 16 * 4 * 2);
      printf("%s operands  waves/SIMD %d  iters %6d  %8.3f ms  %6.1f TFLOP/s\n", random ? "random  " : "constant", bpc, iters, ms, flops / ms / 1e9);
    }
  }
  return 0;
}
