// csrc/gemm_pipe.h built alone: the shipped 256 x 80 tile (8 waves of 32 x 80) against the same block tile on 4 waves of
// 64 x 80 (20 accumulator tiles per wave, 0.3 instead of 0.55 LDS fragment reads per MFMA) — tools/gemm_lab/wide_lab.py.
#include "gemm_pipe.h"
#include "gemm_persist.h"   // (beside this file)
#include <stdlib.h>

using namespace rec;

template <int WM, int OCC, bool TB, int EPI>
static void go(int64_t M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
               const EpiArgs& e, hipStream_t st) {
  constexpr int BM = 256, BN = 80;
  constexpr int A_ELEMS = BM * (kBK + 4), B_ELEMS = kBK * (BN + 4);
  constexpr size_t shmem = 2 * (size_t)(A_ELEMS + B_ELEMS) * sizeof(float);
  const int tiles_n = N / BN;
  const int64_t tiles = (M / BM) * tiles_n;
  hipLaunchKernelGGL((gemm_f32_pipe_kernel<BM, BN, WM, 1, OCC, false, TB, EPI>), dim3((unsigned)tiles, 1), dim3(WM * kWave),
                     shmem, st, M, N, K, A, lda, B, ldb, C, ldc, e, tiles_n, tiles, K, (float*)nullptr, (float*)nullptr, 1);
}

template <int WM, int OCC, bool TB, int EPI>
static void go_persist(int64_t M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                       const EpiArgs& e, hipStream_t st) {
  constexpr int BM = 256, BN = 80;
  constexpr int A_ELEMS = BM * (kBK + 4), B_ELEMS = kBK * (BN + 4);
  constexpr size_t shmem = 2 * (size_t)(A_ELEMS + B_ELEMS) * sizeof(float);
  const int tiles_n = N / BN;
  const int64_t tiles = (M / BM) * tiles_n;
  const char* g = getenv("LAB_GRID");
  const char* sk = getenv("LAB_SKEW");
  int64_t grid = g && *g ? atoi(g) : 256 * OCC;
  if (grid > tiles) grid = tiles - tiles % 8;
  hipLaunchKernelGGL((gemm_f32_persist_kernel<BM, BN, WM, 1, OCC, TB, EPI>), dim3((unsigned)grid), dim3(WM * kWave), shmem, st, M,
                     N, K, A, lda, B, ldb, C, ldc, e, tiles_n, tiles, sk && *sk ? atoi(sk) : 0);
}

// variant: 0 = 8 waves (shipped), 1 = 4 waves x (64 x 80); 3 / 4 = the same two as persistent blocks (gemm_persist.h)
extern "C" int lab_wide(int variant, int64_t M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb,
                        float* C, int64_t ldc, int tb, int epi, const float* bias, const float* aux0, int ld0, void* stream) {
  if (M % 256 || N % 80 || K % 16) return 1;
  const char* nt = getenv("LAB_NT");
  EpiArgs e{bias, aux0, nullptr, nullptr, nullptr, ld0, 0, 0, 0, nt && *nt == '1' ? 1 : 0};
  hipStream_t st = (hipStream_t)stream;
#define GO(WM, OCC)                                                                                        \
  if (!tb && epi == REC_EPI_BIAS_RELU) go<WM, OCC, false, REC_EPI_BIAS_RELU>(M, N, K, A, lda, B, ldb, C, ldc, e, st); \
  else if (tb && epi == REC_EPI_RELU_MASK) go<WM, OCC, true, REC_EPI_RELU_MASK>(M, N, K, A, lda, B, ldb, C, ldc, e, st); \
  else if (tb && epi == REC_EPI_NONE) go<WM, OCC, true, REC_EPI_NONE>(M, N, K, A, lda, B, ldb, C, ldc, e, st);  \
  else return 2;
  if (variant == 0) { GO(8, 2) } else if (variant == 1) { GO(4, 2) }
#undef GO
#define GO(WM, OCC)                                                                                        \
  if (tiles8) return 4;                                                                                    \
  if (!tb && epi == REC_EPI_BIAS_RELU) go_persist<WM, OCC, false, REC_EPI_BIAS_RELU>(M, N, K, A, lda, B, ldb, C, ldc, e, st); \
  else if (tb && epi == REC_EPI_RELU_MASK) go_persist<WM, OCC, true, REC_EPI_RELU_MASK>(M, N, K, A, lda, B, ldb, C, ldc, e, st); \
  else if (tb && epi == REC_EPI_NONE) go_persist<WM, OCC, true, REC_EPI_NONE>(M, N, K, A, lda, B, ldb, C, ldc, e, st);  \
  else return 2;
  const bool tiles8 = ((M / 256) * (N / 80)) % 8 != 0;
  if (variant == 3) { GO(8, 2) } else if (variant == 4) { GO(4, 2) }
#undef GO
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
