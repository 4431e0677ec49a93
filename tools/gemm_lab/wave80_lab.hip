// csrc/gemm_wave80.h built alone: the single-wave 80 x 80 weight-gradient kernel against the engine's tiled kernels
// (tools/gemm_lab/wave80_lab.py).
#include "gemm_wave80.h"   // (beside this file)

using namespace rec;

extern "C" int lab_dw80_splits(int64_t M, int N, int K) {
  rec_gemm_desc d{};
  d.m = M; d.n = N; d.k = K;
  return dw_wave80_splits(&d, 256);
}
extern "C" int lab_dw80(int64_t M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb, int64_t ldc,
                        int splits, float* partial, float* cpart, int perm, void* stream) {
  rec_gemm_desc d{};
  d.m = M; d.n = N; d.k = K; d.lda = lda; d.ldb = ldb; d.ldc = ldc; d.trans_a = 1; d.trans_b = 0; d.epilogue = 0;
  if (!dw_wave80_ok(&d, A, B)) return 1;
  launch_dw_wave80(&d, splits, A, B, partial, cpart, (hipStream_t)stream, perm != 0);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
