#include <stdlib.h>
#include "rec_common.h"
#include "gemm_epi.h"
#include "gemm_panel.h"
using namespace rec;
extern "C" int lab_panel(int64_t M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                         int tb, int epi, const float* bias, const float* aux0, int ld0, void* stream) {
  rec_gemm_desc d{};
  d.m = M; d.n = N; d.k = K; d.lda = lda; d.ldb = ldb; d.ldc = ldc; d.trans_a = 0; d.trans_b = tb; d.epilogue = epi; d.split_k = 0;
  EpiArgs e{bias, aux0, nullptr, nullptr, nullptr, ld0, 0, 0, 0};
  return launch_panel(&d, A, B, C, e, (hipStream_t)stream, 256) ? 0 : 1;
}
