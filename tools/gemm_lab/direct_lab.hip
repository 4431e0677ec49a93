// Lab: csrc/gemm_direct.h alone (seconds to build), one entry point.  hipcc -DREC_DIRECT_PF=<n> ... -shared -o direct_pf<n>.so
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "recengine.h"
#include "gemm_direct.h"
extern "C" int lab_direct(int64_t M, int N, int K, int ta, int tb, int epi, const float* A, const float* B, float* C,
                          const float* bias, const float* aux0, float* colsum, void* stream) {
  rec_gemm_desc d{};
  d.m = M; d.n = N; d.k = K; d.trans_a = ta; d.trans_b = tb; d.epilogue = epi;
  d.lda = ta ? (int)M : K; d.ldb = tb ? K : N; d.ldc = N;
  rec::EpiArgs e{bias, aux0, nullptr, nullptr, nullptr, N, 0, 0, 0, 0};
  return rec::launch_direct(&d, A, B, C, e, colsum, (hipStream_t)stream) ? 0 : -1;
}
