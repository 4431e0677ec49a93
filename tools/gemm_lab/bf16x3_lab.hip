// csrc/gemm_bf16x3.h built alone: lab_x3_split (weight -> plane image), lab_x3_gemm, lab_x3_image_bytes.
#include <stdarg.h>
#include <stdlib.h>
#include "rec_common.h"
#include "gemm_epi.h"
#include "gemm_bf16x3.h"
using namespace rec;
#ifdef REC_X3_LAB_STANDALONE
namespace rec {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
}
#endif
extern "C" size_t lab_x3_image_bytes(int K, int N) { return x3_image_bytes(K, N); }
extern "C" int lab_x3_split(const float* W, int64_t ldw, int K, int N, int trans, char* img, void* stream) {
  return x3_launch_split(W, ldw, K, N, trans, img, (hipStream_t)stream);
}
extern "C" int lab_x3_gemm(int64_t M, int N, int K, const float* A, int64_t lda, const char* img, float* C, int64_t ldc, int epi,
                           const float* bias, const float* aux0, int ld0, void* stream) {
  if (!x3_shape_ok(M, N, K, lda, ldc, A, C)) return -1;
  EpiArgs e{bias, aux0, nullptr, nullptr, nullptr, ld0, 0, 0, 0, 0};
  return x3_launch_gemm(epi, M, N, K, A, lda, img, C, ldc, e, (hipStream_t)stream);
}

// the weight-gradient kernel alone (no reduce): partial tiles into `P` ([slices][kin][ldp]), column sums into cpart
extern "C" int lab_x3_dw(int kin, int nout, int64_t rows, const float* X, int64_t ldx, const float* G, int64_t ldg, float* P,
                         int64_t ldp, float* cpart, int* slices_out, void* stream) {
  X3DwPlan pl;
  if (!x3_dw_plan(kin, nout, rows, 256, &pl)) return -1;
  if (slices_out) *slices_out = pl.slices;
  return x3_launch_dw(pl, kin, nout, rows, X, ldx, G, ldg, P, ldp, cpart, (hipStream_t)stream);
}
