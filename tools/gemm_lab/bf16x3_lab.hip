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
