// gemm_f32_direct_kernel — the launch-bound GEMMs of the reference's OWN batch sizes (deepfm/config_bigdata.yaml:23 bs 512,
// din/config.yaml:20 bs 32: 512 x 400 x 432 is 0.18 GFLOP, a microsecond of the chip) in ONE launch.
//
// The tiled kernels fill the chip for such a problem by splitting K and folding the partial tiles in a second launch
// (profiles/r04_b512_trace.txt: 9-11 us + 4 us, nine times per DeepFM step at batch 512 — 28 launches, 0.18 ms).  Here a
// WAVE owns one 16 x 16 tile of C and walks the whole K alone: 512 x 400 is 800 waves, three per CU; nothing is shared
// between waves, so there is no LDS, no barrier and no second launch, and C[i][j] is ONE k-ordered chain of
// v_mfma_f32_16x16x4_f32 (the exact-f32 contract of recengine.h "ARITHMETIC").  Operand fragments go global -> registers:
// in k-block kb (16 contraction indices) lane (r = lane % 16, g = lane / 16) holds k = 16 kb + 4 g + s for the block's four
// MFMAs s = 0 .. 3 — the same permutation of the contraction index on both operands — so that an operand whose memory
// is contiguous along k (A row-major, B given transposed) is ONE float4 per lane and block; the other form is four
// 4-byte loads whose 16 lanes read 64 contiguous bytes.  kDirectPF blocks (8 registers each) are in flight ahead of the
// MFMAs: ~1000 cycles of prefetch for 128 cycles of matrix work per block.  A workgroup is four waves = a 16 x 64 strip
// (the A rows come from L1 for three of them).
#pragma once

#include "gemm_epi.h"

namespace rec {

constexpr int kDirectPF = 8;          // k-blocks in flight per wave

template <bool TA, bool TB, int EPI, bool VEC>
__global__ __launch_bounds__(256) void gemm_f32_direct_kernel(int64_t M, int N, int K, const float* __restrict__ A,
                                                              int64_t lda, const float* __restrict__ B, int64_t ldb,
                                                              float* __restrict__ C, int64_t ldc, EpiArgs epi,
                                                              float* __restrict__ colsum_out, int strips_n) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int64_t m0 = (int64_t)(blockIdx.x / strips_n) * 16;
  const int n0 = (int)(blockIdx.x % strips_n) * 64 + wave * 16;
  if (n0 >= N) return;                                     // a wave behind the last column tile (no barrier in this kernel)
  // the lane's row of A / column of B, clamped into the matrix (rows / columns behind the edge are computed and dropped)
  const int64_t ar = m0 + r < M ? m0 + r : M - 1;
  const int bc = n0 + r < N ? n0 + r : N - 1;
  // memory is contiguous along k for A when !TA, for B when TB
  const float* ap = TA ? A + ar : A + ar * lda;
  const float* bp = TB ? B + (int64_t)bc * ldb : B + bc;
  const int64_t ak = TA ? lda : 1, bk = TB ? 1 : ldb;      // element stride of one step in k

  struct Frag { float a[4], b[4]; bool ok; };
  const int nfull = K / 16, ktail = K % 16;
  // k-block kb of the full ones; a block behind them re-reads the last one and contributes zeros (0 x 0: the loop below
  // is branch-free — a load under a branch makes the compiler drain every load in flight at the join)
  auto load_sel = [&](Frag& f, int kb) {
    const bool ok = kb < nfull;
    const int64_t k = (int64_t)(ok ? kb : nfull - 1) * 16 + 4 * g;
    if (VEC && !TA) {
      const float4 t = *reinterpret_cast<const float4*>(ap + k);
      f.a[0] = t.x; f.a[1] = t.y; f.a[2] = t.z; f.a[3] = t.w;
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) f.a[s] = ap[(k + s) * ak];
    }
    if (VEC && TB) {
      const float4 t = *reinterpret_cast<const float4*>(bp + k);
      f.b[0] = t.x; f.b[1] = t.y; f.b[2] = t.z; f.b[3] = t.w;
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) f.b[s] = bp[(k + s) * bk];
    }
    f.ok = ok;                                             // the zeroing happens where the block is USED (mma): a select
  };                                                       // here would wait for the load it follows
  auto load_tail = [&](Frag& f, int kb) {                  // the last, partial k-block: indices behind K contribute zeros
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int64_t k = (int64_t)kb * 16 + 4 * g + s;
      const int64_t kc = k < K ? k : K - 1;
      const float a = ap[kc * ak], b = bp[kc * bk];
      f.a[s] = k < K ? a : 0.f;
      f.b[s] = k < K ? b : 0.f;
    }
    f.ok = true;
  };
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  float csum = 0.f;                                        // this lane's share of the column sum of B (TA form: the bias gradient)
  auto mma = [&](const Frag& f) {
    float a[4], b[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      a[s] = f.ok ? f.a[s] : 0.f;
      b[s] = f.ok ? f.b[s] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc, 0, 0, 0);
    csum += (b[0] + b[1]) + (b[2] + b[3]);
  };

  if (nfull > 0) {
    Frag f[kDirectPF];
#pragma unroll
    for (int j = 0; j < kDirectPF; ++j) load_sel(f[j], j);
    __builtin_amdgcn_sched_barrier(0);
    for (int kb = 0; kb < nfull; kb += kDirectPF) {
#pragma unroll
      for (int j = 0; j < kDirectPF; ++j) {
        mma(f[j]);
        load_sel(f[j], kb + j + kDirectPF);
        __builtin_amdgcn_sched_barrier(0);                 // the scheduler otherwise sinks a block's loads down to their use,
      }                                                    // kDirectPF steps later: nothing would be in flight
    }
  }
  if (ktail) {
    Frag t;
    load_tail(t, nfull);
    mma(t);
  }

  // ---- C: lane (r, g) holds rows m0 + 4 g + v, column n0 + r
  const int j = n0 + r;
  if (j < N) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int64_t i = m0 + 4 * g + v;
      if (i < M) {
        const float t = acc[v];
        C[i * ldc + j] = apply_epi<EPI>(t, i, j, epi);
        if (EPI == REC_EPI_CROSS && epi.out2) epi.out2[i * epi.ld2 + j] = t + epi.bias[j];
      }
    }
  }
  // ---- column sums of B over k (the strips of the first row tile): the four k-groups of a column meet by shuffles,
  //      in a fixed order
  if (colsum_out && m0 == 0) {
    float s1 = csum + __shfl_xor(csum, 16, 64);
    s1 = s1 + __shfl_xor(s1, 32, 64);
    if (g == 0 && j < N) colsum_out[j] = s1;
  }
}

// M N K below which a GEMM is a few microseconds whichever way it runs (gemm_f32.hip: plan_gemm `tiny`)
inline bool direct_eligible(const rec_gemm_desc* d) {
  static const bool on = [] { const char* v = getenv("REC_GEMM_DIRECT"); return !(v && *v == '0'); }();
  if (!on || d->split_k > 1) return false;                 // an explicit K split is the caller's choice of schedule
  if ((double)d->m * d->n * d->k >= 1.5e8 || d->k > 1024 || d->n <= 4) return false;
  const int64_t strips = ((d->m + 15) / 16) * ((d->n + 63) / 64);
  return strips < (1ll << 31);
}

template <bool TA, bool TB, int EPI>
inline void launch_direct_epi(const rec_gemm_desc* d, const float* A, const float* B, float* C, const EpiArgs& e,
                              float* colsum_out, hipStream_t st) {
  const int strips_n = (d->n + 63) / 64;
  const unsigned grid = (unsigned)(((d->m + 15) / 16) * strips_n);
  // float4 loads along k: 16-B aligned rows of the operand(s) that are contiguous along k
  const bool vec = (TA || (d->lda % 4 == 0 && ((uintptr_t)A) % 16 == 0)) &&
                   (!TB || (d->ldb % 4 == 0 && ((uintptr_t)B) % 16 == 0)) && (!TA || TB);
  if (vec)
    hipLaunchKernelGGL((gemm_f32_direct_kernel<TA, TB, EPI, true>), dim3(grid), dim3(256), 0, st, d->m, d->n, d->k, A,
                       (int64_t)d->lda, B, (int64_t)d->ldb, C, (int64_t)d->ldc, e, colsum_out, strips_n);
  else
    hipLaunchKernelGGL((gemm_f32_direct_kernel<TA, TB, EPI, false>), dim3(grid), dim3(256), 0, st, d->m, d->n, d->k, A,
                       (int64_t)d->lda, B, (int64_t)d->ldb, C, (int64_t)d->ldc, e, colsum_out, strips_n);
}

template <int EPI>
inline void launch_direct_trans(const rec_gemm_desc* d, const float* A, const float* B, float* C, const EpiArgs& e,
                                float* colsum_out, hipStream_t st) {
  if (!d->trans_a && !d->trans_b) launch_direct_epi<false, false, EPI>(d, A, B, C, e, colsum_out, st);
  else if (!d->trans_a && d->trans_b) launch_direct_epi<false, true, EPI>(d, A, B, C, e, colsum_out, st);
  else if (d->trans_a && !d->trans_b) launch_direct_epi<true, false, EPI>(d, A, B, C, e, colsum_out, st);
  else launch_direct_epi<true, true, EPI>(d, A, B, C, e, colsum_out, st);
}

inline bool launch_direct(const rec_gemm_desc* d, const float* A, const float* B, float* C, const EpiArgs& e,
                          float* colsum_out, hipStream_t st) {
  if (!direct_eligible(d)) return false;
  switch (d->epilogue) {
#define REC_DIRECT_CASE(E) case E: launch_direct_trans<E>(d, A, B, C, e, colsum_out, st); return true;
    REC_DIRECT_CASE(REC_EPI_NONE) REC_DIRECT_CASE(REC_EPI_BIAS) REC_DIRECT_CASE(REC_EPI_BIAS_RELU)
    REC_DIRECT_CASE(REC_EPI_RELU_MASK) REC_DIRECT_CASE(REC_EPI_CROSS) REC_DIRECT_CASE(REC_EPI_BIAS_SIGMOID)
    REC_DIRECT_CASE(REC_EPI_BIAS_TANH) REC_DIRECT_CASE(REC_EPI_ADD) REC_DIRECT_CASE(REC_EPI_MOE)
    REC_DIRECT_CASE(REC_EPI_DSIGMOID) REC_DIRECT_CASE(REC_EPI_DTANH)
#undef REC_DIRECT_CASE
  }
  return false;
}

}  // namespace rec
