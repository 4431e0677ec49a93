// gemm_f32_direct_kernel — the launch-bound GEMMs of the reference's OWN batch sizes (deepfm/config_bigdata.yaml:23 bs 512,
// din/config.yaml:20 bs 32: 512 x 400 x 432 is 0.18 GFLOP, a microsecond of the chip) in ONE launch.
//
// The tiled kernels fill the chip for such a problem by splitting K and folding the partial tiles in a second launch
// (profiles/r04_b512_trace.txt: 9-11 us + 4 us, nine times per DeepFM step at batch 512 — 28 launches, 0.18 ms).  Here a
// WAVE owns one 16 x 16 tile of C (512 x 400 is 800 waves, three per CU); nothing is shared between tiles, so there is no
// second launch, and C[i][j] is a k-ordered chain of v_mfma_f32_16x16x4_f32 (the exact-f32 contract of recengine.h
// "ARITHMETIC"; with few tiles four such chains over a quarter of K each, added in a fixed order — KS below).  Operand
// fragments go global -> registers: in k-block kb (16 contraction indices) lane (r = lane % 16, g = lane / 16) holds
// k = 16 kb + 4 g + s for the block's four MFMAs s = 0 .. 3 — the same permutation of the contraction index on both
// operands — so that an operand whose memory is contiguous along k (A row-major, B given transposed) is ONE float4 per
// lane and block; the other form is four 4-byte loads whose 16 lanes read 64 contiguous bytes.  kDirectPF blocks (8
// registers each) are in flight ahead of the MFMAs.
// Measured (tools/gemm_lab/direct_lab.py, back-to-back launches): 4.2 us for any problem with K 16 — the launch floor —
// and 8.4 ns per k on top for a wave walking K alone: 6.0 / 6.7 / 7.1 us for the forward / dX / dW problems of a DeepFM
// step at batch 512, 7.3 -> 4.7 us for DIN's 32 x 80 x 512 with the K split over the workgroup's four waves.
#pragma once

#include "gemm_epi.h"

namespace rec {

#ifndef REC_DIRECT_PF
#define REC_DIRECT_PF 8
#endif
constexpr int kDirectPF = REC_DIRECT_PF;          // k-blocks in flight per wave

struct DirectArgs {
  int64_t M; int N, K;
  const float* A; int64_t lda;
  const float* B; int64_t ldb;
  float* C; int64_t ldc;
  EpiArgs epi;
  float* colsum_out;
  int strips_n;                 // KS 1: 64-column strips along N; KS 4: 16-column tiles along N
};

// the work of workgroup `block` of one problem (gemm_f32_direct_kernel: block = blockIdx.x; the pair kernel below deals the
// blocks of one launch to two problems)
template <bool TA, bool TB, int EPI, bool VEC, int KS>
__device__ __forceinline__ void direct_body(const DirectArgs& w, unsigned block) {
  const int64_t M = w.M, lda = w.lda, ldb = w.ldb, ldc = w.ldc;
  const int N = w.N, K = w.K, strips_n = w.strips_n;
  const float* __restrict__ A = w.A;
  const float* __restrict__ B = w.B;
  float* __restrict__ C = w.C;
  float* __restrict__ colsum_out = w.colsum_out;
  const EpiArgs& epi = w.epi;
  // KS 1: the four waves of a workgroup own four column tiles of a 16 x 64 strip, each the whole K.
  // KS 4: the four waves own ONE tile and a quarter of K each; the partial tiles meet in LDS and wave 0 adds them in a
  //       fixed order ((w0 + w1) + (w2 + w3)) and writes C.  The chain of dependent MFMAs a wave walks is what a problem
  //       of few tiles costs (8.4 ns per k measured: 4.3 us of a 7.3 us launch at K 512) — a quarter of it each, side by side.
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int64_t m0 = (int64_t)(block / strips_n) * 16;
  const int n0 = KS == 1 ? (int)(block % strips_n) * 64 + wave * 16 : (int)(block % strips_n) * 16;
  if (KS == 1 && n0 >= N) return;                          // a wave behind the last column tile (no barrier in this form)
  // the lane's row of A / column of B, clamped into the matrix (rows / columns behind the edge are computed and dropped)
  const int64_t ar = m0 + r < M ? m0 + r : M - 1;
  const int bc = n0 + r < N ? n0 + r : N - 1;
  // Addresses: element (row, k) of A is A[row * lda + k] (TA: A[k * lda + row]).  Everything that depends on the k-block
  // is wave-uniform and stays in scalar registers (the block's base pointer); the lane's part — its row / column and its
  // k-group, plus s for the strided form — is four loop-invariant 32-bit offsets per operand (direct_eligible keeps the
  // operands below 2^30 elements).  A load is then one instruction with no vector arithmetic in front of it; with 64-bit
  // per-load address arithmetic the k-block took ~300 cycles for 128 cycles of matrix work.
  const int a_step = TA ? (int)lda : 1, b_step = TB ? 1 : (int)ldb;           // elements per step in k
  int a_off[4], b_off[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    a_off[s] = (int)(TA ? ar : ar * lda) + (4 * g + s) * a_step;
    b_off[s] = (int)(TB ? (int64_t)bc * ldb : bc) + (4 * g + s) * b_step;
  }
  struct Frag { float a[4], b[4]; };
  const int nfull = K / 16, ktail = K % 16;
  auto load_blk = [&](Frag& f, int kb) {                   // k-block kb (wave-uniform; the caller keeps it inside K)
    const float* pa = A + (int64_t)kb * 16 * a_step;
    const float* pb = B + (int64_t)kb * 16 * b_step;
    if (VEC && !TA) {
      const float4 t = *reinterpret_cast<const float4*>(pa + a_off[0]);
      f.a[0] = t.x; f.a[1] = t.y; f.a[2] = t.z; f.a[3] = t.w;
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) f.a[s] = pa[a_off[s]];
    }
    if (VEC && TB) {
      const float4 t = *reinterpret_cast<const float4*>(pb + b_off[0]);
      f.b[0] = t.x; f.b[1] = t.y; f.b[2] = t.z; f.b[3] = t.w;
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) f.b[s] = pb[b_off[s]];
    }
  };
  auto load_tail = [&](Frag& f, int kb) {                  // the last, partial k-block: indices behind K contribute zeros
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int k = kb * 16 + 4 * g + s;
      const int kc = (k < K ? k : K - 1) - (4 * g + s);    // (the lane offsets already hold 4 g + s)
      const float a = A[(int64_t)kc * a_step + a_off[s]], b = B[(int64_t)kc * b_step + b_off[s]];
      f.a[s] = k < K ? a : 0.f;
      f.b[s] = k < K ? b : 0.f;
    }
  };
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  float csum = 0.f;                                        // this lane's share of the column sum of B (TA form: the bias gradient)
#ifdef REC_DIRECT_ACC2      // lab: two accumulator chains (even / odd MFMAs of a block), summed at the end
  f32x4_t acc2 = {0.f, 0.f, 0.f, 0.f};
#endif
  auto mma = [&](const Frag& f) {
#ifdef REC_DIRECT_ACC2
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[0], f.b[0], acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[1], f.b[1], acc2, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[2], f.b[2], acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[3], f.b[3], acc2, 0, 0, 0);
#else
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[s], f.b[s], acc, 0, 0, 0);
#endif
    csum += (f.b[0] + f.b[1]) + (f.b[2] + f.b[3]);
  };

  // this wave's k-blocks [kb0, kb1) and whether the partial block behind them is its own
  const int kq = KS == 1 ? nfull : (nfull + KS - 1) / KS;
  const int kb0 = KS == 1 ? 0 : min(wave * kq, nfull), kb1 = KS == 1 ? nfull : min(kb0 + kq, nfull);
  const bool my_tail = ktail != 0 && (KS == 1 || wave == KS - 1);
  if (kb1 > kb0) {
    // kDirectPF blocks in flight.  Whole groups run branch-free (a load under a branch makes the compiler drain every load
    // in flight at the join); a prefetch behind the last full block re-reads that block and is never used.  The blocks
    // behind the last whole group were fetched by it: their MFMAs sit under wave-uniform branches, which cost nothing.
    Frag f[kDirectPF];
    const int last = kb1 - 1;
#pragma unroll
    for (int j = 0; j < kDirectPF; ++j) load_blk(f[j], kb0 + j < last ? kb0 + j : last);
    __builtin_amdgcn_sched_barrier(0);
    int kb = kb0;
    for (; kb + kDirectPF <= kb1; kb += kDirectPF) {
#pragma unroll
      for (int j = 0; j < kDirectPF; ++j) {
        mma(f[j]);
        const int nx = kb + j + kDirectPF;
        load_blk(f[j], nx < last ? nx : last);
        __builtin_amdgcn_sched_barrier(0);                 // the scheduler otherwise sinks a block's loads down to their use,
      }                                                    // kDirectPF steps later: nothing would be in flight
    }
#pragma unroll
    for (int j = 0; j < kDirectPF - 1; ++j)
      if (kb + j < kb1) mma(f[j]);
  }
  if (my_tail) {
    Frag t;
    load_tail(t, nfull);
    mma(t);
  }

#ifdef REC_DIRECT_ACC2
  acc += acc2;
#endif
  if constexpr (KS > 1) {
    __shared__ float red[KS - 1][5][64];
    if (wave > 0) {
#pragma unroll
      for (int v = 0; v < 4; ++v) red[wave - 1][v][lane] = acc[v];
      red[wave - 1][4][lane] = csum;
    }
    __syncthreads();
    if (wave > 0) return;
    static_assert(KS == 4, "the fold below is written for four partial tiles");
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[v] = (acc[v] + red[0][v][lane]) + (red[1][v][lane] + red[2][v][lane]);
    csum = (csum + red[0][4][lane]) + (red[1][4][lane] + red[2][4][lane]);
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

template <bool TA, bool TB, int EPI, bool VEC, int KS>
__global__ __launch_bounds__(256) void gemm_f32_direct_kernel(DirectArgs w) {
  direct_body<TA, TB, EPI, VEC, KS>(w, blockIdx.x);
}

// Two INDEPENDENT launch-bound problems in one launch (rec_gemm_f32_pair): the weight gradient dW = X^T G (+ the bias
// gradient) and the input gradient dX = G W^T (+ ReLU') of one Linear's backward both consume G and nothing of each
// other; at the reference's batch sizes each is a 6-7 us launch of which 4.2 us is the launch itself.  The first g0
// workgroups work on problem 0, the others on problem 1; every workgroup runs exactly the code of the single launch.
template <int EPI1, int KS0, int KS1>
__global__ __launch_bounds__(256) void gemm_f32_direct_pair_kernel(DirectArgs w0, DirectArgs w1, unsigned g0) {
  if (blockIdx.x < g0) direct_body<true, false, REC_EPI_NONE, false, KS0>(w0, blockIdx.x);
  else direct_body<false, true, EPI1, true, KS1>(w1, blockIdx.x - g0);
}

// M N K below which a GEMM is a few microseconds whichever way it runs (gemm_f32.hip: plan_gemm `tiny`)
inline bool direct_eligible(const rec_gemm_desc* d) {
  static const bool on = [] { const char* v = getenv("REC_GEMM_DIRECT"); return !(v && *v == '0'); }();
  if (!on || d->split_k > 1) return false;                 // an explicit K split is the caller's choice of schedule
  if ((double)d->m * d->n * d->k >= 1.5e8 || d->k > 1024 || d->n <= 4) return false;
  // 32-bit lane offsets: both operands below 2^30 elements, leading dimensions included
  const int64_t a_rows = d->trans_a ? d->k : d->m, b_rows = d->trans_b ? d->n : d->k;
  if (a_rows * (int64_t)d->lda >= (1ll << 30) || b_rows * (int64_t)d->ldb >= (1ll << 30)) return false;
  const int64_t strips = ((d->m + 15) / 16) * ((d->n + 63) / 64);
  return strips < (1ll << 31);
}

// float4 loads along k: 16-B aligned rows of the operand(s) that are contiguous along k
inline bool direct_vec(const rec_gemm_desc* d, const float* A, const float* B) {
  const bool ta = d->trans_a != 0, tb = d->trans_b != 0;
  return (ta || (d->lda % 4 == 0 && ((uintptr_t)A) % 16 == 0)) && (!tb || (d->ldb % 4 == 0 && ((uintptr_t)B) % 16 == 0)) &&
         (!ta || tb);
}
// few tiles and a K worth splitting: four waves per tile (see the kernel); REC_GEMM_DIRECT_KS=1 / 4 forces a form
inline bool direct_split(const rec_gemm_desc* d) {
  static const int ks_env = [] { const char* v = getenv("REC_GEMM_DIRECT_KS"); return v && *v ? atoi(v) : 0; }();
  const int64_t tiles = ((d->m + 15) / 16) * ((d->n + 15) / 16);
  return ks_env == 4 || (ks_env != 1 && tiles <= 256 && d->k >= 128);      // at most one wave per SIMD
}
inline DirectArgs direct_args(const rec_gemm_desc* d, const float* A, const float* B, float* C, const EpiArgs& e,
                              float* colsum_out, bool split, unsigned* grid) {
  const int strips_n = split ? (d->n + 15) / 16 : (d->n + 63) / 64;
  *grid = (unsigned)(((d->m + 15) / 16) * strips_n);
  return DirectArgs{d->m, d->n, d->k, A, (int64_t)d->lda, B, (int64_t)d->ldb, C, (int64_t)d->ldc, e, colsum_out, strips_n};
}

template <bool TA, bool TB, int EPI>
inline void launch_direct_epi(const rec_gemm_desc* d, const float* A, const float* B, float* C, const EpiArgs& e,
                              float* colsum_out, hipStream_t st) {
  const bool vec = direct_vec(d, A, B), split = direct_split(d);
  unsigned grid = 0;
  const DirectArgs w = direct_args(d, A, B, C, e, colsum_out, split, &grid);
#define REC_DIRECT_LAUNCH(VEC_, KS_) \
  hipLaunchKernelGGL((gemm_f32_direct_kernel<TA, TB, EPI, VEC_, KS_>), dim3(grid), dim3(256), 0, st, w)
  if (split) {
    if (vec) REC_DIRECT_LAUNCH(true, 4); else REC_DIRECT_LAUNCH(false, 4);
  } else {
    if (vec) REC_DIRECT_LAUNCH(true, 1); else REC_DIRECT_LAUNCH(false, 1);
  }
#undef REC_DIRECT_LAUNCH
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

// the dW / dX pair of a Linear's backward as ONE launch: d0 = the trans_a form without an epilogue (dW, b_colsum allowed),
// d1 = the trans_b form with no epilogue, the ReLU mask or sigmoid' (dX), float4-loadable; both launch-bound.  -> false: not this
// shape of pair, the caller issues the two GEMMs one after the other.
inline bool launch_direct_pair(const rec_gemm_desc* d0, const float* A0, const float* B0, float* C0, const EpiArgs& e0,
                               float* colsum0, const rec_gemm_desc* d1, const float* A1, const float* B1, float* C1,
                               const EpiArgs& e1, hipStream_t st) {
  static const bool on = [] { const char* v = getenv("REC_GEMM_PAIR"); return !(v && *v == '0'); }();
  if (!on || !direct_eligible(d0) || !direct_eligible(d1)) return false;
  if (!d0->trans_a || d0->trans_b || d0->epilogue != REC_EPI_NONE) return false;
  if (d1->trans_a || !d1->trans_b || (d1->epilogue != REC_EPI_NONE && d1->epilogue != REC_EPI_RELU_MASK &&
                                        d1->epilogue != REC_EPI_DSIGMOID))
    return false;
  if (!direct_vec(d1, A1, B1)) return false;
  const bool s0 = direct_split(d0), s1 = direct_split(d1);
  unsigned g0 = 0, g1 = 0;
  const DirectArgs w0 = direct_args(d0, A0, B0, C0, e0, colsum0, s0, &g0);
  const DirectArgs w1 = direct_args(d1, A1, B1, C1, e1, nullptr, s1, &g1);
  if ((uint64_t)g0 + g1 >= (1ull << 31)) return false;
#define REC_PAIR_LAUNCH(E_, K0_, K1_) \
  hipLaunchKernelGGL((gemm_f32_direct_pair_kernel<E_, K0_, K1_>), dim3(g0 + g1), dim3(256), 0, st, w0, w1, g0)
#define REC_PAIR_KS(E_)                                        \
  if (s0 && s1) REC_PAIR_LAUNCH(E_, 4, 4);                     \
  else if (s0) REC_PAIR_LAUNCH(E_, 4, 1);                      \
  else if (s1) REC_PAIR_LAUNCH(E_, 1, 4);                      \
  else REC_PAIR_LAUNCH(E_, 1, 1)
  if (d1->epilogue == REC_EPI_RELU_MASK) { REC_PAIR_KS(REC_EPI_RELU_MASK); }
  else if (d1->epilogue == REC_EPI_DSIGMOID) { REC_PAIR_KS(REC_EPI_DSIGMOID); }
  else { REC_PAIR_KS(REC_EPI_NONE); }
#undef REC_PAIR_KS
#undef REC_PAIR_LAUNCH
  return true;
}

}  // namespace rec
