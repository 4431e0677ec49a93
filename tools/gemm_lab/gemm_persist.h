// LAB: gemm_f32_pipe_kernel (csrc/gemm_pipe.h) made persistent — a block walks a run of tiles of its XCD and issues the
// first two k-slabs of its NEXT tile before the epilogue of the current one, so that neither the prologue's load latency
// nor (with the co-resident block out of phase) the C stores leave the MFMA pipe idle.  Forward / dX forms only.
// The k order inside a tile is the pipe kernel's: results are bit-identical.
#pragma once

#include "gemm_epi.h"

namespace rec {

template <int BM, int BN, int WAVES_M, int WAVES_N, int OCC, bool TB, int EPI>
__global__ __launch_bounds__(WAVES_M* WAVES_N* kWave, (OCC * WAVES_M * WAVES_N + 3) / 4) void gemm_f32_persist_kernel(
    int64_t M, int N, int K, const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
    float* __restrict__ C, int64_t ldc, EpiArgs epi, int tiles_n, int64_t tiles_total, int skew) {
  constexpr int NTHR = WAVES_M * WAVES_N * kWave;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MT = WTM / 16, NT = WTN / 16;
  constexpr int LDA_S = kBK + 4, LDB_S = BN + 4;
  constexpr int A_ELEMS = BM * LDA_S, B_ELEMS = kBK * LDB_S;
  extern __shared__ __attribute__((aligned(16))) float gemm_smem[];
  float* As = gemm_smem;
  float* Bs = gemm_smem + 2 * A_ELEMS;
  // block b lives on XCD b % 8 and walks that XCD's run of tiles, gridDim.x / 8 apart (the N-tiles of one M-tile stay on
  // one XCD, as in the tiled kernels)
  const int64_t per = tiles_total / 8;                 // (the launcher passes tiles_total % 8 == 0)
  const int64_t xcd = blockIdx.x % 8, jstep = gridDim.x / 8;
  int64_t j = blockIdx.x / 8;
  const int nkt = K / kBK;
  const int tid = threadIdx.x;
  const int lane = tid % kWave, wave = tid / kWave;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int li = lane & 15, g = lane >> 4;
  f32x4_t acc[MT][NT];

  constexpr int A_IN = kBK, B_IN = TB ? kBK : BN;      // contiguous extent of a tile row in memory
  constexpr int A_VECS = BM * kBK / 4, B_VECS = kBK * BN / 4;
  constexpr int PA = (A_VECS + NTHR - 1) / NTHR, PB = (B_VECS + NTHR - 1) / NTHR;
  static_assert(PA <= 4 && PB <= 2, "at most four float4s of A and two of B per thread");
  auto vidx = [&](int q, int nvec) { const int v0 = tid + q * NTHR; return v0 < nvec ? v0 : nvec - 1; };
  const int va0 = vidx(0, A_VECS), va1 = vidx(PA > 1 ? 1 : 0, A_VECS), va2 = vidx(PA > 2 ? 2 : 0, A_VECS),
            va3 = vidx(PA > 3 ? 3 : 0, A_VECS), vb0 = vidx(0, B_VECS), vb1 = vidx(PB - 1, B_VECS);
  auto goff = [&](int v, int inner, int64_t ld) { return (uint32_t)((v / (inner / 4)) * ld + (v % (inner / 4)) * 4); };
  const uint32_t oa0 = goff(va0, A_IN, lda), oa1 = goff(va1, A_IN, lda), oa2 = goff(va2, A_IN, lda),
                 oa3 = goff(va3, A_IN, lda);
  const uint32_t ob0 = goff(vb0, B_IN, ldb), ob1 = goff(vb1, B_IN, ldb);
  auto lds_a = [&](int v) { return (v / (A_IN / 4)) * (A_IN + 4) + (v % (A_IN / 4)) * 4; };
  const int la0 = lds_a(va0), la1 = lds_a(va1), la2 = lds_a(va2), la3 = lds_a(va3);
  const int bo0 = vb0 / (B_IN / 4), bi0 = (vb0 % (B_IN / 4)) * 4, bo1 = vb1 / (B_IN / 4), bi1 = (vb1 % (B_IN / 4)) * 4;
  const int64_t a_step = kBK, b_step = TB ? kBK : (int64_t)kBK * ldb;
  const float* a_base = A;
  const float* b_base = B;
  int64_t m0 = 0;
  int n0 = 0;
  float4 p0a0, p0a1, p0a2, p0a3, p0b0, p0b1, p1a0, p1a1, p1a2, p1a3, p1b0, p1b1;
#define REC_PIPE_LOAD(S, T)                                               \
  {                                                                       \
    const float* ap = a_base + (T) * a_step;                              \
    const float* bp = b_base + (T) * b_step;                              \
    S##a0 = *reinterpret_cast<const float4*>(ap + oa0);                   \
    if (PA > 1) S##a1 = *reinterpret_cast<const float4*>(ap + oa1);       \
    if (PA > 2) S##a2 = *reinterpret_cast<const float4*>(ap + oa2);       \
    if (PA > 3) S##a3 = *reinterpret_cast<const float4*>(ap + oa3);       \
    S##b0 = *reinterpret_cast<const float4*>(bp + ob0);                   \
    if (PB > 1) S##b1 = *reinterpret_cast<const float4*>(bp + ob1);       \
  }
#define REC_PIPE_STORE_B(DST, O, I4, X)                                   \
  if (!TB) {                                                              \
    *reinterpret_cast<float4*>((DST) + (O) * LDB_S + (I4)) = X;           \
  } else {                                                                \
    (DST)[((I4) + 0) * LDB_S + (O)] = X.x;                                \
    (DST)[((I4) + 1) * LDB_S + (O)] = X.y;                                \
    (DST)[((I4) + 2) * LDB_S + (O)] = X.z;                                \
    (DST)[((I4) + 3) * LDB_S + (O)] = X.w;                                \
  }
#define REC_PIPE_STORE(S, BUF)                                            \
  {                                                                       \
    float* ad = As + (BUF) * A_ELEMS;                                     \
    float* bd = Bs + (BUF) * B_ELEMS;                                     \
    *reinterpret_cast<float4*>(ad + la0) = S##a0;                         \
    if (PA > 1) *reinterpret_cast<float4*>(ad + la1) = S##a1;             \
    if (PA > 2) *reinterpret_cast<float4*>(ad + la2) = S##a2;             \
    if (PA > 3) *reinterpret_cast<float4*>(ad + la3) = S##a3;             \
    REC_PIPE_STORE_B(bd, bo0, bi0, S##b0)                                 \
    if (PB > 1) REC_PIPE_STORE_B(bd, bo1, bi1, S##b1)                     \
  }
  constexpr int kDsWrites = PA + (TB ? 4 * PB : PB);
  float af[MT][4], bf[NT][4];
  auto frags = [&](int cur) {
    const float* as = As + cur * A_ELEMS;
    const float* bs = Bs + cur * B_ELEMS;
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      const int row = wm * WTM + a * 16 + li;
      const float4 t = *reinterpret_cast<const float4*>(as + row * LDA_S + g * 4);
      af[a][0] = t.x; af[a][1] = t.y; af[a][2] = t.z; af[a][3] = t.w;
    }
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      const int col = wn * WTN + b * 16 + li;
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) bf[b][s_] = bs[(g * 4 + s_) * LDB_S + col];
    }
  };
  auto mfmas = [&]() {
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a][s_], bf[b][s_], acc[a][b], 0, 0, 0);
  };
  auto set_tile = [&](int64_t jj) {      // tile jj of this XCD's run
    const int64_t w = xcd * per + jj;
    m0 = (w / tiles_n) * BM;
    n0 = (int)(w % tiles_n) * BN;
    a_base = A + m0 * lda;
    b_base = TB ? B + (int64_t)n0 * ldb : B + n0;
  };
#define REC_PIPE_TILE(CUR, S)                                             \
  frags(CUR);                                                             \
  REC_PIPE_STORE(S, (CUR) ^ 1)                                            \
  mfmas();                                                                \
  _Pragma("unroll") for (int i_ = 0; i_ < kDsWrites; ++i_) {              \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                    \
    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                    \
  }
  if (j >= per || nkt < 2) return;            // (lab: K >= 32)
  // the second block of a CU (the upper half of an XCD's blocks) starts late: its epilogues then fall into the other one's k-loops
  if ((int64_t)(blockIdx.x / 8) >= jstep / 2)
    for (int q = 0; q < skew; ++q) __builtin_amdgcn_s_sleep(127);      // 127 x 64 cycles = 3.4 us each
  set_tile(j);
  REC_PIPE_LOAD(p0, 0)
  REC_PIPE_STORE(p0, 0)
  REC_PIPE_LOAD(p1, 1)
  __syncthreads();
  for (;;) {
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    int t = 0;
    // invariant at the loop top: LDS[0] = k-tile t, set 1 = k-tile t + 1 in flight
    for (; t + 3 < nkt; t += 2) {
      REC_PIPE_LOAD(p0, t + 2)
      __builtin_amdgcn_sched_barrier(0);
      REC_PIPE_TILE(0, p1)
      __syncthreads();
      REC_PIPE_LOAD(p1, t + 3)
      __builtin_amdgcn_sched_barrier(0);
      REC_PIPE_TILE(1, p0)
      __syncthreads();
    }
    if (t + 1 < nkt) {                  // 2 or 3 k-tiles left
      const bool more = t + 2 < nkt;
      if (more) REC_PIPE_LOAD(p0, t + 2)
      frags(0);
      mfmas();
      REC_PIPE_STORE(p1, 1)
      __syncthreads();
      ++t;
      if (more) {
        frags(1);
        mfmas();
        REC_PIPE_STORE(p0, 0)
        __syncthreads();
        ++t;
      }
    }
    frags(t & 1);                       // the last k-tile: fragments in registers, LDS free once every wave got here
    // this tile's output coordinates, then the next tile's first two k-slabs on their way before the MFMAs and the stores
    const int64_t cm0 = m0;
    const int cn0 = n0;
    j += jstep;
    const bool next = j < per;
    if (next) {
      set_tile(j);
      REC_PIPE_LOAD(p0, 0)
      REC_PIPE_LOAD(p1, 1)
    }
    __builtin_amdgcn_sched_barrier(0);
    mfmas();
    __syncthreads();                    // every wave has read its fragments of the last k-tile: LDS[0] may be rewritten
    // epilogue (every element is inside the matrix)
    float bj[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b) bj[b] = load_bias<EPI>(cn0 + wn * WTN + b * 16 + li, epi);
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      float x0[4][NT], x1[4][NT];
      if (EpiUses<EPI>::aux0 || EpiUses<EPI>::aux1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t i = cm0 + wm * WTM + a * 16 + g * 4 + r;
#pragma unroll
          for (int b = 0; b < NT; ++b) {
            const int jc = cn0 + wn * WTN + b * 16 + li;
            x0[r][b] = load_aux0<EPI>(i, jc, epi);
            x1[r][b] = load_aux1<EPI>(i, jc, epi);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t i = cm0 + wm * WTM + a * 16 + g * 4 + r;
#pragma unroll
        for (int b = 0; b < NT; ++b) {
          const int jc = cn0 + wn * WTN + b * 16 + li;
          C[i * ldc + jc] = apply_epi<EPI>(acc[a][b][r], x0[r][b], x1[r][b], bj[b], i, epi);
        }
      }
    }
    if (!next) break;
    REC_PIPE_STORE(p0, 0)
    __syncthreads();
  }
#undef REC_PIPE_LOAD
#undef REC_PIPE_STORE
#undef REC_PIPE_STORE_B
#undef REC_PIPE_TILE
}

}  // namespace rec
