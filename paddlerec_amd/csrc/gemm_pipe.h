// The software-pipelined whole-tile f32 GEMM kernel (gemm_f32.hip dispatches to it; tools/gemm_lab builds it alone).
#pragma once

#include "gemm_epi.h"

namespace rec {

// ------------------------------------------------------------------------- interior-only pipelined kernel (candidate)
// The same GEMM for problems made of whole tiles (M % BM == 0, N % BN == 0, K % 16 == 0, aligned operands), with the loop
// scheduled the way tools/gemm_lab/ablate.hip's pipe_kernel measured +6 % over gemm_f32_kernel on the DeepFM forward
// and dX GEMMs (profiles/r02f_gemm_ablate.txt): two tiles of global loads in flight (two named register sets, loop
// unrolled by two, per-thread global and LDS offsets fixed before the loop) and the LDS stores of tile kt+1 pinned
// inside the MFMA stream of tile kt, one ds_write per four MFMAs.  Same LDS images, fragment reads, MFMA order (every k
// once, ascending) and epilogue as gemm_f32_kernel: results are bit-identical.
// The default for the problems it covers since r03 (full GPU suite green with it, bench 2.82 / 2.80 -> 2.79 / 2.78 ms
// in two alternating pairs: profiles/r03_pipe_default_ab.txt); REC_GEMM_PIPE=0 switches it off.  (A first attempt
// that bent gemm_f32_kernel's own loop into this schedule lost the gain — tools/gemm_lab/generic_pipe.patch.)
template <int BM, int BN, int WAVES_M, int WAVES_N, int OCC, bool TA, bool TB, int EPI>
__global__ __launch_bounds__(WAVES_M* WAVES_N* kWave, (OCC * WAVES_M * WAVES_N + 3) / 4) void gemm_f32_pipe_kernel(
    int64_t M, int N, int K, const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
    int64_t ldb, float* __restrict__ C, int64_t ldc, EpiArgs epi, int tiles_n, int64_t tiles_total,
    int k_chunk, float* __restrict__ partial, float* __restrict__ colsum_partial, int splits_in_x) {
  constexpr int NTHR = WAVES_M * WAVES_N * kWave;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MT = WTM / 16, NT = WTN / 16;
  constexpr int LDA_S = TA ? BM + 4 : kBK + 4, LDB_S = BN + 4;
  constexpr int A_ELEMS = TA ? kBK * LDA_S : BM * LDA_S, B_ELEMS = kBK * LDB_S;
  extern __shared__ __attribute__((aligned(16))) float gemm_smem[];
  float* As = gemm_smem;
  float* Bs = gemm_smem + 2 * A_ELEMS;
  int64_t w = blockIdx.x;
  int kz = blockIdx.y;
  if (splits_in_x > 1) {
    const int64_t xcd = w % 8, slot = w / 8;
    kz = (int)(xcd + 8 * (slot / tiles_total));
    w = slot % tiles_total;
  } else {
    const int64_t per = tiles_total / 8;
    if (w < per * 8) w = (w % 8) * per + w / 8;
  }
  const int64_t tm = w / tiles_n;
  const int tn = (int)(w % tiles_n);
  const int64_t m0 = tm * BM;
  const int n0 = tn * BN;
  const int k_begin = kz * k_chunk;
  const int k_end = (k_begin + k_chunk < K) ? k_begin + k_chunk : K;
  const int nkt = (k_end - k_begin) / kBK;          // whole tiles only (the launcher checks K % 16 == 0)
  const int tid = threadIdx.x;
  const int lane = tid % kWave, wave = tid / kWave;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int li = lane & 15, g = lane >> 4;
  f32x4_t acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const bool do_colsum = TA && colsum_partial != nullptr && tm == 0;
  float csum = 0.f;

  // this thread's float4s of the A tile [BM x 16] and the B tile [16 x BN]; a thread past the last one holds a copy of
  // it and stores it again (same value, same address: no branch around the stores)
  constexpr int A_IN = TA ? BM : kBK, B_IN = TB ? kBK : BN;      // contiguous extent of a tile row in memory
  constexpr int A_VECS = BM * kBK / 4, B_VECS = kBK * BN / 4;
  constexpr int PA = (A_VECS + NTHR - 1) / NTHR, PB = (B_VECS + NTHR - 1) / NTHR;
  static_assert(PA <= 4 && PB <= 2, "at most four float4s of A and two of B per thread");
  auto vidx = [&](int j, int nvec) { const int v0 = tid + j * NTHR; return v0 < nvec ? v0 : nvec - 1; };
  const int va0 = vidx(0, A_VECS), va1 = vidx(PA > 1 ? 1 : 0, A_VECS), va2 = vidx(PA > 2 ? 2 : 0, A_VECS),
            va3 = vidx(PA > 3 ? 3 : 0, A_VECS), vb0 = vidx(0, B_VECS), vb1 = vidx(PB - 1, B_VECS);
  auto goff = [&](int v, int inner, int64_t ld) { return (uint32_t)((v / (inner / 4)) * ld + (v % (inner / 4)) * 4); };
  const uint32_t oa0 = goff(va0, A_IN, lda), oa1 = goff(va1, A_IN, lda), oa2 = goff(va2, A_IN, lda),
                 oa3 = goff(va3, A_IN, lda);
  const uint32_t ob0 = goff(vb0, B_IN, ldb), ob1 = goff(vb1, B_IN, ldb);
  auto lds_a = [&](int v) { return (v / (A_IN / 4)) * (A_IN + 4) + (v % (A_IN / 4)) * 4; };
  const int la0 = lds_a(va0), la1 = lds_a(va1), la2 = lds_a(va2), la3 = lds_a(va3);
  const int bo0 = vb0 / (B_IN / 4), bi0 = (vb0 % (B_IN / 4)) * 4, bo1 = vb1 / (B_IN / 4), bi1 = (vb1 % (B_IN / 4)) * 4;
  const float* a_base = TA ? A + (int64_t)k_begin * lda + m0 : A + m0 * lda + k_begin;
  const float* b_base = TB ? B + (int64_t)n0 * ldb + k_begin : B + (int64_t)k_begin * ldb + n0;
  const int64_t a_step = TA ? (int64_t)kBK * lda : kBK, b_step = TB ? kBK : (int64_t)kBK * ldb;
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
      if (!TA) {
        const float4 t = *reinterpret_cast<const float4*>(as + row * LDA_S + g * 4);
        af[a][0] = t.x; af[a][1] = t.y; af[a][2] = t.z; af[a][3] = t.w;
      } else {
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) af[a][s_] = as[(g * 4 + s_) * LDA_S + row];
      }
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
  auto colsum = [&](int cur) {        // bias gradient: column sums of the staged B tile, blocks of the first M-tile
    if constexpr (TA) {
      if (do_colsum && tid < BN) {
        const float* bs = Bs + cur * B_ELEMS;
#pragma unroll
        for (int kk = 0; kk < kBK; ++kk) csum += bs[kk * LDB_S + tid];
      }
    }
  };
  // one tile: fragments of LDS[CUR], the LDS stores of register set S (the NEXT tile) into LDS[CUR ^ 1], the MFMAs,
  // with the schedule pinned to 4 MFMA / 1 ds_write / 4 MFMA / ...
#define REC_PIPE_TILE(CUR, S)                                             \
  frags(CUR);                                                             \
  REC_PIPE_STORE(S, (CUR) ^ 1)                                            \
  mfmas();                                                                \
  _Pragma("unroll") for (int i_ = 0; i_ < kDsWrites; ++i_) {              \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                    \
    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                    \
  }                                                                       \
  colsum(CUR);
  if (nkt > 0) {
    REC_PIPE_LOAD(p0, 0)
    REC_PIPE_STORE(p0, 0)
    if (nkt > 1) REC_PIPE_LOAD(p1, 1)
    __syncthreads();
    int t = 0;
    // invariant at the loop top: LDS[0] = tile t, set 1 = tile t+1 in flight
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
    if (t + 1 < nkt) {                  // 2 or 3 tiles left
      const bool more = t + 2 < nkt;
      if (more) REC_PIPE_LOAD(p0, t + 2)
      frags(0);
      mfmas();
      colsum(0);
      REC_PIPE_STORE(p1, 1)
      __syncthreads();
      ++t;
      if (more) {
        frags(1);
        mfmas();
        colsum(1);
        REC_PIPE_STORE(p0, 0)
        __syncthreads();
        ++t;
      }
    }
    frags(t & 1);                       // the last tile
    mfmas();
    colsum(t & 1);
  }
#undef REC_PIPE_LOAD
#undef REC_PIPE_STORE
#undef REC_PIPE_STORE_B
#undef REC_PIPE_TILE
  if (do_colsum && tid < BN) colsum_partial[(int64_t)kz * N + n0 + tid] = csum;

  // epilogue: as gemm_f32_kernel (every element is inside the matrix here)
  float* out = partial ? partial + (int64_t)kz * M * ldc : C;
  // (dX + ReLU' — the epilogue READS a [M, N] operand while it writes C — runs 6 % faster alone with non-temporal writes,
  // 187 against 199 us on 65536 x 400 x 400; the forward form does not care, the DeepFM step neither: an option)
  const bool nt = epi.nt_store != 0 && !partial;
  float bj[NT];
#pragma unroll
  for (int b = 0; b < NT; ++b) bj[b] = !partial ? load_bias<EPI>(n0 + wn * WTN + b * 16 + li, epi) : 0.f;
#pragma unroll
  for (int a = 0; a < MT; ++a) {
    float x0[4][NT], x1[4][NT];
    if ((EpiUses<EPI>::aux0 || EpiUses<EPI>::aux1) && !partial) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t i = m0 + wm * WTM + a * 16 + g * 4 + r;
#pragma unroll
        for (int b = 0; b < NT; ++b) {
          const int j = n0 + wn * WTN + b * 16 + li;
          x0[r][b] = load_aux0<EPI>(i, j, epi);
          x1[r][b] = load_aux1<EPI>(i, j, epi);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t i = m0 + wm * WTM + a * 16 + g * 4 + r;
#pragma unroll
      for (int b = 0; b < NT; ++b) {
        const int j = n0 + wn * WTN + b * 16 + li;
        const float v = acc[a][b][r];
        const float o = partial ? v : apply_epi<EPI>(v, x0[r][b], x1[r][b], bj[b], i, epi);
        if (nt) __builtin_nontemporal_store(o, &out[i * ldc + j]); else out[i * ldc + j] = o;
      }
    }
  }
}

}  // namespace rec
