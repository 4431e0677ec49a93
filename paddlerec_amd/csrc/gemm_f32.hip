// f32 GEMM on the matrix cores with fused epilogues (gfx950): C = epi(op(A) @ op(B)).
//
// Stands in for the paddle.nn.Linear / paddle.matmul calls of the hot path and their backward:
//   top MLP        /root/reference/models/rank/deepfm/net.py:142-174, dcn_v2/net.py:140-184
//   CrossNetV2     dcn_v2/net.py:214-226   X_{l+1} = X_l + X_0 * (X_l W_l + b_l)  (epilogue CROSS)
//   CrossNetMix    dcn_v2/net.py:278-320   tanh / gate projections
//   DIN MLPs       din/net.py:84-137,160-181
// Exact f32: v_mfma_f32_16x16x4_f32 is bit-for-bit a k-ordered fmaf chain (157 TF peak, no
// xf32/TF32 on gfx950), which is what the 1e-5 logit tolerance of the north star needs.
//
// Tiling (measured on MI355X with tools/gemm_lab, profiles/r02b_gemm_lab.txt): block tile BM x BN, K step 16
// through double-buffered LDS (global -> registers -> LDS, next tile's global loads issued before this tile's
// MFMAs).  A wave owns MT x NT MFMA tiles of 16x16; per K step it reads its fragments as lane (i = lane & 15,
// g = lane >> 4) -> k = 4g..4g+3, so MFMA step s multiplies the k = 4g+s slices — every k exactly once, in a
// fixed order.  LDS images: an operand stored k-contiguous in memory (A row-major) keeps that order, rows padded
// by 4 floats, fragments by ds_read_b128; B always ends up [k][n] (a [N,K] operand is transposed by the store
// pass: +33 % on the dX GEMMs against b128 reads of an [n][k] image), fragments by ds_read_b32.
//   "128x80"   4 waves (4x1), 5 blocks/CU: N = 400, the DeepFM MLP width (no column waste)
//   "256x80"   8 waves (8x1), 2 blocks/CU: narrow N whose tile count does not fill whole rounds of 128x80
//   "256x128"  8 waves (4x2), 2 blocks/CU: wide N, many rows (CrossNet 1560^2: 88 -> 117 TF; slot_dnn layer 0)
//   "128x128"  4 waves (2x2), 4 blocks/CU: general
//   "80x80"    5 waves (5x1), 4 blocks/CU: the weight-gradient GEMMs (M, N = 400: no padding rows; 64 -> 98 TF)
//   "64x80"    4 waves (4x1), 6 blocks/CU
// Blocks are numbered so that the N-tiles of one M-tile run on the same XCD back to back (its A
// tile is fetched from HBM once and re-read from that XCD's L2).
// Split-K (trans_a GEMMs with K = batch): partial tiles go to the workspace and are summed in a
// fixed order by a second kernel (deterministic); the split count fills ONE resident round of the config.
// Tried and dropped (measured on MI355X): a barrier-free variant in which every wave stages its own A/B
// K-slices (76 TF vs 96 TF), static s_setprio staggering of co-resident blocks (no effect), K step 32 (-20 %:
// fewer resident blocks), v_mfma_f32_32x32x2_f32 (equal or slower at every tile), XOR-swizzled A rows (equal).
#include <stdlib.h>

#include "rec_common.h"

#include "gemm_epi.h"
#include "gemm_direct.h"
#include "gemm_glds.h"
#include "gemm_panel.h"
#include "gemm_pipe.h"
#include "gemm_bf16x3.h"
namespace rec {

// --------------------------------------------------------------------------------------- kernel
// OCC = blocks per CU the config is built for (launch bound = waves per SIMD).
template <int BM, int BN, int WAVES_M, int WAVES_N, int OCC, bool TA, bool TB, int EPI>
__global__ __launch_bounds__(WAVES_M* WAVES_N* kWave, (OCC * WAVES_M * WAVES_N + 3) / 4) void gemm_f32_kernel(
    int64_t M, int N, int K, const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
    int64_t ldb, float* __restrict__ C, int64_t ldc, EpiArgs epi, int tiles_n, int64_t tiles_total,
    int k_chunk, bool vec_a, bool vec_b, bool fast, float* __restrict__ partial,
    float* __restrict__ colsum_partial, int splits_in_x) {
  constexpr int NTHR = WAVES_M * WAVES_N * kWave;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MT = WTM / 16, NT = WTN / 16;
  static_assert(WTM % 16 == 0 && WTN % 16 == 0, "wave tile must be a multiple of the 16x16 MFMA tile");
  // A in LDS: [BM][BK+4] (b128 fragment reads) — or k-major [BK][BM+4] when A is stored [K,M];  B: [BK][BN+4]
  constexpr int LDA_S = TA ? BM + 4 : kBK + 4, LDB_S = BN + 4;
  constexpr int A_ELEMS = TA ? kBK * LDA_S : BM * LDA_S, B_ELEMS = kBK * LDB_S;
  extern __shared__ __attribute__((aligned(16))) float gemm_smem[];
  float* As = gemm_smem;                  // [2][A_ELEMS]
  float* Bs = gemm_smem + 2 * A_ELEMS;    // [2][B_ELEMS]

  // XCD-aware numbering (block b runs on XCD b % 8, each XCD has its own L2):
  //  * no split-K: every XCD gets a contiguous range of tiles, so the N-tiles that share an A tile run on the
  //    same XCD back to back;
  //  * split-K (grid.y = 1, grid.x = tiles * splits, splits % 8 == 0): ALL tiles of one K-slice run on one XCD
  //    back to back — the slice of A and B (a few MB) is fetched from HBM once into that L2 instead of once
  //    per XCD that happens to hold one of its tiles.
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

  const int tid = threadIdx.x;
  const int lane = tid % kWave, wave = tid / kWave;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int li = lane & 15, g = lane >> 4;

  TileLoader<BM, kBK, TA, NTHR> la;
  TileLoader<kBK, BN, TB, NTHR> lb;

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // column sums of op(B) over this block's K range (bias gradient when B = dY): the blocks of the
  // first M-tile add up the B tiles they stage anyway
  // (only the trans_a instantiations carry the code: in the others the unused, predicated adds cost 32 VALU and 8 LDS
  // instructions per K step beside 40 MFMAs)
  const bool do_colsum = TA && colsum_partial != nullptr && tm == 0;
  float csum = 0.f;

  const int nkt = (k_end - k_begin + kBK - 1) / kBK;
  // The K range splits into tiles loaded branch-free (every full tile, if the operands allow it: `fast`) and, at
  // most, one checked tail tile.
  const bool tail = !fast || (k_end - k_begin) % kBK != 0;
  const int n_fast = fast ? (tail ? nkt - 1 : nkt) : 0;

  auto load_tile = [&](int kt, auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
    const int k0 = k_begin + kt * kBK;
    la.template load<MODE>(A, lda, m0, k0, M, k_end, vec_a, tid);
    lb.template load<MODE>(B, ldb, k0, n0, k_end, N, vec_b, tid);
  };
  auto store_tile = [&](int buf) {
    la.template store<false>(As + buf * A_ELEMS, tid);
    lb.template store<TB>(Bs + buf * B_ELEMS, tid);
  };
  auto compute = [&](int cur) {
    const float* as = As + cur * A_ELEMS;
    const float* bs = Bs + cur * B_ELEMS;
    float af[MT][4], bf[NT][4];
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
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a][s_], bf[b][s_], acc[a][b], 0, 0, 0);
    if constexpr (TA) {
      if (do_colsum && tid < BN) {
#pragma unroll
        for (int kk = 0; kk < kBK; ++kk) csum += bs[kk * LDB_S + tid];
      }
    }
  };
  using Checked = std::integral_constant<int, 2>;
  // steady state over the branch-free tiles: next tile's loads are issued first and fly under this tile's MFMAs
  auto steady = [&](auto mode_tag) {
    load_tile(0, mode_tag);
    store_tile(0);
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < n_fast; ++kt) {
      load_tile(kt + 1, mode_tag);
      __builtin_amdgcn_sched_barrier(0);   // keep the loads ahead of the MFMAs
      compute(kt & 1);
      // ... and the LDS stores of the loaded tile behind ALL of them: hoisted into the MFMA stream, their
      // s_waitcnt vmcnt stalls the wave on the global loads after a third of the MFMAs (seen in the ISA)
      __builtin_amdgcn_sched_barrier(0);
      store_tile((kt + 1) & 1);
      __syncthreads();
    }
    return kt;
  };
  int kt = 0;
  if (n_fast > 0) {
    const bool interior = m0 + BM <= M && n0 + BN <= N;   // block-uniform
    kt = interior ? steady(std::integral_constant<int, 0>{}) : steady(std::integral_constant<int, 1>{});
  } else if (nkt > 0) {
    load_tile(0, Checked{});
    store_tile(0);
    __syncthreads();
  }
  for (; kt < nkt; ++kt) {                 // last branch-free tile and/or the checked tail tile(s)
    const bool next = kt + 1 < nkt;
    if (next) load_tile(kt + 1, Checked{});
    compute(kt & 1);
    __builtin_amdgcn_sched_barrier(0);
    if (next) store_tile((kt + 1) & 1);
    __syncthreads();
  }
  if (do_colsum && tid < BN && n0 + tid < N)
    colsum_partial[(int64_t)kz * N + n0 + tid] = csum;

  // C/D layout of v_mfma_f32_16x16x4_f32: col = lane & 15, row = (lane >> 4) * 4 + reg
  float* out = partial ? partial + (int64_t)kz * M * ldc : C;
  float bj[NT];                     // this lane's NT bias values, read once ahead of every store of the tile
#pragma unroll
  for (int b = 0; b < NT; ++b) {
    const int j = n0 + wn * WTN + b * 16 + li;
    bj[b] = (!partial && j < N) ? load_bias<EPI>(j, epi) : 0.f;
  }
  // one M-tile row block (16 rows x the wave's columns) at a time: all of its aux loads first, then its stores
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
          const bool ok = i < M && j < N;
          x0[r][b] = ok ? load_aux0<EPI>(i, j, epi) : 0.f;
          x1[r][b] = ok ? load_aux1<EPI>(i, j, epi) : 0.f;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t i = m0 + wm * WTM + a * 16 + g * 4 + r;
      if (i < M) {
#pragma unroll
        for (int b = 0; b < NT; ++b) {
          const int j = n0 + wn * WTN + b * 16 + li;
          if (j < N) {
            const float v = acc[a][b][r];
            out[i * ldc + j] = partial ? v : apply_epi<EPI>(v, x0[r][b], x1[r][b], bj[b], i, epi);
            if (EPI == REC_EPI_CROSS && !partial && epi.out2) epi.out2[i * epi.ld2 + j] = v + bj[b];
          }
        }
      }
    }
  }
}

// one column of the [splits][N] partial column sums, ascending z with eight loads in flight and a fixed fold order
__device__ __forceinline__ float colsum_fold(int N, int splits, const float* __restrict__ partial, int j) {
  float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int z = 0;
  for (; z + 8 <= splits; z += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) a8[u] += partial[(int64_t)(z + u) * N + j];
  }
  for (; z < splits; ++z) a8[z & 7] += partial[(int64_t)z * N + j];
  return ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
}

// C = epi(sum_z partial[z]) in ascending z (fixed order).  The LAST colsum_blocks blocks of the grid fold the partial
// column sums of the same GEMM instead (the bias gradient of dW = X^T dY): one launch behind a split-K weight gradient
// instead of two (round 4; the arithmetic of both folds is unchanged).
template <int EPI>
__global__ __launch_bounds__(kBlock) void splitk_reduce_kernel(int64_t M, int N, int64_t ldc,
                                                               int splits,
                                                               const float* __restrict__ partial,
                                                               float* __restrict__ C, EpiArgs epi,
                                                               const float* __restrict__ cpartial,
                                                               float* __restrict__ colsum_out, int colsum_blocks) {
  const int rblocks = (int)gridDim.x - colsum_blocks;
  if ((int)blockIdx.x >= rblocks) {
    const int j = ((int)blockIdx.x - rblocks) * kBlock + threadIdx.x;
    if (j < N) colsum_out[j] = colsum_fold(N, splits, cpartial, j);
    return;
  }
  const int64_t total = M * N;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total;
       e += (int64_t)rblocks * kBlock) {
    const int64_t i = e / N;
    const int j = (int)(e % N);
    const float* pp = partial + i * ldc + j;
    const int64_t zs = M * ldc;
    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // 8 loads in flight; fixed fold order below
    int z = 0;
    for (; z + 8 <= splits; z += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a8[u] += pp[(int64_t)(z + u) * zs];
    }
    for (; z < splits; ++z) a8[z & 7] += pp[(int64_t)z * zs];
    const float t = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
    C[i * ldc + j] = apply_epi<EPI>(t, i, j, epi);
    if (EPI == REC_EPI_CROSS && epi.out2) epi.out2[i * epi.ld2 + j] = t + epi.bias[j];
  }
}

__global__ __launch_bounds__(kBlock) void colsum_reduce_kernel(int N, int splits,
                                                               const float* __restrict__ partial,
                                                               float* __restrict__ out) {
  const int j = blockIdx.x * kBlock + threadIdx.x;
  if (j < N) out[j] = colsum_fold(N, splits, partial, j);
}

// ------------------------------------------------------------------------------- skinny shapes (N <= 4)
// The last Linear of a CTR tower has one output (deepfm/net.py:150 sizes[-1] = 1): as a tiled GEMM it would
// use 1/80 of every MFMA; it is a streaming HBM-bound pass instead.
constexpr int kSkinnyN = 4;

// C[M,N] = epi(A[M,K] @ op(B)):  one wave per row, lanes stride K with float4 loads, xor-shuffle fold.
template <int EPI>
__global__ __launch_bounds__(kBlock) void gemv_rows_kernel(int64_t M, int N, int K, const float* __restrict__ A,
                                                           int64_t lda, const float* __restrict__ B, int64_t ldb,
                                                           bool trans_b, float* __restrict__ C, int64_t ldc,
                                                           EpiArgs epi, bool vec_a) {
  extern __shared__ __attribute__((aligned(16))) float bs[];   // [N][K] (k contiguous)
  for (int i = threadIdx.x; i < N * K; i += kBlock) {
    const int n = i / K, k = i % K;
    bs[i] = trans_b ? B[(int64_t)n * ldb + k] : B[(int64_t)k * ldb + n];
  }
  __syncthreads();
  const int lane = threadIdx.x % kWave;
  const int64_t wpb = kBlock / kWave;
  for (int64_t i = (int64_t)blockIdx.x * wpb + threadIdx.x / kWave; i < M; i += (int64_t)gridDim.x * wpb) {
    float acc[kSkinnyN] = {0.f, 0.f, 0.f, 0.f};
    const float* a = A + i * lda;
    if (vec_a) {
      for (int k = lane * 4; k + 3 < K; k += kWave * 4) {
        const float4 x = *reinterpret_cast<const float4*>(a + k);
#pragma unroll
        for (int n = 0; n < kSkinnyN; ++n)
          if (n < N) {
            const float4 w = *reinterpret_cast<const float4*>(bs + n * K + k);
            acc[n] += x.x * w.x + x.y * w.y + x.z * w.z + x.w * w.w;
          }
      }
      for (int k = (K / 4) * 4 + lane; k < K; k += kWave)
#pragma unroll
        for (int n = 0; n < kSkinnyN; ++n)
          if (n < N) acc[n] += a[k] * bs[n * K + k];
    } else {
      for (int k = lane; k < K; k += kWave)
#pragma unroll
        for (int n = 0; n < kSkinnyN; ++n)
          if (n < N) acc[n] += a[k] * bs[n * K + k];
    }
#pragma unroll
    for (int n = 0; n < kSkinnyN; ++n)
#pragma unroll
      for (int o = kWave / 2; o > 0; o >>= 1) acc[n] += __shfl_xor(acc[n], o, kWave);
    if (lane == 0)
#pragma unroll
      for (int n = 0; n < kSkinnyN; ++n)
        if (n < N) {
          C[i * ldc + n] = apply_epi<EPI>(acc[n], i, n, epi);
          if (EPI == REC_EPI_CROSS && epi.out2) epi.out2[i * epi.ld2 + n] = acc[n] + epi.bias[n];
        }
  }
}

// partial[z][m][n] = sum over the K-chunk z of A[k,m] * B[k,n]   (A stored [K,M]: dW of a 1-output Linear),
// colsum_partial[z][n] = sum_k B[k,n].  Thread t owns columns m = t, t+256, ...; reduced by splitk_reduce.
constexpr int kSkinnyKC = 128;   // k rows per block
constexpr int kSkinnyMT = 4;     // columns per thread (M <= 1024)
__global__ __launch_bounds__(kBlock) void skinny_dw_kernel(int M, int N, int64_t K, const float* __restrict__ A,
                                                           int64_t lda, const float* __restrict__ B, int64_t ldb,
                                                           int64_t ldc, float* __restrict__ partial,
                                                           float* __restrict__ colsum_partial) {
  __shared__ float bs[kSkinnyKC * kSkinnyN];
  const int64_t k0 = (int64_t)blockIdx.x * kSkinnyKC;
  const int kc = (int)((K - k0 < kSkinnyKC) ? K - k0 : kSkinnyKC);
  for (int i = threadIdx.x; i < kc * N; i += kBlock) bs[(i / N) * kSkinnyN + i % N] = B[(k0 + i / N) * ldb + i % N];
  __syncthreads();
  float acc[kSkinnyMT][kSkinnyN];
#pragma unroll
  for (int c = 0; c < kSkinnyMT; ++c)
#pragma unroll
    for (int n = 0; n < kSkinnyN; ++n) acc[c][n] = 0.f;
  // 8 rows of A in flight per thread before any FMA (the loop is pure streaming: latency, not ALU, sets it)
  for (int kb = 0; kb < kc; kb += 8) {
    float x[8][kSkinnyMT];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float* a = A + (k0 + kb + u) * lda;
#pragma unroll
      for (int c = 0; c < kSkinnyMT; ++c) {
        const int m = threadIdx.x + c * kBlock;
        x[u][c] = (kb + u < kc && m < M) ? a[m] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < kSkinnyMT; ++c)
#pragma unroll
        for (int n = 0; n < kSkinnyN; ++n)
          if (n < N) acc[c][n] += x[u][c] * ((kb + u < kc) ? bs[(kb + u) * kSkinnyN + n] : 0.f);
  }
  float* out = partial + (int64_t)blockIdx.x * M * ldc;
#pragma unroll
  for (int c = 0; c < kSkinnyMT; ++c) {
    const int m = threadIdx.x + c * kBlock;
    if (m < M)
#pragma unroll
      for (int n = 0; n < kSkinnyN; ++n)
        if (n < N) out[(int64_t)m * ldc + n] = acc[c][n];
  }
  if (colsum_partial && (int)threadIdx.x < N) {
    float t = 0.f;
    for (int k = 0; k < kc; ++k) t += bs[k * kSkinnyN + threadIdx.x];
    colsum_partial[(int64_t)blockIdx.x * N + threadIdx.x] = t;
  }
}

// The same partials for a NARROW A (M <= 128: the 40-wide head of DIN's top MLP): with one thread per column only M of the
// 256 lanes would read, 128 dependent rows each (54 us at K 4096 x M 40).  Here 256 / M thread groups share the block's
// k rows (group g takes rows g, g + G, ...), and group 0 folds the groups' sums through LDS in group order
// (fixed order: deterministic).
__global__ __launch_bounds__(kBlock) void skinny_dw_narrow_kernel(int M, int N, int64_t K, const float* __restrict__ A,
                                                                  int64_t lda, const float* __restrict__ B, int64_t ldb,
                                                                  int64_t ldc, float* __restrict__ partial,
                                                                  float* __restrict__ colsum_partial) {
  __shared__ float bs[kSkinnyKC * kSkinnyN];
  __shared__ float red[kBlock * kSkinnyN];
  const int64_t k0 = (int64_t)blockIdx.x * kSkinnyKC;
  const int kc = (int)((K - k0 < kSkinnyKC) ? K - k0 : kSkinnyKC);
  for (int i = threadIdx.x; i < kc * N; i += kBlock) bs[(i / N) * kSkinnyN + i % N] = B[(k0 + i / N) * ldb + i % N];
  __syncthreads();
  const int G = kBlock / M, g = (int)threadIdx.x / M, m = (int)threadIdx.x % M;
  float acc[kSkinnyN] = {0.f, 0.f, 0.f, 0.f};
  if (g < G) {
    for (int kb = g; kb < kc; kb += 8 * G) {
      float x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = kb + u * G;
        x[u] = k < kc ? A[(k0 + k) * lda + m] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = kb + u * G;
#pragma unroll
        for (int n = 0; n < kSkinnyN; ++n)
          if (n < N && k < kc) acc[n] += x[u] * bs[k * kSkinnyN + n];
      }
    }
#pragma unroll
    for (int n = 0; n < kSkinnyN; ++n) red[threadIdx.x * kSkinnyN + n] = acc[n];
  }
  __syncthreads();
  if (g == 0) {
    for (int o = 1; o < G; ++o)
#pragma unroll
      for (int n = 0; n < kSkinnyN; ++n) acc[n] += red[(o * M + m) * kSkinnyN + n];
    float* out = partial + (int64_t)blockIdx.x * M * ldc;
#pragma unroll
    for (int n = 0; n < kSkinnyN; ++n)
      if (n < N) out[(int64_t)m * ldc + n] = acc[n];
  }
  if (colsum_partial && (int)threadIdx.x < N) {
    float t = 0.f;
    for (int k = 0; k < kc; ++k) t += bs[k * kSkinnyN + threadIdx.x];
    colsum_partial[(int64_t)blockIdx.x * N + threadIdx.x] = t;
  }
}

// column sums of G [M,N] (bias gradients): deterministic two-level reduction
constexpr int kColsumRows = 64;
__global__ __launch_bounds__(kBlock) void colsum_partial_kernel(int64_t M, int N, int64_t ld,
                                                                const float* __restrict__ G,
                                                                float* __restrict__ partial) {
  // block handles kColsumRows rows x all columns; thread t owns columns t, t+256, ...
  const int64_t r0 = (int64_t)blockIdx.x * kColsumRows;
  const int64_t r1 = r0 + kColsumRows < M ? r0 + kColsumRows : M;
  for (int j = threadIdx.x; j < N; j += kBlock) {
    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // 8 loads in flight, fixed fold order
    int64_t r = r0;
    for (; r + 8 <= r1; r += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a8[u] += G[(r + u) * ld + j];
    }
    for (; r < r1; ++r) a8[(r - r0) & 7] += G[r * ld + j];
    partial[(int64_t)blockIdx.x * N + j] = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
  }
}
// column j of the [nblk][N] partial rows, blocks in ascending order: a block owns 16 columns, thread (c, q) the rows
// q, q + 16, ... on four interleaved chains (loads in flight; one thread walking all 1024 partial rows of a B = 65536
// batch took 37 us), then the 16 row groups fold in ascending q.
__global__ __launch_bounds__(kBlock) void colsum_final_kernel(int nblk, int N,
                                                              const float* __restrict__ partial,
                                                              float* __restrict__ out) {
  __shared__ float red[16][17];
  const int c = threadIdx.x & 15, q = threadIdx.x >> 4;
  const int j = blockIdx.x * 16 + c;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  if (j < N) {
    int r = q;
    for (; r + 48 < nblk; r += 64) {
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] += partial[(int64_t)(r + 16 * u) * N + j];
    }
    for (int u = 0; r < nblk; r += 16, ++u) a[u] += partial[(int64_t)r * N + j];
  }
  red[q][c] = (a[0] + a[1]) + (a[2] + a[3]);
  __syncthreads();
  if (q == 0 && j < N) {
    float t = red[0][c];
#pragma unroll
    for (int k = 1; k < 16; ++k) t += red[k][c];
    out[j] = t;
  }
}

// ------------------------------------------------------------------------------- backward of a one-logit head
// The last Linear of the CTR MLPs has ONE output (deepfm/net.py:169-174: 400 -> 1).  Its backward as two GEMMs is a
// K = 1 "GEMM" for dX (a 256x80-tile MFMA kernel doing an outer product: 96 us for 210 MB in the DeepFM step) plus a
// skinny dW path of three launches (skinny_dw + split-K reduce + column-sum reduce: 220 us of kernel time) — both
// stream the same [B, N] activation.  One HBM-bound pass instead:
//   dx[b, j] = act[b, j] > 0 ? dz[b] * w[j] : 0          (ReLU' fused: the layer in front ends in a ReLU)
//   dW[j]    = sum_b act[b, j] * dz[b],   db = sum_b dz[b]
// Thread t of a block owns float4 column chunk t % 128 (N <= 512) and row parity t / 128; per-thread partial sums over
// its rows in ascending order, the two parities folded in LDS, one partial row [N + 1] per block, folded across blocks
// by colsum_final_kernel in a fixed order: deterministic.
constexpr int kHeadChunks = 128;
__global__ __launch_bounds__(kBlock) void mlp_head_bwd_kernel(int64_t B, int N, const float* __restrict__ act,
                                                              int64_t ld_act, const float* __restrict__ dz,
                                                              const float* __restrict__ w, float* __restrict__ dx,
                                                              int64_t ld_dx, int relu, float* __restrict__ partial) {
  __shared__ float red[kHeadChunks * 4 + 1];
  const int c = threadIdx.x % kHeadChunks, par = threadIdx.x / kHeadChunks;      // kBlock = 2 * kHeadChunks
  const int j0 = c * 4;
  const bool colok = j0 < N;
  const int64_t rows_per = (B + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per;
  const int64_t r1 = r0 + rows_per < B ? r0 + rows_per : B;
  float wv[4] = {0.f, 0.f, 0.f, 0.f}, acc[4] = {0.f, 0.f, 0.f, 0.f};
  float accb = 0.f;
  if (colok) vload<4>(wv, w + j0);
  for (int64_t b = r0 + par; b < r1; b += 2) {
    const float g = dz[b];
    if (c == 0) accb += g;
    if (colok) {
      float a[4], o[4];
      vload_nt<4>(a, act + b * ld_act + j0);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        acc[v] += a[v] * g;
        o[v] = (!relu || a[v] > 0.f) ? g * wv[v] : 0.f;
      }
      vstore<4>(dx + b * ld_dx + j0, o);
    }
  }
  // fold the two row parities (even rows first), then one partial row per block
  if (par == 1) {
#pragma unroll
    for (int v = 0; v < 4; ++v) red[c * 4 + v] = acc[v];
    if (c == 0) red[kHeadChunks * 4] = accb;
  }
  __syncthreads();
  if (par == 0) {
    float* prow = partial + (int64_t)blockIdx.x * (N + 1);
    if (colok) {
#pragma unroll
      for (int v = 0; v < 4; ++v) prow[j0 + v] = acc[v] + red[c * 4 + v];
    }
    if (c == 0) prow[N] = accb + red[kHeadChunks * 4];
  }
}

// One wave per column folds the per-block partial rows of mlp_head_bwd_kernel: lane l sums rows l, l+64, ... in ascending
// order, then a fixed xor-shuffle tree — deterministic; column N goes to db, the others to dw.
__global__ __launch_bounds__(kBlock) void mlp_head_fold_kernel(int nblk, int N, const float* __restrict__ partial,
                                                               float* __restrict__ dw, float* __restrict__ db) {
  const int j = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
  const int lane = threadIdx.x % kWave;
  if (j > N) return;
  float s = 0.f;
  for (int r = lane; r < nblk; r += kWave) s += partial[(int64_t)r * (N + 1) + j];
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, kWave);
  if (lane == 0) (j < N ? dw[j] : db[0]) = s;
}

// ------------------------------------------------------------------------------- config selection
enum GemmCfg { CFG_128x80 = 0, CFG_256x80, CFG_256x128, CFG_128x128, CFG_80x80, CFG_64x80, CFG_128x80_O4, CFG_144x80,
               CFG_COUNT };
struct CfgInfo {
  int bm, bn, threads, occ;
  float eff;   // relative throughput of a full tile of this config (tools/gemm_lab measurements)
};
static const CfgInfo kCfg[CFG_COUNT] = {
    {128, 80, 256, 5, 1.00f},  {256, 80, 512, 2, 0.97f}, {256, 128, 512, 2, 1.08f},
    {128, 128, 256, 4, 1.00f}, {80, 80, 320, 4, 0.98f},  {64, 80, 256, 6, 0.97f},
    {128, 80, 256, 4, 1.00f},   // 128x80 built for 4 blocks/CU (128 VGPRs: no spills) — experiment
    // 9 waves x (16 x 80): the 80x80 weight-gradient tile with 144 rows — DeepFM's layer-0 weight gradient has M = 432 =
    // 27 x 16 = 3 x 144 rows (26 embedding fields + the dense row, x 16), which 80-row tiles pad to 480 and keep off the
    // whole-tile pipe kernel (64 TF, the long pole of the step's tail)
    {144, 80, 576, 2, 0.98f},
};

struct GemmPlan {
  int cfg;
  int tiles_n;
  int64_t tiles_m, tiles_total;
  int splits, k_chunk;
};

static bool skinny_rows(const rec_gemm_desc* d) {   // C = A @ B with N <= 4, A row-major
  return d->n <= kSkinnyN && !d->trans_a && (size_t)d->n * d->k * sizeof(float) <= 48 * 1024;
}
static bool skinny_dw(const rec_gemm_desc* d) {     // C[M<=1024, N<=4] = A^T B over a long K
  return d->n <= kSkinnyN && d->trans_a && !d->trans_b && d->m <= kSkinnyMT * kBlock && d->k >= 1024;
}

static int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Config choice from the measurements of profiles/r02b_gemm_*.txt (MI355X, f32):
//   few output rows (weight gradients, K = batch): 80x80 when it pads at least a fifth less than 128x128 — M, N = 400
//   tile exactly, 93 vs 70 TF — and 128x128 otherwise: a full 128x128 tile does ~1.3x the work per cycle of an
//   80x80 one (64x64 against 16x80 per wave), 103 vs 72 TF on the 1560^2 weight gradients of DCN-v2
//   (profiles/r02f_dw_cfg_probe.txt);
//   many rows: the tile width that wastes fewer columns (80 vs 128), on 256-row / 8-wave blocks when there are
//   enough rows (256x80: 100-105 TF on N = 400/432 and 121 TF on N = 1560, 256x128: 123-131 TF on N = 512 / 1560 /
//   4096); the 4-wave versions (128x80, 128x128) for short matrices.
static GemmPlan plan_gemm(const rec_gemm_desc* d, int num_cus = kNumCU) {
  GemmPlan p;
  const int nkt = (d->k + kBK - 1) / kBK;
  int best;
  if (d->m < 2048) {
    const int64_t a80 = ceil_div64(d->m, 80) * 80 * ceil_div64(d->n, 80) * 80;
    const int64_t a128 = ceil_div64(d->m, 128) * 128 * ceil_div64(d->n, 128) * 128;
    best = a80 * 5 <= a128 * 4 ? CFG_80x80 : CFG_128x128;
    static const bool cfg144 = [] { const char* v = getenv("REC_GEMM_144"); return !(v && *v == '0'); }();
    if (cfg144 && best == CFG_80x80 && d->trans_a && !d->trans_b && d->epilogue == REC_EPI_NONE && d->m % 144 == 0 &&
        d->m % 80 != 0 && d->n % 80 == 0)
      best = CFG_144x80;
    // few rows AND a short K (the reference's own batch sizes: 512 x 400 x 432): a 128x128 tile gives a wave 64 MFMAs
    // per k-step and the problem 16 blocks — 47 us on 16 of 256 CUs.  64x80 tiles (20 MFMAs per wave and k-step, 2.5x
    // the blocks) and a K split fill the chip: ~10 us incl. the reduce (profiles/r03_small_batch.txt)
    if (!d->trans_a && d->m <= 1024 && d->k <= 4096 && best == CFG_128x128 &&
        ceil_div64(d->m, 64) * 64 * ceil_div64(d->n, 80) * 80 * 4 <= a128 * 5)
      best = CFG_64x80;
  } else {
    const int64_t w80 = ceil_div64(d->n, 80) * 80 - d->n, w128 = ceil_div64(d->n, 128) * 128 - d->n;
    const bool tall = d->m >= 8192;
    best = w80 < w128 ? (tall ? CFG_256x80 : CFG_128x80) : (tall ? CFG_256x128 : CFG_128x128);
  }
  {   // experiments: REC_GEMM_FORCE_CFG=<index> overrides the choice (tools/gemm_lab)
    static const int forced = [] { const char* v = getenv("REC_GEMM_FORCE_CFG"); return v && *v ? atoi(v) : -1; }();
    if (forced >= 0 && forced < CFG_COUNT && forced != CFG_144x80) best = forced;
  }
  p.cfg = best;
  const CfgInfo& f = kCfg[best];
  p.tiles_n = (int)ceil_div64(d->n, f.bn);
  p.tiles_m = ceil_div64(d->m, f.bm);
  p.tiles_total = p.tiles_m * p.tiles_n;
  int splits = d->split_k;
  if (splits <= 0) {  // auto: split K when the output alone cannot fill the chip
    splits = 1;
    const int64_t capacity = (int64_t)num_cus * f.occ;
    // a GEMM this small runs for a few microseconds split or not, and a split adds a reduce LAUNCH (4-5 us on the
    // GPU, ~10 us of host time in a launch-bound step: the reference's own batch size of 512 splits every MLP GEMM)
    // round 3: they split again, with shorter slices — the reduce is issued by this same C call (~3 us of host time,
    // not a python round trip) and a block that walks 27 dependent k-steps alone on its CU costs ten times that
    const bool tiny = (double)d->m * d->n * d->k < 1.5e8;
    static const bool tiny_split = [] { const char* v = getenv("REC_GEMM_TINY_SPLIT"); return !(v && *v == '0'); }();
    if (p.tiles_total * 2 <= capacity && (!tiny || (tiny_split && nkt >= 8))) {
      // as many splits as still fit in ONE resident round (one block more would double the time)
      int64_t want = capacity / p.tiles_total;
      const int min_kt = tiny ? 4 : 8;      // keep >= 8 K-tiles (128 k) per split; 4 for the launch-bound sizes
      if (want > nkt / min_kt) want = nkt / min_kt;
      if (want > 512) want = 512;
      if (want >= 8) want -= want % 8;      // multiples of 8: one K-slice per XCD at a time (see the kernel)
      splits = want < 1 ? 1 : (int)want;
    }
  }
  if (splits > nkt) splits = nkt > 0 ? nkt : 1;
  const int kt_per = (nkt + splits - 1) / splits;
  p.k_chunk = kt_per * kBK;
  p.splits = (nkt + kt_per - 1) / (kt_per > 0 ? kt_per : 1);
  if (p.splits < 1) p.splits = 1;
  return p;
}

static int check_gemm(const rec_gemm_desc* d) {
  REC_REQUIRE(d, REC_EINVAL, "desc is NULL");
  REC_REQUIRE(d->m >= 0 && d->n > 0 && d->k > 0, REC_EINVAL, "bad sizes M=%lld N=%d K=%d",
              (long long)d->m, d->n, d->k);
  REC_REQUIRE(d->lda > 0 && d->ldb > 0 && d->ldc >= d->n, REC_EINVAL, "bad leading dimensions");
  REC_REQUIRE(d->epilogue >= 0 && d->epilogue <= REC_EPI_DTANH, REC_EINVAL, "unknown epilogue %d",
              d->epilogue);
  return REC_OK;
}

template <int BM, int BN, int WM_, int WN_, int OCC, bool TA, bool TB, int EPI>
static void launch_one(const rec_gemm_desc* d, const GemmPlan& p, const float* A, const float* B,
                       float* C, const EpiArgs& e, float* partial, float* cpart, hipStream_t st) {
  const bool vec_a = (d->lda % 4 == 0) && (((uintptr_t)A) % 16 == 0);
  const bool vec_b = (d->ldb % 4 == 0) && (((uintptr_t)B) % 16 == 0);
  // branch-free tile loads: aligned operands whose contiguous extents are multiples of 4 (a float4 is then either
  // inside or outside the matrix) and hold at least one float4
  const int64_t a_inner = TA ? d->m : d->k, b_inner = TB ? d->k : d->n;
  const bool fast = vec_a && vec_b && a_inner % 4 == 0 && b_inner % 4 == 0 && a_inner >= 4 && b_inner >= 4;
  const bool fold = p.splits >= 8 && p.splits % 8 == 0 && p.tiles_total * p.splits < (1ll << 31);
  dim3 grid(fold ? (unsigned)(p.tiles_total * p.splits) : (unsigned)p.tiles_total,
            fold ? 1u : (unsigned)p.splits);
  constexpr int A_ELEMS = TA ? kBK * (BM + 4) : BM * (kBK + 4), B_ELEMS = kBK * (BN + 4);
  constexpr size_t shmem = 2 * (size_t)(A_ELEMS + B_ELEMS) * sizeof(float);
  static_assert(shmem <= 64 * 1024, "LDS tile too large");
  // candidate schedule (see gemm_f32_pipe_kernel): whole-tile problems on the two tiles that have the registers for it,
  // epilogues of the MLP path; opt-in until it has been through the full GPU suite
  constexpr bool kPipeCfg = ((BM == 256 && BN == 80 && WM_ == 8) || (BM == 80 && BN == 80 && WM_ == 5) ||
                             (BM == 144 && BN == 80 && WM_ == 9)) && !(TA && TB) &&
                            (EPI == REC_EPI_NONE || EPI == REC_EPI_BIAS || EPI == REC_EPI_BIAS_RELU ||
                             EPI == REC_EPI_RELU_MASK);
  if constexpr (kPipeCfg) {
    static const bool pipe_env = [] { const char* v = getenv("REC_GEMM_PIPE"); return !(v && *v == '0'); }();
    if (pipe_env && fast && d->m % BM == 0 && d->n % BN == 0 && d->k % kBK == 0 && d->lda < (1 << 23) &&
        d->ldb < (1 << 23)) {
      hipLaunchKernelGGL((gemm_f32_pipe_kernel<BM, BN, WM_, WN_, OCC, TA, TB, EPI>), grid, dim3(WM_ * WN_ * kWave), shmem,
                         st, d->m, d->n, d->k, A, (int64_t)d->lda, B, (int64_t)d->ldb, C, (int64_t)d->ldc, e, p.tiles_n,
                         p.tiles_total, p.k_chunk, partial, cpart, fold ? p.splits : 1);
      return;
    }
  }
  hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM_, WN_, OCC, TA, TB, EPI>), grid, dim3(WM_ * WN_ * kWave),
                     shmem, st, d->m, d->n, d->k, A, (int64_t)d->lda, B, (int64_t)d->ldb, C,
                     (int64_t)d->ldc, e, p.tiles_n, p.tiles_total, p.k_chunk, vec_a, vec_b, fast, partial, cpart,
                     fold ? p.splits : 1);
}

template <bool TA, bool TB, int EPI>
static void launch_cfg(const rec_gemm_desc* d, const GemmPlan& p, const float* A, const float* B,
                       float* C, const EpiArgs& e, float* partial, float* cpart, hipStream_t st) {
  switch (p.cfg) {
    case CFG_128x80: launch_one<128, 80, 4, 1, 5, TA, TB, EPI>(d, p, A, B, C, e, partial, cpart, st); break;
    case CFG_256x80: launch_one<256, 80, 8, 1, 2, TA, TB, EPI>(d, p, A, B, C, e, partial, cpart, st); break;
    case CFG_256x128: launch_one<256, 128, 4, 2, 2, TA, TB, EPI>(d, p, A, B, C, e, partial, cpart, st); break;
    case CFG_80x80: launch_one<80, 80, 5, 1, 4, TA, TB, EPI>(d, p, A, B, C, e, partial, cpart, st); break;
    case CFG_64x80: launch_one<64, 80, 4, 1, 6, TA, TB, EPI>(d, p, A, B, C, e, partial, cpart, st); break;
    case CFG_128x80_O4: launch_one<128, 80, 4, 1, 4, TA, TB, EPI>(d, p, A, B, C, e, partial, cpart, st); break;
    case CFG_144x80:     // planned only for dW = X^T G without an epilogue (plan_gemm)
      if constexpr (TA && !TB && EPI == REC_EPI_NONE) {
        launch_one<144, 80, 9, 1, 2, TA, TB, EPI>(d, p, A, B, C, e, partial, cpart, st);
      } else {
        launch_one<80, 80, 5, 1, 4, TA, TB, EPI>(d, p, A, B, C, e, partial, cpart, st);
      }
      break;
    default: launch_one<128, 128, 2, 2, 4, TA, TB, EPI>(d, p, A, B, C, e, partial, cpart, st); break;
  }
}

template <int EPI>
static void launch_epi(const rec_gemm_desc* d, const GemmPlan& p, const float* A, const float* B,
                       float* C, const EpiArgs& e, float* partial, float* cpart, hipStream_t st) {
  if (!d->trans_a && !d->trans_b) launch_cfg<false, false, EPI>(d, p, A, B, C, e, partial, cpart, st);
  else if (!d->trans_a && d->trans_b) launch_cfg<false, true, EPI>(d, p, A, B, C, e, partial, cpart, st);
  else if (d->trans_a && !d->trans_b) launch_cfg<true, false, EPI>(d, p, A, B, C, e, partial, cpart, st);
  else launch_cfg<true, true, EPI>(d, p, A, B, C, e, partial, cpart, st);
}

// The LDS-DMA kernel (gemm_glds.h) for the tall whole-tile problems of the MLP path: A row-major, many rows, N a
// multiple of one of its two block widths, the four epilogues of the forward / dX chain.  REC_GEMM_GLDS=0 switches it
// off (A/B measurements); -> false: not eligible, the caller takes the register-staged kernels.
static int device_cus() {
  static const int cus = [] {
    int dev = 0, v = kNumCU;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
    return v > 0 ? v : kNumCU;
  }();
  return cus;
}

template <int BM, int BN, bool TB>
static bool launch_glds_cfg(const rec_gemm_desc* d, const float* A, const float* B, float* C, const EpiArgs& e,
                            hipStream_t st) {
  const int tiles_n = d->n / BN;
  const int64_t tiles_total = (d->m / BM) * tiles_n;
  constexpr size_t shmem = (size_t)kGldsStages * (BM + BN) * 64;
  static_assert(shmem <= 64 * 1024, "LDS ring too large for the default dynamic-LDS limit");
  // persistent blocks: two per CU (63 KB of LDS each), a multiple of 8 (XCD-aware tile numbering)
  static const int cus = [] {
    int dev = 0, v = kNumCU;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
    return v > 0 ? v : kNumCU;
  }();
  static const int bpc = [] { const char* v = getenv("REC_GEMM_GLDS_BPC"); return v && *v ? atoi(v) : 2; }();
  static const int skew = [] { const char* v = getenv("REC_GEMM_GLDS_SKEW"); return v && *v ? atoi(v) : 0; }();
  int64_t grid = (int64_t)cus * bpc;
  grid -= grid % 8;
  if (grid > tiles_total) grid = tiles_total;
#define REC_GLDS_CASE(E)                                                                                         \
  case E:                                                                                                        \
    hipLaunchKernelGGL((gemm_f32_glds_kernel<BM, BN, TB, E>), dim3((unsigned)grid), dim3(512), shmem, st,        \
                       d->m, d->n, d->k, A, (int64_t)d->lda, B, (int64_t)d->ldb, C, (int64_t)d->ldc, e, tiles_n, \
                       tiles_total, skew);                                                                       \
    return true;
  switch (d->epilogue) {
    REC_GLDS_CASE(REC_EPI_NONE)
    REC_GLDS_CASE(REC_EPI_BIAS)
    REC_GLDS_CASE(REC_EPI_BIAS_RELU)
    REC_GLDS_CASE(REC_EPI_RELU_MASK)
  }
#undef REC_GLDS_CASE
  return false;
}

static bool launch_glds(const rec_gemm_desc* d, const float* A, const float* B, float* C, const EpiArgs& e,
                        hipStream_t st) {
  static const bool on = [] { const char* v = getenv("REC_GEMM_GLDS"); return !(v && *v == '0'); }();
  if (!on || d->trans_a || d->split_k > 1 || d->m < 8192 || d->k % kBK != 0 || d->k < 2 * kBK) return false;
  if (d->lda % 4 || d->ldb % 4 || ((uintptr_t)A) % 16 || ((uintptr_t)B) % 16) return false;
  if (d->ldc % 4 || ((uintptr_t)C) % 16) return false;                       // float4 stores of C
  if (d->epilogue == REC_EPI_RELU_MASK && (e.ld0 % 4 || ((uintptr_t)e.aux0) % 16)) return false;
  const int e_ = d->epilogue;
  if (!(e_ == REC_EPI_NONE || e_ == REC_EPI_BIAS || e_ == REC_EPI_BIAS_RELU || e_ == REC_EPI_RELU_MASK)) return false;
  static const int bm128 = [] { const char* v = getenv("REC_GEMM_GLDS_BM"); return v && atoi(v) == 128; }();
  if (bm128 && d->n % 80 == 0 && d->m % 128 == 0)
    return d->trans_b ? launch_glds_cfg<128, 80, true>(d, A, B, C, e, st)
                      : launch_glds_cfg<128, 80, false>(d, A, B, C, e, st);
  // N a multiple of 80: gemm_f32_pipe_kernel is as fast or 3-5 % faster there (profiles/r03_gemm_glds.txt), so the
  // ring takes these shapes only on request (REC_GEMM_GLDS_80=1: A/B runs, tests)
  const char* v80 = getenv("REC_GEMM_GLDS_80");      // read per call (tests flip it inside one process)
  const bool n80 = v80 && *v80 == '1';
  if (n80 && d->n % 80 == 0 && d->m % 256 == 0 && (d->m / 256) * (int64_t)(d->n / 80) < (1ll << 31))
    return d->trans_b ? launch_glds_cfg<256, 80, true>(d, A, B, C, e, st)
                      : launch_glds_cfg<256, 80, false>(d, A, B, C, e, st);
  if (d->n % 208 == 0 && d->m % 128 == 0 && (d->m / 128) * (int64_t)(d->n / 208) < (1ll << 31))
    return d->trans_b ? launch_glds_cfg<128, 208, true>(d, A, B, C, e, st)
                      : launch_glds_cfg<128, 208, false>(d, A, B, C, e, st);
  if (d->n % 144 == 0 && d->m % 128 == 0 && (d->m / 128) * (int64_t)(d->n / 144) < (1ll << 31))   // 432 = 27 x 16
    return d->trans_b ? launch_glds_cfg<128, 144, true>(d, A, B, C, e, st)
                      : launch_glds_cfg<128, 144, false>(d, A, B, C, e, st);
  return false;
}

// The bf16 x 3 form (gemm_bf16x3.h) for the tall problems of the towers: C = epi(A @ op(B)) with A row-major, N >= 336 in
// column blocks of 26 / 16 / 14 MFMA tiles (x3_cols: at most 15 % padding), the four epilogues of the forward / dX chain.  B is split into its plane image
// in the caller's workspace by a ~4 us launch in front of the GEMM.  REC_GEMM_BF16X3 = 1 / 0 (read per call: tests flip
// it inside one process); results are f32-grade (error against float64 no larger than the exact-f32 kernels',
// profiles/r05_bf16x3.txt) but not the bits of an f32 fma chain.
constexpr bool kX3Default = true;      // every GPU test passes on it (profiles/r05_bf16x3.txt); =0: the exact-f32 MFMA kernels
static bool x3_enabled() {
  const char* v = getenv("REC_GEMM_BF16X3");
  return v && *v ? *v != '0' : kX3Default;
}
static bool x3_eligible(const rec_gemm_desc* d) {
  if (!x3_enabled() || d->trans_a || d->split_k > 1 || d->m < 8192 || d->n % 4 || d->k % 8 || d->k < 64)
    return false;
  return x3_cols(d->n).nt > 0 && x3_epilogue_ok(d->epilogue);
}
static bool launch_x3(const rec_gemm_desc* d, const float* A, const float* B, float* C, const EpiArgs& e,
                      const void* b_image, void* workspace, size_t workspace_bytes, hipStream_t st) {
  if (!x3_eligible(d)) return false;
  if (!x3_shape_ok(d->m, d->n, d->k, d->lda, d->ldc, A, C)) return false;
  if (e.aux0 && (e.ld0 % 4 || ((uintptr_t)e.aux0) % 16)) return false;                 // float4 aux operands
  if (e.aux1 && (e.ld1 % 4 || ((uintptr_t)e.aux1) % 16)) return false;
  if (e.out2 && (e.ld2 % 4 || ((uintptr_t)e.out2) % 16)) return false;
  if (b_image && ((uintptr_t)b_image) % 16 == 0)       // the caller split B ahead of time (rec_gemm_b_images)
    return x3_launch_gemm(d->epilogue, d->m, d->n, d->k, A, d->lda, (const char*)b_image, C, d->ldc, e, st) == REC_OK;
  if (!workspace || ((uintptr_t)workspace) % 16 || workspace_bytes < x3_image_bytes(d->k, d->n)) return false;   // sized without it
  char* img = (char*)workspace;
  if (x3_launch_split(B, d->ldb, d->k, d->n, d->trans_b ? 1 : 0, img, st) != REC_OK) return false;
  return x3_launch_gemm(d->epilogue, d->m, d->n, d->k, A, d->lda, img, C, d->ldc, e, st) == REC_OK;
}

template <int EPI>
static void launch_reduce(const rec_gemm_desc* d, const GemmPlan& p, const float* partial, float* C,
                          const EpiArgs& e, hipStream_t st, const float* cpart = nullptr, float* colsum_out = nullptr) {
  int64_t grid = (d->m * d->n + kBlock - 1) / kBlock;
  if (grid > kNumCU * 8) grid = kNumCU * 8;
  const int cblocks = cpart && colsum_out ? (d->n + kBlock - 1) / kBlock : 0;
  hipLaunchKernelGGL(splitk_reduce_kernel<EPI>, dim3((unsigned)(grid + cblocks)), dim3(kBlock), 0, st, d->m,
                     d->n, (int64_t)d->ldc, p.splits, partial, C, e, cpart, colsum_out, cblocks);
}

// dW = X^T G of the towers' own widths on the same bf16 x 3 arithmetic (gemm_bf16x3_dw_kernel): trans_a form, no
// epilogue, the planner's own K split; partial tiles and partial column sums in the caller's workspace, folded by the
// engine's split-K reduce in ascending slice order (deterministic).
static bool x3_dw_eligible(const rec_gemm_desc* d, X3DwPlan* pl) {
  // an explicit K split is a caller's request for few, long blocks that leave wave slots to a kernel running beside the
  // GEMM (the deferred dW_0 under the sparse update).  This kernel owns every SIMD's register file, so the two take turns
  // instead — and the step is still 30 us shorter (1.76 -> 1.73 ms, profiles/r05_bf16x3.txt); REC_GEMM_BF16X3=1 leaves
  // those calls on the exact-f32 kernel
  const char* v = getenv("REC_GEMM_BF16X3");
  const bool forced = !(v && *v == '1');
  const char* vd = getenv("REC_GEMM_BF16X3_DW");           // 0: forward / dX only (A/B runs)
  if (vd && *vd == '0') return false;
  if (!x3_enabled() || !d->trans_a || d->trans_b || (d->split_k != 0 && !forced && d->num_cus <= 0) ||
      d->epilogue != REC_EPI_NONE)
    return false;
  if (d->k < 8192 || d->m < 336 || d->m > 448 || d->n < 336 || d->n > 416) return false;
  if (d->lda % 4 || d->ldb % 4 || d->ldc % 4) return false;
  // a caller on a CU-restricted stream (the row-sharded step's partitioned tail) gets one resident round on ITS units
  const int cus = d->num_cus > 0 && d->num_cus < device_cus() ? d->num_cus : device_cus();
  return x3_dw_plan((int)d->m, d->n, d->k, cus, pl);
}
static size_t x3_dw_bytes(const rec_gemm_desc* d, const X3DwPlan& pl) {
  return align_up((size_t)pl.slices * d->m * d->ldc * sizeof(float), 256) + align_up((size_t)pl.slices * d->n * sizeof(float), 256);
}
static bool launch_x3_dw(const rec_gemm_desc* d, const float* A, const float* B, float* C, const EpiArgs& e, float* b_colsum,
                         void* workspace, size_t workspace_bytes, hipStream_t st) {
  X3DwPlan pl;
  if (!x3_dw_eligible(d, &pl)) return false;
  if (((uintptr_t)A) % 16 || ((uintptr_t)B) % 16 || ((uintptr_t)C) % 16) return false;
  if (!workspace || ((uintptr_t)workspace) % 16 || workspace_bytes < x3_dw_bytes(d, pl)) return false;
  float* partial = (float*)workspace;
  float* cpart = (float*)((char*)workspace + align_up((size_t)pl.slices * d->m * d->ldc * sizeof(float), 256));
  if (x3_launch_dw(pl, (int)d->m, d->n, d->k, A, d->lda, B, d->ldb, partial, d->ldc, b_colsum ? cpart : nullptr, st) != REC_OK)
    return false;
  GemmPlan sp{};
  sp.splits = pl.slices;
  launch_reduce<REC_EPI_NONE>(d, sp, partial, C, e, st, b_colsum ? cpart : nullptr, b_colsum);
  return true;
}

}  // namespace rec

using namespace rec;

// what the exact-f32 kernels need (the bf16 x 3 forms ask for more; a caller that sized its workspace with the switch off
// gets the exact-f32 kernels when it is flipped on, not an error)
static size_t plain_workspace_bytes(const rec_gemm_desc* desc) {
  if (skinny_dw(desc)) {
    const size_t z = (size_t)((desc->k + kSkinnyKC - 1) / kSkinnyKC);
    return align_up(z * desc->m * desc->ldc * sizeof(float), 256) + align_up(z * desc->n * sizeof(float), 256);
  }
  const GemmPlan p = plan_gemm(desc);
  return (p.splits > 1 ? align_up((size_t)p.splits * desc->m * desc->ldc * sizeof(float), 256) : 0) +
         align_up((size_t)p.splits * desc->n * sizeof(float), 256);
}

// a forward / dX call whose ReLU mask may travel as bits: it takes the bf16 x 3 forward / dX kernel
static bool relu_bits_ok(const rec_gemm_desc* d) {
  return (d->epilogue == REC_EPI_BIAS_RELU || d->epilogue == REC_EPI_RELU_MASK) && x3_eligible(d) && d->lda % 4 == 0 &&
         d->ldc % 4 == 0;
}
extern "C" int rec_gemm_relu_bits_bytes(const rec_gemm_desc* desc, int32_t* eligible, size_t* bytes) {
  if (int rc = check_gemm(desc)) return rc;
  REC_REQUIRE(bytes, REC_EINVAL, "bytes is NULL");
  const bool ok = relu_bits_ok(desc);
  if (eligible) *eligible = ok ? 1 : 0;
  *bytes = ok ? align_up(x3_relu_bits_bytes(desc->m, desc->n), 256) : 0;
  return REC_OK;
}

extern "C" int rec_gemm_f32_workspace_bytes(const rec_gemm_desc* desc, size_t* bytes) {
  if (int rc = check_gemm(desc)) return rc;
  REC_REQUIRE(bytes, REC_EINVAL, "bytes is NULL");
  if (skinny_dw(desc)) {
    const size_t z = (size_t)((desc->k + kSkinnyKC - 1) / kSkinnyKC);
    *bytes = align_up(z * desc->m * desc->ldc * sizeof(float), 256) + align_up(z * desc->n * sizeof(float), 256);
    return REC_OK;
  }
  const GemmPlan p = plan_gemm(desc);
  // [splits][M][ldc] partial tiles (split-K only) + [splits][N] partial column sums
  *bytes = (p.splits > 1 ? align_up((size_t)p.splits * desc->m * desc->ldc * sizeof(float), 256) : 0) +
           align_up((size_t)p.splits * desc->n * sizeof(float), 256);
  if (x3_eligible(desc) && *bytes < x3_image_bytes(desc->k, desc->n)) *bytes = align_up(x3_image_bytes(desc->k, desc->n), 256);   // B's plane image
  X3DwPlan dwp;
  if (x3_dw_eligible(desc, &dwp) && *bytes < x3_dw_bytes(desc, dwp)) *bytes = x3_dw_bytes(desc, dwp);
  return REC_OK;
}

// ---- weight images made ahead of the GEMMs that consume them (rec_gemm_epilogue_args.b_image) ---------------------------
extern "C" int rec_gemm_b_image_bytes(int32_t k, int32_t n, int32_t* eligible, size_t* bytes) {
  REC_REQUIRE(k > 0 && n > 0 && bytes, REC_EINVAL, "bad arguments");
  const bool ok = n % 4 == 0 && k % 8 == 0 && k >= 64 && x3_cols(n).nt > 0;
  if (eligible) *eligible = ok ? 1 : 0;
  *bytes = ok ? align_up(x3_image_bytes(k, n), 256) : 0;
  return REC_OK;
}

extern "C" int rec_gemm_b_images(int32_t count, const rec_gemm_b_image* items, void* stream) {
  REC_REQUIRE(count >= 0 && (count == 0 || items), REC_EINVAL, "bad arguments");
  for (int32_t at = 0; at < count; at += kX3BatchMax) {
    X3SplitBatch b{};
    b.count = count - at < kX3BatchMax ? count - at : kX3BatchMax;
    int64_t blocks = 0;
    for (int i = 0; i < b.count; ++i) {
      const rec_gemm_b_image& it = items[at + i];
      REC_REQUIRE(it.B && it.image && ((uintptr_t)it.image) % 16 == 0, REC_EINVAL, "item %d: null / unaligned pointer", at + i);
      REC_REQUIRE(it.k > 0 && it.n > 0 && it.n % 4 == 0 && it.k % 8 == 0 && it.k >= 64 && x3_cols(it.n).nt > 0, REC_ESHAPE,
                  "item %d: k %d x n %d has no image form (rec_gemm_b_image_bytes)", at + i, it.k, it.n);
      REC_REQUIRE(it.ldb >= (it.trans_b ? it.k : it.n), REC_EINVAL, "item %d: ldb too small", at + i);
      const X3Cols c = x3_cols(it.n);
      b.W[i] = it.B; b.img[i] = (char*)it.image; b.ldw[i] = it.ldb; b.K[i] = it.k; b.N[i] = it.n;
      b.trans[i] = it.trans_b ? 1 : 0;
      b.nkt[i] = (it.k + 31) / 32; b.np[i] = c.nt * 16; b.ncb[i] = 2 * c.ncb;
      b.first[i] = (unsigned)blocks;
      blocks += ((int64_t)b.ncb[i] * b.nkt[i] * b.np[i] * 4 + kBlock - 1) / kBlock;
    }
    b.first[b.count] = (unsigned)blocks;
    REC_REQUIRE(blocks < (1ll << 31), REC_ESHAPE, "too many blocks");
    hipLaunchKernelGGL(x3_split_batch_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, b);
    if (int rc = check_launch("rec_gemm_b_images")) return rc;
  }
  return REC_OK;
}

extern "C" int rec_gemm_plan_splits(const rec_gemm_desc* desc, int32_t num_cus, int32_t* splits) {
  if (int rc = check_gemm(desc)) return rc;
  REC_REQUIRE(splits && num_cus <= kNumCU, REC_EINVAL, "bad arguments");
  rec_gemm_desc d = *desc;
  d.split_k = 0;
  *splits = plan_gemm(&d, num_cus > 0 ? num_cus : kNumCU).splits;
  return REC_OK;
}

// Two independent GEMMs (neither reads what the other writes).  The dW / dX pair of one Linear's backward at the
// reference's batch sizes — both launch-bound, both reading the same gradient — goes out as ONE launch (gemm_direct.h:
// gemm_f32_direct_pair_kernel; every workgroup runs the code of the single launch: bit-identical to two calls); every
// other pair is two rec_gemm_f32 calls, desc0 first.
extern "C" int rec_gemm_f32_pair(const rec_gemm_desc* desc0, const float* A0, const float* B0, float* C0,
                                 const rec_gemm_epilogue_args* x0, const rec_gemm_desc* desc1, const float* A1,
                                 const float* B1, float* C1, const rec_gemm_epilogue_args* x1, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  if (int rc = check_gemm(desc0)) return rc;
  if (int rc = check_gemm(desc1)) return rc;
  static const rec_gemm_epilogue_args kNoArgs = {};
  const rec_gemm_epilogue_args* y0 = x0 ? x0 : &kNoArgs;
  const rec_gemm_epilogue_args* y1 = x1 ? x1 : &kNoArgs;
  const bool shape = desc0->m > 0 && desc1->m > 0 && A0 && B0 && C0 && A1 && B1 && C1 && desc0->trans_a &&
                     desc0->epilogue == REC_EPI_NONE && !desc1->trans_a &&
                     (desc1->epilogue == REC_EPI_NONE ||
                      ((desc1->epilogue == REC_EPI_RELU_MASK || desc1->epilogue == REC_EPI_DSIGMOID) && y1->aux0 &&
                       y1->ld_aux0 >= desc1->n)) &&
                     !y1->b_colsum;
  if (shape) {
    EpiArgs e0{}, e1{};
    e1.aux0 = y1->aux0;
    e1.ld0 = y1->ld_aux0;
    if (launch_direct_pair(desc0, A0, B0, C0, e0, y0->b_colsum, desc1, A1, B1, C1, e1, (hipStream_t)stream))
      return check_launch("rec_gemm_f32_pair");
  }
  if (int rc = rec_gemm_f32(desc0, A0, B0, C0, x0, workspace, workspace_bytes, stream)) return rc;
  return rec_gemm_f32(desc1, A1, B1, C1, x1, workspace, workspace_bytes, stream);
}

extern "C" int rec_gemm_f32(const rec_gemm_desc* desc, const float* A, const float* B, float* C,
                            const rec_gemm_epilogue_args* x, void* workspace,
                            size_t workspace_bytes, void* stream) {
  if (int rc = check_gemm(desc)) return rc;
  if (desc->m == 0) return REC_OK;
  REC_REQUIRE(A && B && C, REC_EINVAL, "null pointer argument");
  static const rec_gemm_epilogue_args kNoArgs = {};
  if (!x) x = &kNoArgs;
  const int epi = desc->epilogue;
  const float* bias = x->bias;
  const float *aux0 = x->aux0, *aux1 = x->aux1;
  const int ld_aux0 = x->ld_aux0, ld_aux1 = x->ld_aux1;
  float* b_colsum = x->b_colsum;
  REC_REQUIRE(!b_colsum || desc->trans_a, REC_EINVAL,
              "b_colsum (the bias gradient) is produced by the trans_a form dW = X^T dY only");
  REC_REQUIRE(!(epi == REC_EPI_BIAS || epi == REC_EPI_BIAS_RELU || epi == REC_EPI_CROSS ||
                epi == REC_EPI_BIAS_SIGMOID || epi == REC_EPI_MOE) || bias, REC_EINVAL,
              "epilogue needs bias");
  unsigned long long* relu_bits = (unsigned long long*)x->relu_bits;
  REC_REQUIRE(!relu_bits || ((epi == REC_EPI_BIAS_RELU || epi == REC_EPI_RELU_MASK) && relu_bits_ok(desc) &&
                             ((uintptr_t)relu_bits) % 8 == 0), REC_EINVAL,
              "relu_bits: only a BIAS_RELU / RELU_MASK call that takes the bf16 x 3 forward / dX kernel "
              "(rec_gemm_relu_bits_bytes says which)");
  REC_REQUIRE(!(epi == REC_EPI_RELU_MASK || epi == REC_EPI_CROSS || epi == REC_EPI_MOE ||
                epi == REC_EPI_DSIGMOID || epi == REC_EPI_DTANH) || (epi == REC_EPI_RELU_MASK && relu_bits) ||
                  (aux0 && ld_aux0 >= desc->n), REC_EINVAL, "epilogue needs aux0");
  REC_REQUIRE(!(epi == REC_EPI_CROSS || epi == REC_EPI_ADD || epi == REC_EPI_MOE) ||
                  (aux1 && ld_aux1 >= desc->n), REC_EINVAL, "epilogue needs aux1");
  REC_REQUIRE(epi != REC_EPI_ADD || !aux0 || ld_aux0 >= desc->n, REC_EINVAL, "bad ld_aux0");
  REC_REQUIRE(epi != REC_EPI_MOE || (x->row_scale && x->row_scale_stride >= 1), REC_EINVAL,
              "epilogue needs row_scale");
  hipStream_t st = (hipStream_t)stream;
  // REC_GEMM_NT_STORE=1: non-temporal C stores for the dX + ReLU' form (alone 6 % faster, in the DeepFM step nothing:
  // profiles/r04_gemm_variants.txt), 2: for every whole-tile GEMM; default off
  static const int nt_env = [] { const char* v = getenv("REC_GEMM_NT_STORE"); return v && *v ? atoi(v) : 0; }();
  const int nt_store = nt_env == 2 || (nt_env == 1 && epi == REC_EPI_RELU_MASK) ? 1 : 0;
  EpiArgs e{bias, aux0, aux1, x->row_scale, x->out2, ld_aux0, ld_aux1, x->row_scale_stride, x->ld_out2, nt_store, relu_bits};
  REC_REQUIRE(!x->out2 || x->ld_out2 >= desc->n, REC_EINVAL, "bad ld_out2");
  if (skinny_rows(desc) && !b_colsum) {
    const bool vec_a = desc->lda % 4 == 0 && ((uintptr_t)A) % 16 == 0 && desc->k % 4 == 0;
    int64_t grid = (desc->m + kBlock / kWave - 1) / (kBlock / kWave);
    if (grid > kNumCU * 8) grid = kNumCU * 8;
    const size_t shmem = (size_t)desc->n * desc->k * sizeof(float);
#define REC_GEMV_CASE(E)                                                                                   \
  case E:                                                                                                  \
    hipLaunchKernelGGL(gemv_rows_kernel<E>, dim3((unsigned)grid), dim3(kBlock), shmem, st, desc->m, desc->n, \
                       desc->k, A, (int64_t)desc->lda, B, (int64_t)desc->ldb, desc->trans_b != 0, C,       \
                       (int64_t)desc->ldc, e, vec_a);                                                      \
    break;
    switch (epi) {
      REC_GEMV_CASE(REC_EPI_NONE) REC_GEMV_CASE(REC_EPI_BIAS) REC_GEMV_CASE(REC_EPI_BIAS_RELU)
      REC_GEMV_CASE(REC_EPI_RELU_MASK) REC_GEMV_CASE(REC_EPI_CROSS) REC_GEMV_CASE(REC_EPI_BIAS_SIGMOID)
      REC_GEMV_CASE(REC_EPI_BIAS_TANH) REC_GEMV_CASE(REC_EPI_ADD) REC_GEMV_CASE(REC_EPI_MOE)
      REC_GEMV_CASE(REC_EPI_DSIGMOID) REC_GEMV_CASE(REC_EPI_DTANH)
    }
#undef REC_GEMV_CASE
    return check_launch("rec_gemm_f32 (skinny rows)");
  }
  if (skinny_dw(desc) && epi == REC_EPI_NONE) {
    const size_t need = plain_workspace_bytes(desc);
    REC_REQUIRE(workspace && workspace_bytes >= need, REC_EWORKSPACE, "workspace %zu < %zu", workspace_bytes, need);
    const int z = (int)((desc->k + kSkinnyKC - 1) / kSkinnyKC);
    float* part = (float*)workspace;
    float* cpart2 = (float*)((char*)workspace + align_up((size_t)z * desc->m * desc->ldc * sizeof(float), 256));
    hipLaunchKernelGGL(desc->m <= kBlock / 2 ? skinny_dw_narrow_kernel : skinny_dw_kernel, dim3(z), dim3(kBlock), 0, st,
                       (int)desc->m, desc->n, (int64_t)desc->k, A, (int64_t)desc->lda, B, (int64_t)desc->ldb,
                       (int64_t)desc->ldc, part, b_colsum ? cpart2 : nullptr);
    GemmPlan sp;
    sp.splits = z;
    launch_reduce<REC_EPI_NONE>(desc, sp, part, C, e, st);
    if (b_colsum)
      hipLaunchKernelGGL(colsum_reduce_kernel, dim3((desc->n + kBlock - 1) / kBlock), dim3(kBlock), 0, st,
                         desc->n, z, (const float*)cpart2, b_colsum);
    return check_launch("rec_gemm_f32 (skinny dW)");
  }
  if (launch_x3_dw(desc, A, B, C, e, b_colsum, workspace, workspace_bytes, st)) return check_launch("rec_gemm_f32 (bf16x3 dW)");
  // tall problems of the towers' own widths: whole row panels, one resident round (gemm_panel.h)
  if (!b_colsum && launch_x3(desc, A, B, C, e, x->b_image, workspace, workspace_bytes, st))
    return check_launch("rec_gemm_f32 (bf16x3)");
  // (no other kernel knows the bit mask: a call that asked for it never continues on one that would ignore it)
  REC_REQUIRE(!relu_bits, REC_EINVAL, "relu_bits: the call did not take the bf16 x 3 kernel (alignment / workspace)");
  if (!b_colsum && launch_panel(desc, A, B, C, e, st, device_cus())) return check_launch("rec_gemm_f32 (panel)");
  if (!b_colsum && launch_glds(desc, A, B, C, e, st)) return check_launch("rec_gemm_f32 (glds)");
  // the launch-bound sizes (the reference's own batches): one launch, a wave per 16 x 16 tile over the whole K (gemm_direct.h)
  if (launch_direct(desc, A, B, C, e, b_colsum, st)) return check_launch("rec_gemm_f32 (direct)");
  const GemmPlan p = plan_gemm(desc);
  // (The half-empty last round of blocks — 65536 x 400 on 256x80 tiles is 2.5 rounds of the 512 resident blocks and
  // costs three — was attacked in round 3 by giving the rows behind the whole rounds to a second launch on 64x80 tiles:
  // 2.28 -> 2.38 ms per DeepFM step, i.e. worse; removed.  profiles/r03_schedule_ab.txt)
  REC_REQUIRE(p.tiles_total < (1ll << 31), REC_ESHAPE, "too many tiles");
  float* partial = nullptr;
  float* cpart = nullptr;
  if (p.splits > 1 || b_colsum) {
    const size_t need = plain_workspace_bytes(desc);
    REC_REQUIRE(workspace && workspace_bytes >= need, REC_EWORKSPACE, "workspace %zu < %zu",
                workspace_bytes, need);
    size_t off = 0;
    if (p.splits > 1) {
      partial = (float*)workspace;
      off = align_up((size_t)p.splits * desc->m * desc->ldc * sizeof(float), 256);
    }
    // one K slice: the blocks of the first M-tile write the column sums straight into b_colsum (the fold of a single
    // partial row adds zeros to it: same bits, one launch less)
    if (b_colsum) cpart = p.splits == 1 ? b_colsum : (float*)((char*)workspace + off);
  }
  // MEASUREMENT ONLY (REC_GEMM_SKIP_REDUCE=1: wrong results): what the split-K reduce launches cost a step
  static const bool skip_reduce = [] { const char* v = getenv("REC_GEMM_SKIP_REDUCE"); return v && *v == '1'; }();
#define REC_EPI_CASE(E)                                                   \
  case E:                                                                 \
    if (partial) {                                                        \
      launch_epi<REC_EPI_NONE>(desc, p, A, B, C, e, partial, cpart, st);  \
      if (!skip_reduce) launch_reduce<E>(desc, p, partial, C, e, st, cpart, b_colsum); \
    } else {                                                              \
      launch_epi<E>(desc, p, A, B, C, e, nullptr, cpart, st);             \
    }                                                                     \
    break;
  switch (epi) {
    REC_EPI_CASE(REC_EPI_NONE)
    REC_EPI_CASE(REC_EPI_BIAS)
    REC_EPI_CASE(REC_EPI_BIAS_RELU)
    REC_EPI_CASE(REC_EPI_RELU_MASK)
    REC_EPI_CASE(REC_EPI_CROSS)
    REC_EPI_CASE(REC_EPI_BIAS_SIGMOID)
    REC_EPI_CASE(REC_EPI_BIAS_TANH)
    REC_EPI_CASE(REC_EPI_ADD)
    REC_EPI_CASE(REC_EPI_MOE)
    REC_EPI_CASE(REC_EPI_DSIGMOID)
    REC_EPI_CASE(REC_EPI_DTANH)
  }
#undef REC_EPI_CASE
  if (b_colsum && !partial && cpart != b_colsum && !skip_reduce)   // (split-K: the reduce launch above folded them too)
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((desc->n + kBlock - 1) / kBlock), dim3(kBlock), 0,
                       st, desc->n, p.splits, (const float*)cpart, b_colsum);
  return check_launch("rec_gemm_f32");
}

extern "C" int rec_colsum_workspace_bytes(int64_t m, int32_t n, size_t* bytes) {
  REC_REQUIRE(bytes && m >= 0 && n > 0, REC_EINVAL, "bad arguments");
  const int64_t nblk = (m + kColsumRows - 1) / kColsumRows;
  *bytes = align_up((size_t)(nblk > 0 ? nblk : 1) * n * sizeof(float), 256);
  return REC_OK;
}

extern "C" int rec_colsum(int64_t m, int32_t n, int32_t ld, const float* G, float* out,
                          void* workspace, size_t workspace_bytes, void* stream) {
  REC_REQUIRE(m >= 0 && n > 0 && ld >= n && out, REC_EINVAL, "bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (m == 0) {
    (void)hipMemsetAsync(out, 0, (size_t)n * sizeof(float), st);
    return REC_OK;
  }
  REC_REQUIRE(G, REC_EINVAL, "G is NULL");
  size_t need = 0;
  rec_colsum_workspace_bytes(m, n, &need);
  REC_REQUIRE(workspace && workspace_bytes >= need, REC_EWORKSPACE, "workspace %zu < %zu",
              workspace_bytes, need);
  const int nblk = (int)((m + kColsumRows - 1) / kColsumRows);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk), dim3(kBlock), 0, st, m, n, (int64_t)ld, G,
                     (float*)workspace);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((n + 15) / 16), dim3(kBlock), 0, st,
                     nblk, n, (const float*)workspace, out);
  return check_launch("rec_colsum");
}

extern "C" int rec_mlp_head_bwd_workspace_bytes(int64_t batch, int32_t n, size_t* bytes) {
  REC_REQUIRE(bytes && batch >= 0 && n > 0, REC_EINVAL, "bad arguments");
  *bytes = align_up((size_t)kNumCU * 8 * (n + 1) * sizeof(float), 256);
  return REC_OK;
}

extern "C" int rec_mlp_head_bwd(int64_t batch, int32_t n, const float* act, int64_t ld_act, const float* dz,
                                const float* w, int32_t relu, float* dx, int64_t ld_dx, float* dw, float* db,
                                void* workspace, size_t workspace_bytes, void* stream) {
  REC_REQUIRE(batch >= 0 && n > 0 && n % 4 == 0 && n <= kHeadChunks * 4, REC_ESHAPE,
              "rec_mlp_head_bwd: n must be a multiple of 4 and <= %d", kHeadChunks * 4);
  REC_REQUIRE(act && dz && w && dx && dw && db && ld_act >= n && ld_dx >= n && ld_act % 4 == 0 && ld_dx % 4 == 0 &&
                  ((uintptr_t)act) % 16 == 0 && ((uintptr_t)dx) % 16 == 0 && ((uintptr_t)w) % 16 == 0,
              REC_EINVAL, "bad arguments (16-byte aligned rows required)");
  size_t need = 0;
  rec_mlp_head_bwd_workspace_bytes(batch, n, &need);
  REC_REQUIRE(workspace && workspace_bytes >= need, REC_EWORKSPACE, "workspace %zu < %zu", workspace_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)workspace;
  int64_t grid = (batch + 31) / 32;                 // >= 32 rows per block
  if (grid > kNumCU * 8) grid = kNumCU * 8;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(mlp_head_bwd_kernel, dim3((unsigned)grid), dim3(kBlock), 0, st, batch, n, act, ld_act, dz, w, dx,
                     ld_dx, relu, partial);
  const int cols_per_block = kBlock / kWave;
  hipLaunchKernelGGL(mlp_head_fold_kernel, dim3((n + 1 + cols_per_block - 1) / cols_per_block), dim3(kBlock), 0, st,
                     (int)grid, n, (const float*)partial, dw, db);
  return check_launch("rec_mlp_head_bwd");
}
