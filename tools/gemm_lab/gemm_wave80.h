// Weight gradients of the 400-wide towers: C[M, N] = A^T B over a long K (A = activations [K, M], B = dZ [K, N], K = batch)
// with M, N multiples of 80 — ONE WAVE per 80 x 80 output tile and K slice.
//
// The tiled kernels give a wave 16 x 80 of the tile (gemm_f32_pipe_kernel<80, 80, 5 waves>): every MFMA needs 1.2 LDS
// fragment reads and the five waves of a block read the same B fragments five times.  Here a wave owns the whole tile, 25
// accumulator tiles (100 VGPRs): 0.4 fragment reads per MFMA as ds_read_b32, 0.16 with the accumulator columns permuted
// (fragment tile t < 4 of lane li = column 4 li + t, tile 4 = column 64 + li: the fragments of four tiles are one
// ds_read_b128 and the epilogue stores float4s — the layout of gemm_panel.h).  No block-level barrier exists (the block IS
// the wave); the K split supplies the parallelism: 25 tiles x 40 slices = 1000 single-wave blocks, four per CU.
// Partial tiles go to the split-K workspace like every other weight gradient ([splits][M][ldc], folded by
// splitk_reduce_kernel in ascending slice order), the bias gradient's partial column sums come from the blocks of tile row 0.
#pragma once

#include "gemm_epi.h"

namespace rec {

constexpr int kW80 = 80;                 // tile edge
constexpr int kW80Ld = kW80 + 4;         // LDS row stride (floats)

template <bool PERM>
__global__ __launch_bounds__(kWave) void gemm_dw_wave80_kernel(
    int M, int N, int K, const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
    int64_t ldc, int tiles_n, int tiles_total, int k_chunk, int splits, float* __restrict__ partial,
    float* __restrict__ colsum_partial) {
  __shared__ __attribute__((aligned(16))) float As[2][kBK * kW80Ld];
  __shared__ __attribute__((aligned(16))) float Bs[2][kBK * kW80Ld];
  // blocks of one K slice share an XCD (its L2 holds the slice's rows of A and B for the 25 tiles that read them)
  const int w = blockIdx.x;
  const int xcd = w % 8, slot = w / 8;
  const int kz = xcd + 8 * (slot / tiles_total);
  const int tile = slot % tiles_total;
  if (kz >= splits) return;
  const int m0 = (tile / tiles_n) * kW80, n0 = (tile % tiles_n) * kW80;
  const int k_begin = kz * k_chunk;
  const int k_end = (k_begin + k_chunk < K) ? k_begin + k_chunk : K;
  const int nkt = k_end > k_begin ? (k_end - k_begin) / kBK : 0;
  const int lane = threadIdx.x;
  const int li = lane & 15, g = lane >> 4;
  f32x4_t acc[5][5];
#pragma unroll
  for (int a = 0; a < 5; ++a)
#pragma unroll
    for (int b = 0; b < 5; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // this lane's five float4s of each [16 x 80] tile: v = lane + 64 i -> k row v / 20, columns 4 (v % 20) ..
  uint32_t oa[5], ob[5];
  int ls[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int v = lane + 64 * i, kr = v / 20, c4 = (v % 20) * 4;
    oa[i] = (uint32_t)(kr * lda + c4);
    ob[i] = (uint32_t)(kr * ldb + c4);
    ls[i] = kr * kW80Ld + c4;
  }
  const float* a_base = A + (int64_t)k_begin * lda + m0;
  const float* b_base = B + (int64_t)k_begin * ldb + n0;
  const int64_t a_step = (int64_t)kBK * lda, b_step = (int64_t)kBK * ldb;
  float4 p0a0, p0a1, p0a2, p0a3, p0a4, p0b0, p0b1, p0b2, p0b3, p0b4;
  float4 p1a0, p1a1, p1a2, p1a3, p1a4, p1b0, p1b1, p1b2, p1b3, p1b4;
#define REC_W80_LOAD(S, T)                                                     \
  {                                                                            \
    const float* ap = a_base + (T) * a_step;                                   \
    const float* bp = b_base + (T) * b_step;                                   \
    S##a0 = *reinterpret_cast<const float4*>(ap + oa[0]);                      \
    S##a1 = *reinterpret_cast<const float4*>(ap + oa[1]);                      \
    S##a2 = *reinterpret_cast<const float4*>(ap + oa[2]);                      \
    S##a3 = *reinterpret_cast<const float4*>(ap + oa[3]);                      \
    S##a4 = *reinterpret_cast<const float4*>(ap + oa[4]);                      \
    S##b0 = *reinterpret_cast<const float4*>(bp + ob[0]);                      \
    S##b1 = *reinterpret_cast<const float4*>(bp + ob[1]);                      \
    S##b2 = *reinterpret_cast<const float4*>(bp + ob[2]);                      \
    S##b3 = *reinterpret_cast<const float4*>(bp + ob[3]);                      \
    S##b4 = *reinterpret_cast<const float4*>(bp + ob[4]);                      \
  }
#define REC_W80_STORE(S, BUF)                                                  \
  {                                                                            \
    *reinterpret_cast<float4*>(&As[BUF][ls[0]]) = S##a0;                       \
    *reinterpret_cast<float4*>(&As[BUF][ls[1]]) = S##a1;                       \
    *reinterpret_cast<float4*>(&As[BUF][ls[2]]) = S##a2;                       \
    *reinterpret_cast<float4*>(&As[BUF][ls[3]]) = S##a3;                       \
    *reinterpret_cast<float4*>(&As[BUF][ls[4]]) = S##a4;                       \
    *reinterpret_cast<float4*>(&Bs[BUF][ls[0]]) = S##b0;                       \
    *reinterpret_cast<float4*>(&Bs[BUF][ls[1]]) = S##b1;                       \
    *reinterpret_cast<float4*>(&Bs[BUF][ls[2]]) = S##b2;                       \
    *reinterpret_cast<float4*>(&Bs[BUF][ls[3]]) = S##b3;                       \
    *reinterpret_cast<float4*>(&Bs[BUF][ls[4]]) = S##b4;                       \
  }
  const bool do_colsum = colsum_partial != nullptr && m0 == 0;
  float csum0 = 0.f, csum1 = 0.f;          // columns lane and 64 + lane (< 80) of the tile
  // the k-steps of one staged tile: lane (li, g) feeds k = 4 g + s of the tile in step s
#define REC_W80_TILE(BUF)                                                                                  \
  {                                                                                                        \
    _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) {                                                     \
      const float* ar = &As[BUF][(g * 4 + s_) * kW80Ld];                                                   \
      const float* br = &Bs[BUF][(g * 4 + s_) * kW80Ld];                                                   \
      float af[5], bf[5];                                                                                  \
      if (PERM) {                                                                                          \
        const float4 ta = *reinterpret_cast<const float4*>(ar + 4 * li);                                   \
        const float4 tb = *reinterpret_cast<const float4*>(br + 4 * li);                                   \
        af[0] = ta.x; af[1] = ta.y; af[2] = ta.z; af[3] = ta.w; af[4] = ar[64 + li];                       \
        bf[0] = tb.x; bf[1] = tb.y; bf[2] = tb.z; bf[3] = tb.w; bf[4] = br[64 + li];                       \
      } else {                                                                                             \
        _Pragma("unroll") for (int t_ = 0; t_ < 5; ++t_) { af[t_] = ar[t_ * 16 + li]; bf[t_] = br[t_ * 16 + li]; } \
      }                                                                                                    \
      _Pragma("unroll") for (int a = 0; a < 5; ++a)                                                        \
        _Pragma("unroll") for (int b = 0; b < 5; ++b)                                                      \
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a], bf[b], acc[a][b], 0, 0, 0);              \
    }                                                                                                      \
    if (do_colsum) {                                                                                       \
      _Pragma("unroll") for (int kk = 0; kk < kBK; ++kk) {                                                 \
        csum0 += Bs[BUF][kk * kW80Ld + lane];                                                              \
        if (lane < kW80 - 64) csum1 += Bs[BUF][kk * kW80Ld + 64 + lane];                                   \
      }                                                                                                    \
    }                                                                                                      \
  }
  if (nkt > 0) {
    REC_W80_LOAD(p0, 0)
    REC_W80_STORE(p0, 0)
    if (nkt > 1) REC_W80_LOAD(p1, 1)
    __syncthreads();
    int t = 0;
    // invariant at the loop top: LDS[0] = tile t, set 1 = tile t + 1 in flight
    for (; t + 3 < nkt; t += 2) {
      REC_W80_LOAD(p0, t + 2)                 // (pinned: left alone, the loads sink to the end of the tile's MFMAs
      __builtin_amdgcn_sched_barrier(0);      //  and the stores below wait out their whole latency)
      REC_W80_STORE(p1, 1)
      REC_W80_TILE(0)
      __syncthreads();
      REC_W80_LOAD(p1, t + 3)
      __builtin_amdgcn_sched_barrier(0);
      REC_W80_STORE(p0, 0)
      REC_W80_TILE(1)
      __syncthreads();
    }
    if (t + 1 < nkt) {                  // 2 or 3 tiles left: LDS[0] = tile t, set 1 = tile t + 1
      const bool more = t + 2 < nkt;
      if (more) REC_W80_LOAD(p0, t + 2)
      REC_W80_STORE(p1, 1)
      REC_W80_TILE(0)
      __syncthreads();
      ++t;
      if (more) {
        REC_W80_STORE(p0, 0)
        REC_W80_TILE(1)
        __syncthreads();
        ++t;
      }
    }
    if (t & 1) REC_W80_TILE(1) else REC_W80_TILE(0)
  }
#undef REC_W80_LOAD
#undef REC_W80_STORE
#undef REC_W80_TILE
  if (do_colsum) {
    colsum_partial[(int64_t)kz * N + n0 + lane] = csum0;
    if (lane < kW80 - 64) colsum_partial[(int64_t)kz * N + n0 + 64 + lane] = csum1;
  }
  // C/D of a 16 x 16 tile: register r of lane (li, g) = element (row 4 g + r, column li)
  float* out = partial + (int64_t)kz * M * ldc;
#pragma unroll
  for (int a = 0; a < 5; ++a) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = g * 4 + r;
      const int64_t i = m0 + (PERM ? (a < 4 ? 4 * rr + a : 64 + rr) : a * 16 + rr);
      float* o = out + i * ldc + n0;
      if (PERM) {
        *reinterpret_cast<float4*>(o + 4 * li) = make_float4(acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]);
        o[64 + li] = acc[a][4][r];
      } else {
#pragma unroll
        for (int b = 0; b < 5; ++b) o[b * 16 + li] = acc[a][b][r];
      }
    }
  }
}

// true when the shape is this kernel's: A^T B, whole 80 x 80 tiles, float4-loadable operands
static inline bool dw_wave80_ok(const rec_gemm_desc* d, const float* A, const float* B) {
  return d->trans_a && !d->trans_b && d->epilogue == REC_EPI_NONE && d->m % kW80 == 0 && d->n % kW80 == 0 &&
         d->m <= 1600 && d->n <= 1600 && d->k % kBK == 0 && d->k >= 64 * kBK && d->lda % 4 == 0 && d->ldb % 4 == 0 &&
         d->ldc % 4 == 0 && ((uintptr_t)A) % 16 == 0 && ((uintptr_t)B) % 16 == 0 && d->lda < (1 << 23) && d->ldb < (1 << 23);
}
// K slices for one resident round of single-wave blocks: four per CU, a multiple of 8 (one slice per XCD at a time)
static inline int dw_wave80_splits(const rec_gemm_desc* d, int num_cus) {
  const int tiles = (int)(d->m / kW80) * (d->n / kW80), nkt = d->k / kBK;
  int s = num_cus * 4 / tiles;
  s -= s % 8;
  if (s > nkt / 8) s = (nkt / 8) - (nkt / 8) % 8;
  return s < 8 ? 0 : s;
}
static inline void launch_dw_wave80(const rec_gemm_desc* d, int splits, const float* A, const float* B, float* partial,
                                    float* cpart, hipStream_t st, bool perm) {
  const int tiles_n = d->n / kW80, tiles = (int)(d->m / kW80) * tiles_n, nkt = d->k / kBK;
  const int kt_per = (nkt + splits - 1) / splits;
  dim3 grid((unsigned)(tiles * splits));
  if (perm)
    hipLaunchKernelGGL(gemm_dw_wave80_kernel<true>, grid, dim3(kWave), 0, st, (int)d->m, d->n, d->k, A, (int64_t)d->lda, B,
                       (int64_t)d->ldb, (int64_t)d->ldc, tiles_n, tiles, kt_per * kBK, splits, partial, cpart);
  else
    hipLaunchKernelGGL(gemm_dw_wave80_kernel<false>, grid, dim3(kWave), 0, st, (int)d->m, d->n, d->k, A, (int64_t)d->lda, B,
                       (int64_t)d->ldb, (int64_t)d->ldc, tiles_n, tiles, kt_per * kBK, splits, partial, cpart);
}

}  // namespace rec
