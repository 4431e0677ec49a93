// f32 GEMM on the bf16 matrix pipe (the towers' tall GEMMs: deepfm/net.py:142-174, dcn_v2/net.py:140-184,214-226,
// slot_dnn/net.py:77-84 and their backward).  Every f32 operand is split into three bf16 terms (x = x0 + x1 + x2 to
// 2^-25 |x|, each the round-to-nearest bf16 of the running remainder: 8 + 8 + 8 significand bits = the whole f32
// significand) and the product is accumulated in f32 from the six term products with i + j <= 2 (a0 b0, a0 b1, a1 b0,
// a0 b2, a1 b1, a2 b0) — what is dropped (a1 b2, a2 b1, a2 b2) is below 2^-24 |a b| per product.  Against float64 the
// results carry the error of the exact-f32 kernels (1-4e-7 of sum |a||b| either way: the MFMA's f32 accumulation sets
// it, tests/test_gemm_gpu.py::test_gemm_bf16x3*, tests/test_bf16x3_split.py, profiles/r05_bf16x3.txt); they are f32-grade
// but not the bit pattern of an f32 fma chain, hence the switch: REC_GEMM_BF16X3=0 selects the exact-f32 MFMA kernels.
// Six v_mfma_f32_16x16x32_bf16 per K 32 are 96 matrix-pipe cycles where eight v_mfma_f32_16x16x4_f32 are 256
// (the "bf16 x 3" form): 1.4-1.5x in time on the 65 536 x 400 x 400 problems.
//
// (1) forward / dX form:  C[M,N] = epi(A[M,K] @ W'),  A f32 row-major (k contiguous, lda), W' a pre-split IMAGE
// The weight (400 x 400: ~1 MB of planes, L2-resident) is split once per call by x3_split_kernel into the exact LDS
// image of every k-step; the activation operand is split in registers by the wave that multiplies it:
//   * block tile 128 rows x one COLUMN BLOCK of 2 NT MFMA tiles (NT 13 / 8 / 7 per wave: N 400 -> one block of 416
//     columns, 432 -> two of 224, 512 -> two of 256, 1560 -> four of 416), one block per CU (LDS); 8 waves as 4 (M) x 2 (N),
//     wave tile 32 x 208 = 2 x 13 MFMA tiles, 104 accumulator registers, TWO waves per SIMD (a wave's conversion and
//     waits run beside its partner's MFMAs: 7-8 % over the four-wave form with 64-row wave tiles, which the CrossNet
//     epilogues keep for their registers);
//   * A: a lane's fragment of v_mfma_f32_16x16x32_bf16 is 8 consecutive k of one row = 32 contiguous bytes of f32:
//     loaded global -> registers (a wave instruction covers 16 rows x one full 128-B line each), one k-step ahead, and
//     split there (4.5 VALU instructions per element: v_cvt_pk_bf16_f32, shift / mask, v_pk_add_f32) — no LDS traffic for A;
//   * W': k-step image [plane 3][n NP][4 chunks of 8 k] bf16, chunk c of row n stored at slot c ^ f(n % 16),
//     f = {0,3,2,1}[i / 4] (the swizzle of gemm_glds.h: one conflict-free ds_read_b128 per fragment), 78 KB per k-step
//     at NT 13, two stages = 156 of the 160 KB; filled by LDS-DMA (global_load_lds_dwordx4: the global image IS the LDS
//     image, so the copy is lane-linear), one k-step ahead, one barrier per k-step;
//   * MFMAs with the operands swapped (W' fragment first) as in gemm_glds.h: lane (i, g) ends up with
//     C[row i][columns 4g .. 4g+3] of a tile -> float4 stores, float4 bias / aux loads.
// Shapes: K % 8 == 0 (a lane's 8-k chunk is inside or outside K as a whole), N % 4 == 0, rows 16-B aligned, M >= 8192
// (the caller's rule: short problems do not fill the chip with 128-row blocks).  Non-finite operands: an Inf (or a finite
// |x| > 3.39e38, which rounds to the bf16 Inf) leaves Inf - Inf in its remainder and the output is NaN where the f32 fma
// chain gives Inf — both mean the same for a training step.
// (2) weight-gradient form (gemm_bf16x3_dw_kernel): further down.
#pragma once

#include <atomic>

#include "gemm_epi.h"
#include "gemm_glds.h"      // glb_void_t / lds_void_t, wait_vmcnt

namespace rec {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

constexpr int kX3NT = 13;                       // MFMA column tiles per wave of the widest column block
constexpr int kX3NP = 2 * kX3NT * 16;           // 416 columns per block
constexpr int kX3BM = 128;                      // rows per block
constexpr int kX3MT = 4;                        // MFMA row tiles per wave (64 rows)
constexpr int kX3Stage = 3 * kX3NP * 64;        // the largest k-step stage: 79 872 B (two stages = 156 of the 160 KB)
// Column blocks: ONE wave's NT MFMA column tiles (NT in {13, 8, 7}), always an even number of them: 400 -> 2 x 13 tiles,
// 432 -> 4 x 7, 512 -> 4 x 8, 1560 -> 8 x 13; the weight image is [column block][k-step][plane][NP rows][64 B].  A
// workgroup covers WN = 1 or 2 adjacent column blocks (its waves are WM along M x WN along N).
template <int NT>
struct X3Geo {
  static constexpr int NP = NT * 16;            // columns per column block (zero weight columns behind N)
  static constexpr int Plane = NP * 64;         // one plane of one k-step: NP rows x 32 bf16
  static constexpr int Stage = 3 * Plane;       // one column block's k-step image
  static constexpr int Pieces = Stage / 1024;   // LDS-DMA pieces per column block and k-step (3 * NT)
};
struct X3Cols { int nt, ncb; };                 // tiles per wave, PAIRS of column blocks
inline X3Cols x3_cols(int N) {                  // least padded choice; nt = 0: none within 15 % of N
  const int tiles = (N + 15) / 16;
  X3Cols best{0, 0};
  int best_pad = 1 << 30;
  const int cand[3] = {13, 8, 7};
  for (int c = 0; c < 3; ++c) {
    const int ncb = (tiles + 2 * cand[c] - 1) / (2 * cand[c]);
    const int pad = ncb * 2 * cand[c] - tiles;
    if (pad < best_pad) { best_pad = pad; best = X3Cols{cand[c], ncb}; }
  }
  if (best_pad * 100 > tiles * 15) best.nt = 0;
  return best;
}
inline size_t x3_image_bytes(int K, int N) {
  const X3Cols c = x3_cols(N);
  return (size_t)c.ncb * ((K + 31) / 32) * 3 * (2 * c.nt * 16) * 64;
}

__device__ __forceinline__ int x3_swz(int i) { return (4 - (i >> 2)) & 3; }   // {0,3,2,1}[i / 4], i = n % 16

// the three bf16 terms of an (even, odd) pair of floats, packed [even | odd << 16] per plane.  Each term is the
// round-to-nearest bf16 of the running remainder (v_cvt_pk_bf16_f32); the remainders are exact f32 subtractions
// (|x - x0| <= 2^-9 |x| fits 16 bits, |r1 - x1| <= 2^-17 |x| fits 8-9 bits), so x0 + x1 + x2 = x to 2^-25 |x| and the
// dropped products a1 b2 + a2 b1 + a2 b2 stay below 2^-24 |a b| with either sign (a truncating split leaves them at
// 2^-21 |a b| and all of one sign: a bias of ~1e-7 of sum |a||b| that showed against the float64 bound).
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned x3_cvt_pk(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}
__device__ __forceinline__ void x3_split_pair(float xe, float xo, unsigned& p0, unsigned& p1, unsigned& p2) {
  p0 = x3_cvt_pk(xe, xo);
  const float re = xe - __uint_as_float(p0 << 16), ro = xo - __uint_as_float(p0 & 0xffff0000u);   // exact
  p1 = x3_cvt_pk(re, ro);
  const float se = re - __uint_as_float(p1 << 16), so = ro - __uint_as_float(p1 & 0xffff0000u);   // exact
  p2 = x3_cvt_pk(se, so);
}

// W [K,N] (trans 0: element (n, k) = W[k * ldw + n]) or W [N,K] (trans 1: W[n * ldw + k]) -> image; one thread per
// (column block, k-step, row of the block, 8-k chunk).  Rows n >= N and chunks k >= K are zero.
__device__ __forceinline__ void x3_split_one(const float* __restrict__ W, int64_t ldw, int K, int N, int trans,
                                             char* __restrict__ img, int nkt, int np, int ncb, int64_t tid) {
  if (tid >= (int64_t)ncb * nkt * np * 4) return;
  const int c = (int)(tid & 3), nl = (int)((tid >> 2) % np);
  const int kt = (int)(((tid >> 2) / np) % nkt), cb = (int)((tid >> 2) / np / nkt);
  const int n = cb * np + nl;
  float x[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = kt * 32 + c * 8 + j;
    x[j] = (n < N && k < K) ? (trans ? W[(int64_t)n * ldw + k] : W[(int64_t)k * ldw + n]) : 0.f;
  }
  u32x4_t p0, p1, p2;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    unsigned a, b, e;
    x3_split_pair(x[2 * d], x[2 * d + 1], a, b, e);
    p0[d] = a; p1[d] = b; p2[d] = e;
  }
  const size_t plane = (size_t)np * 64;
  char* dst = img + ((size_t)cb * nkt + kt) * 3 * plane + (size_t)nl * 64 + ((c ^ x3_swz(nl & 15)) * 16);
  *reinterpret_cast<u32x4_t*>(dst) = p0;
  *reinterpret_cast<u32x4_t*>(dst + plane) = p1;
  *reinterpret_cast<u32x4_t*>(dst + 2 * plane) = p2;
}
__global__ __launch_bounds__(kBlock) void x3_split_kernel(const float* __restrict__ W, int64_t ldw, int K, int N, int trans,
                                                          char* __restrict__ img, int nkt, int np, int ncb) {
  x3_split_one(W, ldw, K, N, trans, img, nkt, np, ncb, (int64_t)blockIdx.x * kBlock + threadIdx.x);
}
// every weight image of a step in ONE launch (rec_gemm_b_images): item i owns blocks [first[i], first[i + 1])
constexpr int kX3BatchMax = 8;
struct X3SplitBatch {
  const float* W[kX3BatchMax];
  char* img[kX3BatchMax];
  int64_t ldw[kX3BatchMax];
  int K[kX3BatchMax], N[kX3BatchMax], trans[kX3BatchMax], nkt[kX3BatchMax], np[kX3BatchMax], ncb[kX3BatchMax];
  unsigned first[kX3BatchMax + 1];
  int count;
};
__global__ __launch_bounds__(kBlock) void x3_split_batch_kernel(const X3SplitBatch b) {
  int i = 0;
#pragma unroll
  for (int j = 1; j < kX3BatchMax; ++j)
    if (j < b.count && blockIdx.x >= b.first[j]) i = j;
  x3_split_one(b.W[i], b.ldw[i], b.K[i], b.N[i], b.trans[i], b.img[i], b.nkt[i], b.np[i], b.ncb[i],
               (int64_t)(blockIdx.x - b.first[i]) * kBlock + threadIdx.x);
}

// one tile's three plane fragments (ds_read_b128, offset = tile * 1024 + plane * PLANE; plane 2 through a second
// base because 2 * 26 624 + 12 * 1024 does not fit the 16-bit offset field)
template <int T, int PLANE>
__device__ __forceinline__ void x3_read_frags(u32x4_t (&bf)[3], unsigned sb, unsigned sb2) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bf[0]) : "v"(sb), "n"(T * 1024));
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bf[1]) : "v"(sb), "n"(T * 1024 + PLANE));
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bf[2]) : "v"(sb2), "n"(T * 1024));
}
template <int PLANE>
__device__ __forceinline__ void x3_read_frags_t(u32x4_t (&bf)[3], unsigned sb, unsigned sb2, int t) {
  switch (t) {            // t is a compile-time constant at every call (fully unrolled tile loop)
    case 0: x3_read_frags<0, PLANE>(bf, sb, sb2); break;
    case 1: x3_read_frags<1, PLANE>(bf, sb, sb2); break;
    case 2: x3_read_frags<2, PLANE>(bf, sb, sb2); break;
    case 3: x3_read_frags<3, PLANE>(bf, sb, sb2); break;
    case 4: x3_read_frags<4, PLANE>(bf, sb, sb2); break;
    case 5: x3_read_frags<5, PLANE>(bf, sb, sb2); break;
    case 6: x3_read_frags<6, PLANE>(bf, sb, sb2); break;
    case 7: x3_read_frags<7, PLANE>(bf, sb, sb2); break;
    case 8: x3_read_frags<8, PLANE>(bf, sb, sb2); break;
    case 9: x3_read_frags<9, PLANE>(bf, sb, sb2); break;
    case 10: x3_read_frags<10, PLANE>(bf, sb, sb2); break;
    case 11: x3_read_frags<11, PLANE>(bf, sb, sb2); break;
    default: x3_read_frags<12, PLANE>(bf, sb, sb2); break;
  }
}

#ifndef REC_X3_LAB
#define REC_X3_LAB 0           // measurement builds: 1 no C stores, 2 no A loads in the k-loop, 3 no LDS-DMA in the k-loop, 4 no conversion
#endif
#ifndef REC_X3_TG
#define REC_X3_TG 2            // column tiles per MFMA group of the forward / dX kernel (lab knob: 1 = tile by tile)
#endif
// wait until at most `left` LDS reads are outstanding (a constant after unrolling: 0, 3, 6 or 9), tied to the TG x 3
// fragments that must have landed (LDS returns in order)
template <int TG>
__device__ __forceinline__ void x3_wait_group(u32x4_t (&f)[TG][3], int left) {
#define REC_X3_TIE(U) if constexpr (TG > U) asm volatile("" : "+v"(f[U][0]), "+v"(f[U][1]), "+v"(f[U][2]));
  switch (left) {
    case 0: asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory"); break;
    default: asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); break;
  }
  REC_X3_TIE(0) REC_X3_TIE(1) REC_X3_TIE(2) REC_X3_TIE(3)
#undef REC_X3_TIE
}

#ifndef REC_X3_DW_PIPE_STORES
#define REC_X3_DW_PIPE_STORES 1      // default schedule of the weight-gradient kernel's LDS stores (REC_X3_DW_PIPE=0 / 1 at run time)
#endif
#ifndef REC_X3_DW_VALU_PER_MFMA
#define REC_X3_DW_VALU_PER_MFMA 2      // conversion instructions laid behind every MFMA of the tiles that carry a patch column
                                       // (a 16-cycle MFMA leaves one wave ~2 issue slots: 3 measured 162 us against 159)
#endif
#ifndef REC_X3_PRODUCTS
#define REC_X3_PRODUCTS 6      // lab knob: 3 = a0 b0 + a0 b1 + a1 b0 only (~2^-14 of scale: NOT f32-grade), 1 = plain bf16
#endif

// WM = waves along M (2: four waves, one per SIMD, 64-row wave tiles; 4: EIGHT waves, two per SIMD, 32-row wave tiles —
// a wave's conversion / waits sit beside its SIMD partner's MFMAs, at twice the W' fragment reads per MFMA).
// WN = 1 (four waves, WM = 4): a workgroup is 128 rows x ONE column block, 80 KB of LDS, 256 registers per wave — TWO
// workgroups share a CU, each SIMD holds one wave of either.  Unlike the two waves of one 8-wave workgroup, which the
// k-step barrier keeps in lockstep (both convert, both wait, both store their C tiles at the same time: the matrix pipe
// idles), the two workgroups drift apart and one's prologue / conversion / C stores run beside the other's MFMAs.
// Grid: 1-D; workgroup b runs on XCD b % 8 (round-robin dispatch), and the column blocks of one row tile are CONSECUTIVE
// on that XCD (j = b / 8: cb = j % ncb, row tile = (j / ncb) * 8 + b % 8): the second reader of an A tile finds it in L2.
template <int NT, int EPI, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64, WN == 1 ? 2 : 1) void gemm_bf16x3_kernel(
    int64_t M, int N, int K, const float* __restrict__ A, int64_t lda, const char* __restrict__ Bimg_all,
    float* __restrict__ C, int64_t ldc, EpiArgs epi, int ncb) {
  using Geo = X3Geo<NT>;
  constexpr int MT = kX3BM / 16 / WM;                         // MFMA row tiles per wave
  constexpr int NW = WM * WN;                                 // waves per block
  constexpr int BStage = WN * Geo::Stage;                     // LDS bytes of one k-step stage
  extern __shared__ __attribute__((aligned(1024))) char x3_smem[];
  const int lane = threadIdx.x % kWave;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int li = lane & 15, g = lane >> 4;
  const int wm = wave % WM, wn = wave / WM;
  const int nkt = (K + 31) / 32;
  int64_t row_tile;
  int cb0;                                                    // first column block of this workgroup
  if constexpr (WN == 1) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    cb0 = j % ncb;
    row_tile = (int64_t)(j / ncb) * 8 + xcd;
    if (row_tile * kX3BM >= M) return;                        // the grid is padded to whole groups of 8 row tiles
  } else {
    row_tile = blockIdx.x;
    cb0 = blockIdx.y * WN;
  }
  const int64_t m0 = row_tile * kX3BM + wm * (MT * 16);
  const char* Bimg = Bimg_all + (size_t)cb0 * nkt * Geo::Stage;

  // ---- A: per-lane row pointers (rows behind M re-read row M-1: finite data, never stored)
  const float* ap[MT];
#pragma unroll
  for (int a = 0; a < MT; ++a) {
    int64_t r = m0 + a * 16 + li;
    r = r < M ? r : M - 1;
    ap[a] = A + r * lda + g * 8;
  }
  float4 araw[MT][2];
  // a chunk behind K (only in the last k-step, K % 32 != 0) is read from the row's first chunk instead and zeroed when
  // it is CONVERTED, a k-step later: a select right behind the load would park the wave until the load returns
  auto load_a = [&](int kt) {
#if REC_X3_LAB == 2      // lab: no A loads in the k-loop (the first k-step's registers are converted every step)
    if (kt > 0) return;
#endif
    const int koff = kt * 32 + g * 8 < K ? kt * 32 : 0;      // K % 8 == 0: the lane's chunk is in or out as a whole
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      araw[a][0] = *reinterpret_cast<const float4*>(ap[a] + koff);
      araw[a][1] = *reinterpret_cast<const float4*>(ap[a] + koff + 4);
    }
  };
  // ---- W': LDS-DMA of one k-step image; pieces dealt round-robin to the four waves
  auto issue_b = [&](int kt, int stage) {
#if REC_X3_LAB == 3      // lab: no LDS-DMA in the k-loop (every k-step multiplies the first image)
    if (kt > 0) return;
#endif
    const char* src = Bimg + (size_t)kt * Geo::Stage + lane * 16;
    const char* dst = x3_smem + stage * BStage;
#pragma unroll
    for (int j = 0; j < (WN * Geo::Pieces + NW - 1) / NW; ++j) {
      const int piece = wave + NW * j;                        // column block w = piece / Pieces, piece q of its image
      if (piece < WN * Geo::Pieces) {
        const int w = WN == 1 ? 0 : piece / Geo::Pieces, q = piece - w * Geo::Pieces;
        __builtin_amdgcn_global_load_lds((glb_void_t*)(src + (size_t)w * nkt * Geo::Stage + q * 1024),
                                         (lds_void_t*)(dst + w * Geo::Stage + q * 1024), 16, 0, 0);
      }
    }
  };

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[a][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // byte address (LDS offset = low half of the flat address) of this lane's fragment slot in tile 0, plane 0, stage 0
  const unsigned lds_base = (unsigned)(uintptr_t)x3_smem + wn * Geo::Stage + li * 64 + ((g ^ x3_swz(li)) * 16);

  issue_b(0, 0);
  load_a(0);
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();

  for (int kt = 0; kt < nkt; ++kt) {
    const int stage = kt & 1;
    // this k-step's A fragments: three planes per row tile
    u32x4_t af[MT][3];
    if (kt == nkt - 1 && (K & 31) != 0 && kt * 32 + g * 8 >= K) {       // the K tail: this lane's chunk does not exist
#pragma unroll
      for (int a = 0; a < MT; ++a) araw[a][0] = araw[a][1] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      const float x[8] = {araw[a][0].x, araw[a][0].y, araw[a][0].z, araw[a][0].w,
                          araw[a][1].x, araw[a][1].y, araw[a][1].z, araw[a][1].w};
#pragma unroll
      for (int d = 0; d < 4; ++d) {
#if REC_X3_LAB == 4      // lab: no conversion (the raw bits as planes)
        af[a][0][d] = __float_as_uint(x[2 * d]); af[a][1][d] = __float_as_uint(x[2 * d + 1]); af[a][2][d] = __float_as_uint(x[2 * d]) ^ 1u;
#else
        unsigned p0, p1, p2;
        x3_split_pair(x[2 * d], x[2 * d + 1], p0, p1, p2);
        af[a][0][d] = p0; af[a][1][d] = p1; af[a][2][d] = p2;
#endif
      }
    }
    if (kt + 1 < nkt) {                                      // next k-step: W' image into the other stage, A into registers
      issue_b(kt + 1, stage ^ 1);
      load_a(kt + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    // W' fragments: ds_read_b128 as inline asm with COUNTED waits.  Through the builtin path the compiler waits
    // lgkmcnt(0) before every tile while LDS-DMA loads are in flight (it cannot order them against LDS reads), i.e.
    // also for the fragments it has just requested for the NEXT tile; here tile t's MFMAs wait for their own three
    // reads only (lgkmcnt(3): LDS returns in order) and the next tile's reads stay in flight underneath them.
#if REC_X3_LAB == 3
    const unsigned sb = lds_base;
#else
    const unsigned sb = lds_base + stage * BStage;
#endif
    // Column tiles are multiplied in GROUPS of TG: the six term products of a group go product by product over all
    // TG x MT accumulators, so that two MFMAs on ONE accumulator are TG x MT instructions apart (TG = 1: every other
    // instruction waits for its predecessor's result — the issue stalls of profiles/r05_bf16x3.txt section 8).
    constexpr int TG = REC_X3_TG, NG = (NT + TG - 1) / TG;
    u32x4_t bf[2][TG][3];
    auto read_group = [&](int gi, int slot) {
#pragma unroll
      for (int u = 0; u < TG; ++u)
        if (gi * TG + u < NT) x3_read_frags_t<Geo::Plane>(bf[slot][u], sb, sb + 2 * Geo::Plane, gi * TG + u);
    };
    read_group(0, 0);
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      const int slot = gi & 1;
      const int n_cur = NT - gi * TG < TG ? NT - gi * TG : TG;
      int left = 0;                                            // reads that may stay in flight: the next group's
      if (gi + 1 < NG) {
        read_group(gi + 1, slot ^ 1);
        left = 3 * (NT - (gi + 1) * TG < TG ? NT - (gi + 1) * TG : TG);
      }
      x3_wait_group<TG>(bf[slot], left);
      // smallest terms first
#define REC_X3_MFMA(PB, PA)                                                                                          \
  _Pragma("unroll") for (int u = 0; u < TG; ++u) if (u < n_cur) {                                                    \
    _Pragma("unroll") for (int a = 0; a < MT; ++a) acc[a][gi * TG + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(    \
        __builtin_bit_cast(bf16x8_t, bf[slot][u][PB]), __builtin_bit_cast(bf16x8_t, af[a][PA]), acc[a][gi * TG + u], 0, 0, 0); \
  }
#if REC_X3_PRODUCTS >= 6
      REC_X3_MFMA(2, 0)
      REC_X3_MFMA(1, 1)
      REC_X3_MFMA(0, 2)
#endif
#if REC_X3_PRODUCTS >= 3
      REC_X3_MFMA(1, 0)
      REC_X3_MFMA(0, 1)
#endif
      REC_X3_MFMA(0, 0)
#undef REC_X3_MFMA
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kt + 1 < nkt) {
      wait_vmcnt<0>();                     // my pieces of the next image (and my next A registers) have landed
      __builtin_amdgcn_s_barrier();        // ... everybody's; everybody is done reading this stage
    }
  }

  // ---- epilogue: float4 per lane and tile, aux / bias operands of a row tile loaded ahead of its stores
  const int n_base = (cb0 + wn) * Geo::NP + g * 4;
#pragma unroll
  for (int a = 0; a < MT; ++a) {
    const int64_t i = m0 + a * 16 + li;
    const bool row_ok = i < M;
    const int64_t ic = row_ok ? i : M - 1;
    f32x4_t x0[NT], x1[EpiUses<EPI>::aux1 ? NT : 1], bj[NT];
    // ReLU mask as BITS (rec_gemm_epilogue_args.relu_bits): the forward of a layer (BIAS_RELU) leaves one 64-bit word per
    // (row, column block, lane group g) — bit 4 t + c = "column n_base + 16 t + c of the output is > 0" — and the dX GEMM
    // that needs the layer's ReLU' (RELU_MASK: same N, hence the same column blocks and the same lane <-> element map)
    // reads ITS OWN word back instead of the whole activation: 8 B per lane and row tile for NT float4 loads
    unsigned long long* bits_at = epi.relu_bits ? epi.relu_bits + ((ic * ncb + (cb0 + wn)) * 4 + g) : nullptr;
    unsigned long long mbits = 0ull;
    if constexpr (EPI == REC_EPI_RELU_MASK) {
      if (bits_at) mbits = *bits_at;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int j = n_base + t * 16;
      const int jc = j < N ? j : 0;
#pragma unroll
      for (int c = 0; c < 4; ++c) bj[t][c] = load_bias<EPI>(jc + c, epi);
      if constexpr (EPI == REC_EPI_ADD) {                    // aux0 may be absent (gemm_epi.h load_aux0)
        x0[t] = epi.aux0 ? *reinterpret_cast<const f32x4_t*>(epi.aux0 + ic * epi.ld0 + jc) : f32x4_t{0.f, 0.f, 0.f, 0.f};
      } else if constexpr (EPI == REC_EPI_RELU_MASK) {
        if (bits_at) {
#pragma unroll
          for (int c = 0; c < 4; ++c) x0[t][c] = ((mbits >> (4 * t + c)) & 1ull) ? 1.f : 0.f;
        } else {
          x0[t] = *reinterpret_cast<const f32x4_t*>(epi.aux0 + ic * epi.ld0 + jc);
        }
      } else if constexpr (EpiUses<EPI>::aux0) {
        x0[t] = *reinterpret_cast<const f32x4_t*>(epi.aux0 + ic * epi.ld0 + jc);
      } else {
        x0[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
      if constexpr (EpiUses<EPI>::aux1) x1[t] = *reinterpret_cast<const f32x4_t*>(epi.aux1 + ic * epi.ld1 + jc);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int j = n_base + t * 16;
      f32x4_t v, u;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float xx1 = EpiUses<EPI>::aux1 ? x1[EpiUses<EPI>::aux1 ? t : 0][c] : 0.f;
        v[c] = apply_epi<EPI>(acc[a][t][c], x0[t][c], xx1, bj[t][c], ic, epi);
        u[c] = acc[a][t][c] + bj[t][c];
        if constexpr (EPI == REC_EPI_BIAS_RELU) mbits |= (unsigned long long)(v[c] > 0.f ? 1 : 0) << (4 * t + c);
      }
#if REC_X3_LAB == 1      // lab: the kernel without its C stores (the compare keeps the epilogue arithmetic alive)
      if (row_ok && j < N && v[0] == 1.2345678e33f) {
#else
      if (row_ok && j < N) {
#endif
        *reinterpret_cast<f32x4_t*>(C + i * ldc + j) = v;
        if constexpr (EPI == REC_EPI_CROSS) {                // CROSS also stores u = acc + bias (saved for the backward)
          if (epi.out2) *reinterpret_cast<f32x4_t*>(epi.out2 + i * epi.ld2 + j) = u;
        }
      }
    }
    if constexpr (EPI == REC_EPI_BIAS_RELU) {
      if (bits_at && row_ok) *bits_at = mbits;
    }
  }
}

// ------------------------------------------------------------------------------------------ weight gradient (TN form)
//   P[z][Kin,Nout] = X[rows of slice z, Kin]^T @ G[rows of slice z, Nout]       (dW = X^T G, K = the batch, split over z)
// Both operands are activations: f32, row-major, the contraction index (the batch row m) is the STRIDED one.  A thread
// loads an 8 (m) x 4 (columns) patch as eight coalesced float4s, splits it in registers into the three bf16 planes with
// the pairs packed ALONG m, and writes each column's 8 m as one 16-B chunk into LDS — the transpose happens in that
// write, and the fragment of v_mfma_f32_16x16x32_bf16 (8 consecutive contraction indices of one column) is again one
// ds_read_b128.  LDS stage = [X planes 3 x 208 columns x 64 B][G planes the same] = 79 872 B, two stages.
// Slot of (column c, m-group g) inside a plane: g * 208 + 16 a + q, a = c / 16, i = c % 16,
// q = i / 4 + 4 ((i % 4 + a) % 4): conflict-free for the fragment read (16 lanes = 16 columns of one tile: q is a
// bijection of i) AND for the patch write (16 lanes = columns 4 L + e: q = L % 4 + 4 ((e + L / 4) % 4)).
// Block = 4 waves as 2 x 2 over an output block of up to 13 x 13 MFMA tiles (208 x 208), a wave up to 7 x 7 tiles; output
// blocks x slices = grid, block b on XCD b % 8 takes slice (j / nob) * 8 + b % 8, j = b / 8, output block j % nob: the
// output blocks of a slice run on ONE XCD, so X and G rows come from HBM once and from that L2 the other times.
// Partial sums go to P (float4 along Nout), the column sums of G's slice (the bias gradient) to cpart[z][Nout] from the
// blocks of the first Kin block; the engine's split-K reduce folds both in ascending z.
constexpr int kX3DwCols = 208;                         // columns of one operand per block (13 MFMA tiles)
constexpr int kX3DwPlane = kX3DwCols * 64;             // 13 312 B
constexpr int kX3DwOperand = 3 * kX3DwPlane;           // 39 936 B
constexpr int kX3DwWT = 7;                             // MFMA tiles per wave along Nout (and, at most, along Kin)

__device__ __forceinline__ int x3_dw_q(int i, int a) { return (i >> 2) + 4 * (((i & 3) + a) & 3); }

template <int T>
__device__ __forceinline__ void x3_dw_read(u32x4_t (&f)[3], unsigned base) {     // tile T of this lane's operand region
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[0]) : "v"(base), "n"(T * 256));
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[1]) : "v"(base), "n"(T * 256 + kX3DwPlane));
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[2]) : "v"(base), "n"(T * 256 + 2 * kX3DwPlane));
}

__device__ __forceinline__ void x3_dw_read_u(u32x4_t (&f)[3], unsigned base, int u) {   // u: constant after unrolling
  switch (u) {
    case 0: x3_dw_read<0>(f, base); break;
    case 1: x3_dw_read<1>(f, base); break;
    case 2: x3_dw_read<2>(f, base); break;
    case 3: x3_dw_read<3>(f, base); break;
    case 4: x3_dw_read<4>(f, base); break;
    case 5: x3_dw_read<5>(f, base); break;
    default: x3_dw_read<6>(f, base); break;
  }
}

// wait until at most `left` LDS reads are outstanding (a compile-time constant after unrolling), tied to the fragments
// that must have landed
__device__ __forceinline__ void x3_dw_wait_frags(u32x4_t (&f)[3], int left) {
#define REC_X3_WAIT_CASE(N) case N: asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2])); break;
  switch (left) {
    REC_X3_WAIT_CASE(0) REC_X3_WAIT_CASE(1) REC_X3_WAIT_CASE(2) REC_X3_WAIT_CASE(3) REC_X3_WAIT_CASE(4) REC_X3_WAIT_CASE(5)
    REC_X3_WAIT_CASE(6) REC_X3_WAIT_CASE(7) REC_X3_WAIT_CASE(8) REC_X3_WAIT_CASE(9) REC_X3_WAIT_CASE(10) REC_X3_WAIT_CASE(11)
    REC_X3_WAIT_CASE(12) REC_X3_WAIT_CASE(13) REC_X3_WAIT_CASE(14) REC_X3_WAIT_CASE(15)
    default: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2])); break;
  }
#undef REC_X3_WAIT_CASE
}

struct X3DwArgs {
  const float* X; int64_t ldx;      // [rows, Kin]
  const float* G; int64_t ldg;      // [rows, Nout]
  int64_t rows;                     // the batch (contraction length), % 32 == 0
  int kin, nout;
  int kb_tiles, nb_tiles;           // MFMA tiles per output block along Kin / Nout (<= 13)
  int kblocks, nblocks;             // output blocks along Kin / Nout
  int slices, steps_per_slice;      // slices % 8 == 0; 32-row k-steps per slice
  float* P; int64_t ldp;            // [slices][Kin][ldp]
  float* cpart;                     // [slices][Nout] or null
};

// PW = MFMA tiles per wave along Kin (7: blocks of 13 / 12 tiles, 5: blocks of 9 / 10).  A wave always computes its whole
// PW x 7 tile set: the tiles behind its share (the smaller half of an odd split, the block's edge) multiply whatever
// those LDS slots hold and are never stored — the block waits for its largest wave anyway, and a branch per MFMA costs
// more than the MFMA (measured in the ISA: 674 branches and 725 accumulator moves with guards, none without).
template <int PW, bool PIPE>
__global__ __launch_bounds__(256) void gemm_bf16x3_dw_kernel(X3DwArgs w) {
  extern __shared__ __attribute__((aligned(1024))) char x3_smem[];
  const int tid = threadIdx.x;
  const int lane = tid % kWave;
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
  const int li = lane & 15, g = lane >> 4;
  const int wk = wave & 1, wn = wave >> 1;
  // which output block, which slice (see the header)
  const int nob = w.kblocks * w.nblocks;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int ob = j % nob, slice = (j / nob) * 8 + xcd;
  const int kb = ob / w.nblocks, nb = ob % w.nblocks;
  const int k_tile0 = kb * w.kb_tiles, n_tile0 = nb * w.nb_tiles;                       // first MFMA tile of the block
  const int kt_blk = min(w.kb_tiles, (w.kin + 15) / 16 - k_tile0);                       // tiles of this block
  const int nt_blk = min(w.nb_tiles, (w.nout + 15) / 16 - n_tile0);
  const int k_split = (kt_blk + 1) / 2, n_split = (nt_blk + 1) / 2;                      // wave 0 takes the larger half
  const int my_k0 = wk ? k_split : 0, my_p = __builtin_amdgcn_readfirstlane(wk ? kt_blk - k_split : k_split);
  const int my_n0 = wn ? n_split : 0, my_q = __builtin_amdgcn_readfirstlane(wn ? nt_blk - n_split : n_split);
  const int64_t row0 = (int64_t)slice * w.steps_per_slice * 32;
  int nsteps = (int)min((int64_t)w.steps_per_slice, (w.rows - row0) / 32);
  if (nsteps < 0) nsteps = 0;

  // ---- staging: thread t < 208 owns patch (m-group t / 52, column group t % 52) of BOTH operands
  const bool stager = tid < 4 * (kX3DwCols / 4);
  const int mg = stager ? tid / (kX3DwCols / 4) : 0, cg = tid % (kX3DwCols / 4);     // (the idle threads re-read m-group 0)
  const int xc = k_tile0 * 16 + cg * 4, gc = n_tile0 * 16 + cg * 4;                      // first global column of the patch
  const bool x_ok = stager && cg * 4 < kt_blk * 16 && xc < w.kin;                        // kin, nout % 4 == 0
  const bool g_ok = stager && cg * 4 < nt_blk * 16 && gc < w.nout;
  const float* xp = w.X + (row0 + mg * 8) * w.ldx + (x_ok ? xc : 0);
  const float* gp = w.G + (row0 + mg * 8) * w.ldg + (g_ok ? gc : 0);
  float4 xr[8], gr[8];
  auto load_patches = [&](int kt) {
#ifdef REC_X3_DW_NO_LOADS      // lab: the k-loop without its global loads (the step-0 registers are converted every step)
    if (kt > 0) return;
#endif
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      xr[r] = *reinterpret_cast<const float4*>(xp + ((int64_t)kt * 32 + r) * w.ldx);
      gr[r] = *reinterpret_cast<const float4*>(gp + ((int64_t)kt * 32 + r) * w.ldg);
    }
  };
  // LDS byte offsets of the patch's four columns (chunk mg): column c = cg * 4 + e -> a = cg / 4, i = 4 (cg % 4) + e
  unsigned wr_off[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int a = cg >> 2, i = 4 * (cg & 3) + e;
    wr_off[e] = (unsigned)((mg * kX3DwCols + 16 * a + x3_dw_q(i, a)) * 16);
  }
  float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
  // Branch-free stores: a thread without a patch (tid >= 208) writes its chunks to a 16-B slot of its own behind the
  // two stages (the 4 KB that are left of the 160), so that the conversion is straight-line code the scheduler can lay
  // between the MFMAs of a tile.
  const unsigned w_stage = stager ? (unsigned)kX3Stage : 0u, w_oper = stager ? (unsigned)kX3DwOperand : 0u,
                 w_plane = stager ? (unsigned)kX3DwPlane : 0u;
  char* const w_base = stager ? x3_smem : x3_smem + 2 * kX3Stage + tid * 16;
  if (!stager) { wr_off[0] = wr_off[1] = wr_off[2] = wr_off[3] = 0u; }
  // column e of this thread's X and G patches -> six 16-B chunks (X planes 0..2, G planes 0..2) ...
  auto convert_compute = [&](int e, u32x4_t (&ch)[6]) {
    float xv[8], gv[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      xv[r] = e == 0 ? xr[r].x : e == 1 ? xr[r].y : e == 2 ? xr[r].z : xr[r].w;
      gv[r] = e == 0 ? gr[r].x : e == 1 ? gr[r].y : e == 2 ? gr[r].z : gr[r].w;
    }
    // the values pass through an empty volatile asm: pure arithmetic on registers loaded at the top of the k-step is
    // otherwise scheduled right behind those loads (in front of every MFMA of the step, waiting for the loads) whatever
    // the source order says
    asm volatile("" : "+v"(xv[0]), "+v"(xv[1]), "+v"(xv[2]), "+v"(xv[3]), "+v"(xv[4]), "+v"(xv[5]), "+v"(xv[6]), "+v"(xv[7]),
                      "+v"(gv[0]), "+v"(gv[1]), "+v"(gv[2]), "+v"(gv[3]), "+v"(gv[4]), "+v"(gv[5]), "+v"(gv[6]), "+v"(gv[7]));
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      unsigned a, b, c;
#ifdef REC_X3_DW_NOCVT      /* lab (timing only): the k-loop without the f32 -> 3 x bf16 arithmetic */
      a = __float_as_uint(xv[2 * d]); b = __float_as_uint(xv[2 * d + 1]); c = a ^ b;
#else
      x3_split_pair(x_ok ? xv[2 * d] : 0.f, x_ok ? xv[2 * d + 1] : 0.f, a, b, c);
#endif
      ch[0][d] = a; ch[1][d] = b; ch[2][d] = c;
    }
    float cs = 0.f;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const float u0 = g_ok ? gv[2 * d] : 0.f, u1 = g_ok ? gv[2 * d + 1] : 0.f;
      cs += u0 + u1;
      unsigned a, b, c;
#ifdef REC_X3_DW_NOCVT
      a = __float_as_uint(u0); b = __float_as_uint(u1); c = a ^ b;
#else
      x3_split_pair(u0, u1, a, b, c);
#endif
      ch[3][d] = a; ch[4][d] = b; ch[5][d] = c;
    }
    if (e == 0) csum.x += cs; else if (e == 1) csum.y += cs; else if (e == 2) csum.z += cs; else csum.w += cs;
  };
  // ... and the chunks into the stage: the transpose happens in this write
  auto convert_write = [&](int stage, int e, const u32x4_t (&ch)[6]) {
    char* sx = w_base + stage * w_stage + wr_off[e];
    char* sg = sx + w_oper;
#ifdef REC_X3_DW_NOWRITE    /* lab (timing only): one LDS store per patch column and operand instead of three */
    *reinterpret_cast<u32x4_t*>(sx) = ch[0] ^ ch[1] ^ ch[2];
    *reinterpret_cast<u32x4_t*>(sg) = ch[3] ^ ch[4] ^ ch[5];
#else
    *reinterpret_cast<u32x4_t*>(sx) = ch[0];
    *reinterpret_cast<u32x4_t*>(sx + w_plane) = ch[1];
    *reinterpret_cast<u32x4_t*>(sx + 2 * w_plane) = ch[2];
    *reinterpret_cast<u32x4_t*>(sg) = ch[3];
    *reinterpret_cast<u32x4_t*>(sg + w_plane) = ch[4];
    *reinterpret_cast<u32x4_t*>(sg + 2 * w_plane) = ch[5];
#endif
  };
  auto convert_part = [&](int stage, int e) {
    u32x4_t ch[6];
    convert_compute(e, ch);
    convert_write(stage, e, ch);
  };
  auto convert_store = [&](int stage) {
#pragma unroll
    for (int e = 0; e < 4; ++e) convert_part(stage, e);
  };

  f32x4_t acc[PW][kX3DwWT];
#pragma unroll
  for (int a = 0; a < PW; ++a)
#pragma unroll
    for (int t = 0; t < kX3DwWT; ++t) acc[a][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragment read bases: tile (first tile of the wave + u) lives at 256 B x tile + the lane's slot, whose q depends on
  // tile % 4 -> four bases per operand, indexed by u % 4 (the unrolled loops pick them statically)
  const unsigned lds0 = (unsigned)(uintptr_t)x3_smem;
  unsigned xb[4], gb[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int ax = my_k0 + u, an = my_n0 + u;
    xb[u] = lds0 + (unsigned)((g * kX3DwCols + 16 * my_k0 + x3_dw_q(li, ax)) * 16);
    gb[u] = lds0 + kX3DwOperand + (unsigned)((g * kX3DwCols + 16 * my_n0 + x3_dw_q(li, an)) * 16);
  }

  if (nsteps > 0) {
    load_patches(0);
    convert_store(0);
  }
  __syncthreads();

  // One k-step.  MORE: there is a next step — its patches are loaded at the top and converted / written to the other stage
  // IN the MFMA stream of tiles 3 .. 6, one patch column per tile (sched_group_barrier: one MFMA, two VALU, ... — the
  // conversion's ~85 VALU instructions per column ride in the shadow of the tile's 6 PW MFMAs instead of following them).
  auto step = [&](int kt, auto more_c) {
    constexpr bool MORE = decltype(more_c)::value;
    const int stage = kt & 1;
    if (MORE) load_patches(kt + 1);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned so = stage * kX3Stage;
    // this wave's X fragments (the MFMA's second operand: Kin becomes the accumulator's lane index)
    // Fragment reads in the order of their first use: G tile 0, then the X tiles one by one, then G tile 1.  Tile 0 runs
    // row tile by row tile behind counted waits (its first MFMAs need 6 of the 3 PW + 6 reads, not all of them);
    // from tile 1 on the products run across the row tiles as in the forward kernel.
    u32x4_t af[PW][3];
    u32x4_t bf[2][3];
    u32x4_t pend[6];                                      // (REC_X3_DW_PIPE_STORES: a converted column waiting for its stores)
    x3_dw_read_u(bf[0], gb[0] + so, 0);
#pragma unroll
    for (int u = 0; u < PW; ++u) x3_dw_read_u(af[u], xb[u & 3] + so, u);
    x3_dw_read_u(bf[1], gb[1] + so, 1);
#pragma unroll
    for (int a = 0; a < PW; ++a) {                          // tile 0: reads still outstanding behind X tile a: 3 (PW - 1 - a) + 3
      x3_dw_wait_frags(af[a], 3 * (PW - 1 - a) + 3);
      const u32x4_t* b0 = bf[0];
#define REC_X3_DW_MFMA0(PB, PA)                                                                                \
  acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, b0[PB]),                      \
                                                      __builtin_bit_cast(bf16x8_t, af[a][PA]), acc[a][0], 0, 0, 0);
#if REC_X3_PRODUCTS >= 6
      REC_X3_DW_MFMA0(2, 0) REC_X3_DW_MFMA0(1, 1) REC_X3_DW_MFMA0(0, 2)
#endif
#if REC_X3_PRODUCTS >= 3
      REC_X3_DW_MFMA0(1, 0) REC_X3_DW_MFMA0(0, 1)
#endif
      REC_X3_DW_MFMA0(0, 0)
#undef REC_X3_DW_MFMA0
      __builtin_amdgcn_sched_barrier(0);                    // (the MFMAs of row tile a stay in front of the next wait)
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 1; t < kX3DwWT; ++t) {
      if (t + 1 < kX3DwWT) {
        x3_dw_read_u(bf[(t + 1) & 1], gb[(t + 1) & 3] + so, t + 1);
        asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(bf[t & 1][0]), "+v"(bf[t & 1][1]), "+v"(bf[t & 1][2]));
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bf[t & 1][0]), "+v"(bf[t & 1][1]), "+v"(bf[t & 1][2]));
      }
      const u32x4_t* b = bf[t & 1];
#define REC_X3_DW_MFMA(PB, PA)                                                                                 \
  _Pragma("unroll") for (int a = 0; a < PW; ++a) acc[a][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(            \
      __builtin_bit_cast(bf16x8_t, b[PB]), __builtin_bit_cast(bf16x8_t, af[a][PA]), acc[a][t], 0, 0, 0);
#if REC_X3_PRODUCTS >= 6
      REC_X3_DW_MFMA(2, 0) REC_X3_DW_MFMA(1, 1) REC_X3_DW_MFMA(0, 2)
#endif
#if REC_X3_PRODUCTS >= 3
      REC_X3_DW_MFMA(1, 0) REC_X3_DW_MFMA(0, 1)
#endif
      REC_X3_DW_MFMA(0, 0)
#undef REC_X3_DW_MFMA
#ifndef REC_X3_DW_SERIAL_CONVERT
      if constexpr (PIPE) {
      // the chunks of a patch column are converted under one tile's MFMAs and STORED under the next tile's, one
      // ds_write_b128 per 6 PW / 6 MFMAs (a burst of six stores from each of the four waves at the end of a tile had the
      // waves queue for the LDS store port with the matrix pipe idle): columns 0..3 converted in tiles WT-5..WT-2,
      // stored in tiles WT-4..WT-1
      if (MORE && t >= kX3DwWT - 5) {
        const bool wr = t >= kX3DwWT - 4, cv = t <= kX3DwWT - 2;
        if (wr) convert_write(stage ^ 1, t - (kX3DwWT - 4), pend);
        if (cv) convert_compute(t - (kX3DwWT - 5), pend);
#pragma unroll
        for (int i = 0; i < 6 * PW; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
          if (cv) __builtin_amdgcn_sched_group_barrier(0x002, REC_X3_DW_VALU_PER_MFMA, 0);
          if (wr && i % PW == PW / 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // one ds_write_b128
        }
        if (cv) __builtin_amdgcn_sched_group_barrier(0x002, 32, 0);       // what is left of the conversion
      }
      } else if (MORE && t >= kX3DwWT - 4) {
        convert_part(stage ^ 1, t - (kX3DwWT - 4));
#pragma unroll
        for (int i = 0; i < 6 * PW; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, REC_X3_DW_VALU_PER_MFMA, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x002, 32, 0);       // what is left of the conversion
        __builtin_amdgcn_sched_group_barrier(0x200, 6, 0);        // the column's six ds_write_b128
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
#ifdef REC_X3_DW_SERIAL_CONVERT
    if (MORE) convert_store(stage ^ 1);                 // the other stage: nobody reads it since the last barrier
#endif
    __syncthreads();
  };
  for (int kt = 0; kt + 1 < nsteps; ++kt) step(kt, std::true_type{});
  if (nsteps > 0) step(nsteps - 1, std::false_type{});

  // ---- partial tile: lane (i, g) holds P[Kin = tile a, row i][Nout = tile t, columns 4g .. 4g+3]
  float* P = w.P + (int64_t)slice * w.kin * w.ldp;
#pragma unroll
  for (int a = 0; a < PW; ++a) {
    if (a >= my_p) continue;
    const int ki = (k_tile0 + my_k0 + a) * 16 + li;
#pragma unroll
    for (int t = 0; t < kX3DwWT; ++t) {
      if (t >= my_q) continue;
      const int nj = (n_tile0 + my_n0 + t) * 16 + g * 4;
      if (ki < w.kin && nj < w.nout) *reinterpret_cast<f32x4_t*>(P + (int64_t)ki * w.ldp + nj) = acc[a][t];
    }
  }
  // ---- column sums of G's slice: the four m-groups of a column group meet in LDS (every stage is free now)
  if (w.cpart && kb == 0) {
    float4* red = reinterpret_cast<float4*>(x3_smem);
    if (stager) red[(tid / (kX3DwCols / 4)) * (kX3DwCols / 4) + cg] = csum;
    __syncthreads();
    if (tid < kX3DwCols / 4 && tid * 4 < nt_blk * 16 && n_tile0 * 16 + tid * 4 < w.nout) {
      const float4 s0 = red[tid], s1 = red[kX3DwCols / 4 + tid], s2 = red[2 * (kX3DwCols / 4) + tid],
                   s3 = red[3 * (kX3DwCols / 4) + tid];
      float4 o;
      o.x = (s0.x + s1.x) + (s2.x + s3.x); o.y = (s0.y + s1.y) + (s2.y + s3.y);
      o.z = (s0.z + s1.z) + (s2.z + s3.z); o.w = (s0.w + s1.w) + (s2.w + s3.w);
      *reinterpret_cast<float4*>(w.cpart + (int64_t)slice * w.nout + n_tile0 * 16 + tid * 4) = o;
    }
  }
}

// host side -------------------------------------------------------------------------------------------------------
inline int x3_launch_split(const float* W, int64_t ldw, int K, int N, int trans, char* img, hipStream_t st) {
  const X3Cols c = x3_cols(N);
  const int nkt = (K + 31) / 32, np = c.nt * 16, ncb = 2 * c.ncb;
  const int64_t thr = (int64_t)ncb * nkt * np * 4;
  hipLaunchKernelGGL(x3_split_kernel, dim3((unsigned)((thr + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, W, ldw, K, N,
                     trans, img, nkt, np, ncb);
  return check_launch("x3_split_kernel");
}

// bytes of the ReLU bit mask of an [M, N] activation (one 64-bit word per row, column block and lane group)
inline size_t x3_relu_bits_bytes(int64_t M, int N) { return (size_t)M * (2 * x3_cols(N).ncb) * 4 * sizeof(unsigned long long); }

inline bool x3_shape_ok(int64_t M, int N, int K, int64_t lda, int64_t ldc, const void* A, const void* C) {
  return M > 0 && N > 0 && x3_cols(N).nt > 0 && N % 4 == 0 && K > 0 && K % 8 == 0 && lda % 4 == 0 && ldc % 4 == 0 &&
         ((uintptr_t)A % 16) == 0 && ((uintptr_t)C % 16) == 0;
}

template <int NT, int EPI, int WM, int WN>
inline int x3_launch_gemm_wm(int64_t M, int N, int K, const float* A, int64_t lda, const char* img, float* C, int64_t ldc,
                             const EpiArgs& e, int ncb, hipStream_t st) {
  static std::atomic<bool> attr_set{false};          // > 64 KB of dynamic LDS needs the attribute once per kernel
  constexpr int lds = 2 * WN * X3Geo<NT>::Stage;
  if (!attr_set.load(std::memory_order_acquire)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16x3_kernel<NT, EPI, WM, WN>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
      (void)hipGetLastError();
      set_error("gemm_bf16x3: %d B of dynamic LDS refused", lds);
      return REC_EHIP;
    }
    attr_set.store(true, std::memory_order_release);
  }
  const int64_t row_tiles = (M + kX3BM - 1) / kX3BM;
  if constexpr (WN == 1) {         // 1-D grid, column blocks of a row tile consecutive on one XCD (see the kernel)
    const int64_t grid = (row_tiles + 7) / 8 * 8 * ncb;
    if (grid >= (1ll << 31)) { set_error("gemm_bf16x3: too many workgroups"); return REC_ESHAPE; }
    hipLaunchKernelGGL((gemm_bf16x3_kernel<NT, EPI, WM, WN>), dim3((unsigned)grid), dim3(WM * WN * 64), lds, st, M, N, K, A,
                       lda, img, C, ldc, e, ncb);
  } else {
    hipLaunchKernelGGL((gemm_bf16x3_kernel<NT, EPI, WM, WN>), dim3((unsigned)row_tiles, (unsigned)(ncb / WN)),
                       dim3(WM * WN * 64), lds, st, M, N, K, A, lda, img, C, ldc, e, ncb);
  }
  return check_launch("gemm_bf16x3_kernel");
}

// Workgroup shapes (WM waves along M x WN along N, wave tile 128 / WM rows x NT column tiles):
//   4 x 1 (round 6, the MLP epilogues): four waves, 80 KB of LDS, two workgroups per CU that run out of phase — see
//         the kernel; REC_X3_WN=2 selects the round-5 shape for A/B runs;
//   4 x 2 (round 5): eight waves in one workgroup, two per SIMD in lockstep: 129 -> 119 us at 65 536 x 400 x 400 against
//   2 x 2: four waves with 64-row wave tiles, one per SIMD (512 registers) — kept for the CrossNet epilogues, which hold
//         two more operand tiles in registers and spill at a 256-register budget.
#ifndef REC_X3_WM
#define REC_X3_WM 4            // lab knob (tools/gemm_lab): 2 = four waves per block for every epilogue
#endif
inline int x3_wn_default() {
  static const int v = [] { const char* e = getenv("REC_X3_WN"); return e && *e == '2' ? 2 : 1; }();
  return v;
}
template <int NT, int EPI>
inline int x3_launch_gemm_cfg(int64_t M, int N, int K, const float* A, int64_t lda, const char* img, float* C, int64_t ldc,
                              const EpiArgs& e, int ncb, hipStream_t st) {
  if constexpr (EpiUses<EPI>::aux1) {                           // CROSS / ADD / MOE
    return x3_launch_gemm_wm<NT, EPI, 2, 2>(M, N, K, A, lda, img, C, ldc, e, ncb, st);
  } else {
    if (REC_X3_WM == 4 && x3_wn_default() == 1)
      return x3_launch_gemm_wm<NT, EPI, 4, 1>(M, N, K, A, lda, img, C, ldc, e, ncb, st);
    return x3_launch_gemm_wm<NT, EPI, REC_X3_WM, 2>(M, N, K, A, lda, img, C, ldc, e, ncb, st);
  }
}

template <int EPI>
inline int x3_launch_gemm_epi(int64_t M, int N, int K, const float* A, int64_t lda, const char* img, float* C, int64_t ldc,
                              const EpiArgs& e, hipStream_t st) {
  const X3Cols c = x3_cols(N);
  switch (c.nt) {
    case 13: return x3_launch_gemm_cfg<13, EPI>(M, N, K, A, lda, img, C, ldc, e, 2 * c.ncb, st);
    case 8: return x3_launch_gemm_cfg<8, EPI>(M, N, K, A, lda, img, C, ldc, e, 2 * c.ncb, st);
    case 7: return x3_launch_gemm_cfg<7, EPI>(M, N, K, A, lda, img, C, ldc, e, 2 * c.ncb, st);
    default: set_error("gemm_bf16x3: no column blocking for N = %d", N); return REC_ESHAPE;
  }
}

inline bool x3_epilogue_ok(int epi) { return epi >= REC_EPI_NONE && epi <= REC_EPI_DTANH; }     // every rec_gemm_f32 epilogue

inline int x3_launch_gemm(int epi, int64_t M, int N, int K, const float* A, int64_t lda, const char* img, float* C,
                          int64_t ldc, const EpiArgs& e, hipStream_t st) {
  switch (epi) {
    case REC_EPI_NONE: return x3_launch_gemm_epi<REC_EPI_NONE>(M, N, K, A, lda, img, C, ldc, e, st);
    case REC_EPI_BIAS: return x3_launch_gemm_epi<REC_EPI_BIAS>(M, N, K, A, lda, img, C, ldc, e, st);
    case REC_EPI_BIAS_RELU: return x3_launch_gemm_epi<REC_EPI_BIAS_RELU>(M, N, K, A, lda, img, C, ldc, e, st);
    case REC_EPI_RELU_MASK: return x3_launch_gemm_epi<REC_EPI_RELU_MASK>(M, N, K, A, lda, img, C, ldc, e, st);
    case REC_EPI_CROSS: return x3_launch_gemm_epi<REC_EPI_CROSS>(M, N, K, A, lda, img, C, ldc, e, st);
    case REC_EPI_BIAS_SIGMOID: return x3_launch_gemm_epi<REC_EPI_BIAS_SIGMOID>(M, N, K, A, lda, img, C, ldc, e, st);
    case REC_EPI_BIAS_TANH: return x3_launch_gemm_epi<REC_EPI_BIAS_TANH>(M, N, K, A, lda, img, C, ldc, e, st);
    case REC_EPI_ADD: return x3_launch_gemm_epi<REC_EPI_ADD>(M, N, K, A, lda, img, C, ldc, e, st);
    case REC_EPI_MOE: return x3_launch_gemm_epi<REC_EPI_MOE>(M, N, K, A, lda, img, C, ldc, e, st);
    case REC_EPI_DSIGMOID: return x3_launch_gemm_epi<REC_EPI_DSIGMOID>(M, N, K, A, lda, img, C, ldc, e, st);
    case REC_EPI_DTANH: return x3_launch_gemm_epi<REC_EPI_DTANH>(M, N, K, A, lda, img, C, ldc, e, st);
    default: set_error("gemm_bf16x3: epilogue %d not built", epi); return REC_EINVAL;
  }
}

// dW plan: output blocks of <= 13 x 13 tiles, as equal as they come; slices = one resident round of blocks
struct X3DwPlan { int kb_tiles, nb_tiles, kblocks, nblocks, slices, steps_per_slice; };
inline bool x3_dw_plan(int kin, int nout, int64_t rows, int cus, X3DwPlan* p) {
  if (kin % 4 || nout % 4 || rows % 32 || kin < 16 || nout < 16) return false;
  const int kt = (kin + 15) / 16, nt = (nout + 15) / 16;
  p->kblocks = (kt + 12) / 13; p->nblocks = (nt + 12) / 13;
  p->kb_tiles = (kt + p->kblocks - 1) / p->kblocks; p->nb_tiles = (nt + p->nblocks - 1) / p->nblocks;
  const int nob = p->kblocks * p->nblocks;
  int slices = cus / nob;
  slices -= slices % 8;
  if (slices < 8) return false;
  const int64_t steps = rows / 32;
  p->steps_per_slice = (int)((steps + slices - 1) / slices);
  if (p->steps_per_slice < 4) return false;
  p->slices = (int)((steps + p->steps_per_slice - 1) / p->steps_per_slice);
  p->slices = (p->slices + 7) / 8 * 8;             // whole groups of 8 (slices behind the rows write zeros)
  return true;
}
inline int x3_launch_dw(const X3DwPlan& pl, int kin, int nout, int64_t rows, const float* X, int64_t ldx, const float* G,
                        int64_t ldg, float* P, int64_t ldp, float* cpart, hipStream_t st) {
  static std::atomic<bool> attr_set{false};
  constexpr int lds = 2 * kX3Stage + 4096;        // + a 16-B slot per thread for the patch-less threads' stores
  if (!attr_set.load(std::memory_order_acquire)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16x3_dw_kernel<7, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16x3_dw_kernel<5, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16x3_dw_kernel<7, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16x3_dw_kernel<5, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
      (void)hipGetLastError();
      set_error("gemm_bf16x3_dw: %d B of dynamic LDS refused", lds);
      return REC_EHIP;
    }
    attr_set.store(true, std::memory_order_release);
  }
  X3DwArgs w{X, ldx, G, ldg, rows, kin, nout, pl.kb_tiles, pl.nb_tiles, pl.kblocks, pl.nblocks, pl.slices,
             pl.steps_per_slice, P, ldp, cpart};
  const unsigned grid = (unsigned)(pl.slices * pl.kblocks * pl.nblocks);
  // REC_X3_DW_PIPE=0: the round-5 schedule (a column's six LDS stores in one burst behind its conversion) for A/B runs
  static const bool pipe = [] { const char* v = getenv("REC_X3_DW_PIPE"); return v && *v ? *v != '0' : REC_X3_DW_PIPE_STORES != 0; }();
  const bool p5 = (pl.kb_tiles + 1) / 2 <= 5;
  if (p5 && pipe) hipLaunchKernelGGL((gemm_bf16x3_dw_kernel<5, true>), dim3(grid), dim3(256), lds, st, w);
  else if (p5) hipLaunchKernelGGL((gemm_bf16x3_dw_kernel<5, false>), dim3(grid), dim3(256), lds, st, w);
  else if (pipe) hipLaunchKernelGGL((gemm_bf16x3_dw_kernel<7, true>), dim3(grid), dim3(256), lds, st, w);
  else hipLaunchKernelGGL((gemm_bf16x3_dw_kernel<7, false>), dim3(grid), dim3(256), lds, st, w);
  return check_launch("gemm_bf16x3_dw_kernel");
}

}  // namespace rec
