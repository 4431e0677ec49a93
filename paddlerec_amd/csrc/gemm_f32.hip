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
// Tiling: 256 threads = 4 waves per block; block tile BM x BN, K step 16 through double-buffered
// LDS (global -> registers -> LDS, next tile's global loads issued before this tile's MFMAs).
// A wave owns MT x NT MFMA tiles of 16x16; per K step it reads MT A-fragments with one
// ds_read_b128 each (lane group g = lane>>4 takes k = 4g..4g+3, so MFMA step s multiplies the
// k = 4g+s slices — every k exactly once, in a fixed order) and NT*4 B values with ds_read_b32.
//   config "128x128": waves 2x2, wave tile 64x64  (general shapes)
//   config "128x80" : waves 4x1, wave tile 32x80  (N = 400: the DeepFM MLP width, no column waste)
// Blocks are numbered so that the N-tiles of one M-tile run on the same XCD back to back (its A
// tile is fetched from HBM once and re-read from that XCD's L2).
// Split-K (trans_a GEMMs with K = batch): partial tiles go to the workspace and are summed in a
// fixed order by a second kernel (deterministic).
// Tried and dropped (measured on MI355X, [65536x624]x[624x400]): a barrier-free variant in which every wave
// stages its own A/B K-slices (76 TF vs 96 TF — the 4x re-read of B costs more than the barriers), static
// s_setprio staggering of co-resident blocks (no effect).
#include "rec_common.h"

namespace rec {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int kBK = 16;
constexpr int kLdsPadA = 4;   // A_lds[BM][BK+4]  : 80-B rows, 16-B aligned
constexpr int kLdsPadB = 4;   // B_lds[BK][BN+4]  : (BN+4) % 8 == 4 -> the two 32-lane halves hit disjoint banks

// ---------------------------------------------------------------------------------- tile loader
// Logical tile T[R][C].  Memory is either contiguous along C ("N": element (r,c) at p[r*ld + c]) or
// along R ("T": element (r,c) at p[c*ld + r]).  LDS keeps it as [R][LDS_LD] — or, with LDST, as
// [C][LDS_LD] (the memory order of a "T" tile, so its float4s are stored whole instead of being
// scattered across LDS rows, which costs 16-way bank conflicts for R = 128).
template <int R, int C, int LDS_LD, bool MEMT, bool LDST = false, int NTHR = kBlock>
struct TileLoader {
  static constexpr int kVecs = R * C / 4;
  static constexpr int kPerThread = (kVecs + NTHR - 1) / NTHR;
  float4 stage[kPerThread];
  int tid;   // index of this thread among the NTHR cooperating ones
  __device__ __forceinline__ explicit TileLoader(int t) : tid(t) {}

  // FAST: the tile is fully inside the matrix and 16-B aligned — straight float4 loads, no checks
  template <bool FAST>
  __device__ __forceinline__ void load(const float* __restrict__ p, int64_t ld, int64_t r0,
                                       int64_t c0, int64_t rmax, int64_t cmax, bool vec_ok) {
#pragma unroll
    for (int it = 0; it < kPerThread; ++it) {
      const int v = tid + it * NTHR;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (FAST) {
        if (kVecs % NTHR == 0 || v < kVecs) {
          if (!MEMT) {
            const int r = v / (C / 4), c4 = (v % (C / 4)) * 4;
            x = *reinterpret_cast<const float4*>(p + (r0 + r) * ld + (c0 + c4));
          } else {
            const int c = v / (R / 4), r4 = (v % (R / 4)) * 4;
            x = *reinterpret_cast<const float4*>(p + (c0 + c) * ld + (r0 + r4));
          }
        }
      } else if (kVecs % NTHR == 0 || v < kVecs) {
        if (!MEMT) {
          const int r = v / (C / 4), c4 = (v % (C / 4)) * 4;
          const int64_t gr = r0 + r, gc = c0 + c4;
          if (gr < rmax) {
            const float* q = p + gr * ld + gc;
            if (vec_ok && gc + 3 < cmax) {
              x = *reinterpret_cast<const float4*>(q);
            } else {
              if (gc + 0 < cmax) x.x = q[0];
              if (gc + 1 < cmax) x.y = q[1];
              if (gc + 2 < cmax) x.z = q[2];
              if (gc + 3 < cmax) x.w = q[3];
            }
          }
        } else {
          const int c = v / (R / 4), r4 = (v % (R / 4)) * 4;
          const int64_t gr = r0 + r4, gc = c0 + c;
          if (gc < cmax) {
            const float* q = p + gc * ld + gr;
            if (vec_ok && gr + 3 < rmax) {
              x = *reinterpret_cast<const float4*>(q);
            } else {
              if (gr + 0 < rmax) x.x = q[0];
              if (gr + 1 < rmax) x.y = q[1];
              if (gr + 2 < rmax) x.z = q[2];
              if (gr + 3 < rmax) x.w = q[3];
            }
          }
        }
      }
      stage[it] = x;
    }
  }

  __device__ __forceinline__ void store(float* __restrict__ lds) const {
#pragma unroll
    for (int it = 0; it < kPerThread; ++it) {
      const int v = tid + it * NTHR;
      if (kVecs % NTHR == 0 || v < kVecs) {
        if (!MEMT) {
          const int r = v / (C / 4), c4 = (v % (C / 4)) * 4;
          *reinterpret_cast<float4*>(lds + r * LDS_LD + c4) = stage[it];
        } else if (LDST) {
          const int c = v / (R / 4), r4 = (v % (R / 4)) * 4;
          *reinterpret_cast<float4*>(lds + c * LDS_LD + r4) = stage[it];
        } else {
          const int c = v / (R / 4), r4 = (v % (R / 4)) * 4;
          lds[(r4 + 0) * LDS_LD + c] = stage[it].x;
          lds[(r4 + 1) * LDS_LD + c] = stage[it].y;
          lds[(r4 + 2) * LDS_LD + c] = stage[it].z;
          lds[(r4 + 3) * LDS_LD + c] = stage[it].w;
        }
      }
    }
  }
};

// ------------------------------------------------------------------------------------ epilogues
struct EpiArgs {
  const float* bias;       // [N] or null
  const float* aux0;       // [M,ld0]: RELU_MASK source / CROSS, MOE X_0 / second ADD operand
  const float* aux1;       // [M,ld1]: CROSS, MOE X_l / ADD operand
  const float* row_scale;  // [M] (stride rs_stride): MOE gate probability of this expert
  float* out2;             // [M,ldc] or null: CROSS also stores u = acc + bias (saved for backward)
  int ld0, ld1, rs_stride, ld2;
};

template <int EPI>
__device__ __forceinline__ float apply_epi(float acc, int64_t i, int j, const EpiArgs& e) {
  if (EPI == REC_EPI_NONE) return acc;
  if (EPI == REC_EPI_BIAS) return acc + e.bias[j];
  if (EPI == REC_EPI_BIAS_RELU) return fmaxf(acc + e.bias[j], 0.f);
  if (EPI == REC_EPI_RELU_MASK) return e.aux0[i * e.ld0 + j] > 0.f ? acc : 0.f;
  if (EPI == REC_EPI_CROSS) return e.aux1[i * e.ld1 + j] + e.aux0[i * e.ld0 + j] * (acc + e.bias[j]);
  if (EPI == REC_EPI_DTANH) {
    const float a = e.aux0[i * e.ld0 + j];
    return acc * (1.f - a * a);
  }
  if (EPI == REC_EPI_DSIGMOID) {
    const float a = e.aux0[i * e.ld0 + j];
    return acc * a * (1.f - a);
  }
  if (EPI == REC_EPI_MOE)
    return e.aux1[i * e.ld1 + j] +
           e.aux0[i * e.ld0 + j] * (e.row_scale[i * e.rs_stride] * (acc + e.bias[j]));
  if (EPI == REC_EPI_BIAS_SIGMOID) return 1.f / (1.f + expf(-(acc + e.bias[j])));
  if (EPI == REC_EPI_BIAS_TANH) return tanhf(acc + (e.bias ? e.bias[j] : 0.f));
  if (EPI == REC_EPI_ADD)
    return acc + (e.bias ? e.bias[j] : 0.f) + e.aux1[i * e.ld1 + j] +
           (e.aux0 ? e.aux0[i * e.ld0 + j] : 0.f);
  return acc;
}

// --------------------------------------------------------------------------------------- kernel
// Blocks per CU each config is built for (launch bound = waves per SIMD for 256-thread blocks): the
// narrow config at 5 makes the 2560 tiles of a [65536 x 400] product exactly two full rounds.
template <int BN>
constexpr int gemm_blocks_per_cu() { return BN == 80 ? 5 : 4; }

template <int BM, int BN, int WAVES_M, int WAVES_N, bool TA, bool TB, int EPI>
__global__ __launch_bounds__(kBlock, gemm_blocks_per_cu<BN>()) void gemm_f32_kernel(
    int64_t M, int N, int K, const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
    int64_t ldb, float* __restrict__ C, int64_t ldc, EpiArgs epi, int tiles_n, int64_t tiles_total,
    int k_chunk, bool vec_a, bool vec_b, float* __restrict__ partial,
    float* __restrict__ colsum_partial, int splits_in_x) {
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MT = WTM / 16, NT = WTN / 16;
  // A in LDS: [BM][BK+4] (b128 fragment reads) — or k-major [BK][BM+4] when A is stored [K,M]
  constexpr int LDA_S = TA ? BM + kLdsPadB : kBK + kLdsPadA, LDB_S = BN + kLdsPadB;
  constexpr int A_ELEMS = TA ? kBK * LDA_S : BM * LDA_S;
  __shared__ __attribute__((aligned(16))) float As[2][A_ELEMS];
  __shared__ __attribute__((aligned(16))) float Bs[2][kBK * LDB_S];

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

  const int lane = threadIdx.x % kWave;
  const int wave = threadIdx.x / kWave;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int li = lane & 15, g = lane >> 4;

  TileLoader<BM, kBK, LDA_S, TA, TA> la(threadIdx.x);
  TileLoader<kBK, BN, LDB_S, TB> lb(threadIdx.x);

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // column sums of op(B) over this block's K range (bias gradient when B = dY): the blocks of the
  // first M-tile add up the B tiles they stage anyway
  const bool do_colsum = colsum_partial != nullptr && tm == 0;
  float csum = 0.f;

  const int nkt = (k_end - k_begin + kBK - 1) / kBK;
  // interior blocks (tile fully inside, K range a multiple of 16, aligned operands) take the
  // check-free loader; edge blocks the predicated one.  The choice is block-uniform.
  const bool interior = vec_a && vec_b && m0 + BM <= M && n0 + BN <= N &&
                        (k_end - k_begin) % kBK == 0;
  auto mainloop = [&](auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
  if (nkt > 0) {
    la.template load<FAST>(A, lda, m0, k_begin, M, k_end, vec_a);
    lb.template load<FAST>(B, ldb, k_begin, n0, k_end, N, vec_b);
    la.store(As[0]);
    lb.store(Bs[0]);
  }
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nkt) {   // next tile's global loads fly under this tile's MFMAs
      const int k0 = k_begin + (kt + 1) * kBK;
      la.template load<FAST>(A, lda, m0, k0, M, k_end, vec_a);
      lb.template load<FAST>(B, ldb, k0, n0, k_end, N, vec_b);
    }
    const float* bs = Bs[cur] + (g * 4) * LDB_S + wn * WTN + li;
    float4 af[MT];
    float bf[NT][4];
    if (TA) {
      const float* as = As[cur] + (g * 4) * LDA_S + wm * WTM + li;
#pragma unroll
      for (int a = 0; a < MT; ++a)
        af[a] = make_float4(as[a * 16], as[LDA_S + a * 16], as[2 * LDA_S + a * 16],
                            as[3 * LDA_S + a * 16]);
    } else {
      const float* as = As[cur] + (wm * WTM + li) * LDA_S + g * 4;
#pragma unroll
      for (int a = 0; a < MT; ++a) af[a] = *reinterpret_cast<const float4*>(as + a * 16 * LDA_S);
    }
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int s = 0; s < 4; ++s) bf[b][s] = bs[s * LDB_S + b * 16];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int a = 0; a < MT; ++a) {
        const float av = s == 0 ? af[a].x : s == 1 ? af[a].y : s == 2 ? af[a].z : af[a].w;
#pragma unroll
        for (int b = 0; b < NT; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bf[b][s], acc[a][b], 0, 0, 0);
      }
    }
    if (do_colsum && threadIdx.x < BN) {
#pragma unroll
      for (int kk = 0; kk < kBK; ++kk) csum += Bs[cur][kk * LDB_S + threadIdx.x];
    }
    if (kt + 1 < nkt) {
      la.store(As[cur ^ 1]);
      lb.store(Bs[cur ^ 1]);
    }
    __syncthreads();
  }
  };
  if (interior) mainloop(std::integral_constant<bool, true>{});
  else mainloop(std::integral_constant<bool, false>{});
  if (do_colsum && threadIdx.x < BN && n0 + (int)threadIdx.x < N)
    colsum_partial[(int64_t)kz * N + n0 + threadIdx.x] = csum;

  // C/D layout of v_mfma_f32_16x16x4_f32: col = lane & 15, row = (lane >> 4) * 4 + reg
  float* out = partial ? partial + (int64_t)kz * M * ldc : C;
#pragma unroll
  for (int a = 0; a < MT; ++a) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t i = m0 + wm * WTM + a * 16 + g * 4 + r;
      if (i < M) {
#pragma unroll
        for (int b = 0; b < NT; ++b) {
          const int j = n0 + wn * WTN + b * 16 + li;
          if (j < N) {
            const float v = acc[a][b][r];
            out[i * ldc + j] = partial ? v : apply_epi<EPI>(v, i, j, epi);
            if (EPI == REC_EPI_CROSS && !partial && epi.out2) epi.out2[i * epi.ld2 + j] = v + epi.bias[j];
          }
        }
      }
    }
  }
}

// C = epi(sum_z partial[z]) in ascending z (fixed order)
template <int EPI>
__global__ __launch_bounds__(kBlock) void splitk_reduce_kernel(int64_t M, int N, int64_t ldc,
                                                               int splits,
                                                               const float* __restrict__ partial,
                                                               float* __restrict__ C, EpiArgs epi) {
  const int64_t total = M * N;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * kBlock) {
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
  if (j >= N) return;
  float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int z = 0;
  for (; z + 8 <= splits; z += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) a8[u] += partial[(int64_t)(z + u) * N + j];
  }
  for (; z < splits; ++z) a8[z & 7] += partial[(int64_t)z * N + j];
  out[j] = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
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
__global__ __launch_bounds__(kBlock) void colsum_final_kernel(int nblk, int N,
                                                              const float* __restrict__ partial,
                                                              float* __restrict__ out) {
  const int j = blockIdx.x * kBlock + threadIdx.x;
  if (j >= N) return;
  float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int b = 0;
  for (; b + 8 <= nblk; b += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) a8[u] += partial[(int64_t)(b + u) * N + j];
  }
  for (; b < nblk; ++b) a8[b & 7] += partial[(int64_t)b * N + j];
  out[j] = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
}

struct GemmPlan {
  bool narrow;      // 128x80 config
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

static GemmPlan plan_gemm(const rec_gemm_desc* d, int num_cus = kNumCU) {
  GemmPlan p;
  const int N = d->n;
  // 128x80 when it wastes fewer columns than 128x128 (N = 400 -> 5 x 80 exactly)
  const int w128 = (N + 127) / 128 * 128 - N, w80 = (N + 79) / 80 * 80 - N;
  p.narrow = w80 < w128;
  const int bn = p.narrow ? 80 : 128;
  p.tiles_n = (N + bn - 1) / bn;
  p.tiles_m = (d->m + 127) / 128;
  p.tiles_total = p.tiles_m * p.tiles_n;
  int splits = d->split_k;
  const int nkt = (d->k + kBK - 1) / kBK;
  if (splits <= 0) {  // auto: split K when the output alone cannot fill the chip
    splits = 1;
    const int64_t capacity = (int64_t)num_cus * (p.narrow ? gemm_blocks_per_cu<80>() : gemm_blocks_per_cu<128>());
    if (p.tiles_total * 2 <= capacity) {
      // as many splits as still fit in ONE resident round (one block more would double the time)
      int64_t want = capacity / p.tiles_total;
      if (want > nkt / 8) want = nkt / 8;   // keep >= 8 K-tiles (128 k) per split
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

template <int BM, int BN, int WM_, int WN_, bool TA, bool TB, int EPI>
static void launch_one(const rec_gemm_desc* d, const GemmPlan& p, const float* A, const float* B,
                       float* C, const EpiArgs& e, float* partial, float* cpart, hipStream_t st) {
  const bool vec_a = (d->lda % 4 == 0) && (((uintptr_t)A) % 16 == 0);
  const bool vec_b = (d->ldb % 4 == 0) && (((uintptr_t)B) % 16 == 0);
  const bool fold = p.splits >= 8 && p.splits % 8 == 0 && p.tiles_total * p.splits < (1ll << 31);
  dim3 grid(fold ? (unsigned)(p.tiles_total * p.splits) : (unsigned)p.tiles_total,
            fold ? 1u : (unsigned)p.splits);
  hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM_, WN_, TA, TB, EPI>), grid, dim3(kBlock), 0, st,
                     d->m, d->n, d->k, A, (int64_t)d->lda, B, (int64_t)d->ldb, C, (int64_t)d->ldc,
                     e, p.tiles_n, p.tiles_total, p.k_chunk, vec_a, vec_b, partial, cpart,
                     fold ? p.splits : 1);
}

template <bool TA, bool TB, int EPI>
static void launch_cfg(const rec_gemm_desc* d, const GemmPlan& p, const float* A, const float* B,
                       float* C, const EpiArgs& e, float* partial, float* cpart, hipStream_t st) {
  if (p.narrow) launch_one<128, 80, 4, 1, TA, TB, EPI>(d, p, A, B, C, e, partial, cpart, st);
  else launch_one<128, 128, 2, 2, TA, TB, EPI>(d, p, A, B, C, e, partial, cpart, st);
}

template <int EPI>
static void launch_epi(const rec_gemm_desc* d, const GemmPlan& p, const float* A, const float* B,
                       float* C, const EpiArgs& e, float* partial, float* cpart, hipStream_t st) {
  if (!d->trans_a && !d->trans_b) launch_cfg<false, false, EPI>(d, p, A, B, C, e, partial, cpart, st);
  else if (!d->trans_a && d->trans_b) launch_cfg<false, true, EPI>(d, p, A, B, C, e, partial, cpart, st);
  else if (d->trans_a && !d->trans_b) launch_cfg<true, false, EPI>(d, p, A, B, C, e, partial, cpart, st);
  else launch_cfg<true, true, EPI>(d, p, A, B, C, e, partial, cpart, st);
}

template <int EPI>
static void launch_reduce(const rec_gemm_desc* d, const GemmPlan& p, const float* partial, float* C,
                          const EpiArgs& e, hipStream_t st) {
  int64_t grid = (d->m * d->n + kBlock - 1) / kBlock;
  if (grid > kNumCU * 8) grid = kNumCU * 8;
  hipLaunchKernelGGL(splitk_reduce_kernel<EPI>, dim3((unsigned)grid), dim3(kBlock), 0, st, d->m,
                     d->n, (int64_t)d->ldc, p.splits, partial, C, e);
}

}  // namespace rec

using namespace rec;

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
  REC_REQUIRE(!(epi == REC_EPI_BIAS || epi == REC_EPI_BIAS_RELU || epi == REC_EPI_CROSS ||
                epi == REC_EPI_BIAS_SIGMOID || epi == REC_EPI_MOE) || bias, REC_EINVAL,
              "epilogue needs bias");
  REC_REQUIRE(!(epi == REC_EPI_RELU_MASK || epi == REC_EPI_CROSS || epi == REC_EPI_MOE ||
                epi == REC_EPI_DSIGMOID || epi == REC_EPI_DTANH) ||
                  (aux0 && ld_aux0 >= desc->n), REC_EINVAL, "epilogue needs aux0");
  REC_REQUIRE(!(epi == REC_EPI_CROSS || epi == REC_EPI_ADD || epi == REC_EPI_MOE) ||
                  (aux1 && ld_aux1 >= desc->n), REC_EINVAL, "epilogue needs aux1");
  REC_REQUIRE(epi != REC_EPI_ADD || !aux0 || ld_aux0 >= desc->n, REC_EINVAL, "bad ld_aux0");
  REC_REQUIRE(epi != REC_EPI_MOE || (x->row_scale && x->row_scale_stride >= 1), REC_EINVAL,
              "epilogue needs row_scale");
  hipStream_t st = (hipStream_t)stream;
  EpiArgs e{bias, aux0, aux1, x->row_scale, x->out2, ld_aux0, ld_aux1, x->row_scale_stride, x->ld_out2};
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
    size_t need = 0;
    rec_gemm_f32_workspace_bytes(desc, &need);
    REC_REQUIRE(workspace && workspace_bytes >= need, REC_EWORKSPACE, "workspace %zu < %zu", workspace_bytes, need);
    const int z = (int)((desc->k + kSkinnyKC - 1) / kSkinnyKC);
    float* part = (float*)workspace;
    float* cpart2 = (float*)((char*)workspace + align_up((size_t)z * desc->m * desc->ldc * sizeof(float), 256));
    hipLaunchKernelGGL(skinny_dw_kernel, dim3(z), dim3(kBlock), 0, st, (int)desc->m, desc->n, (int64_t)desc->k, A,
                       (int64_t)desc->lda, B, (int64_t)desc->ldb, (int64_t)desc->ldc, part,
                       b_colsum ? cpart2 : nullptr);
    GemmPlan sp;
    sp.splits = z;
    launch_reduce<REC_EPI_NONE>(desc, sp, part, C, e, st);
    if (b_colsum)
      hipLaunchKernelGGL(colsum_reduce_kernel, dim3((desc->n + kBlock - 1) / kBlock), dim3(kBlock), 0, st,
                         desc->n, z, (const float*)cpart2, b_colsum);
    return check_launch("rec_gemm_f32 (skinny dW)");
  }
  const GemmPlan p = plan_gemm(desc);
  REC_REQUIRE(p.tiles_total < (1ll << 31), REC_ESHAPE, "too many tiles");
  float* partial = nullptr;
  float* cpart = nullptr;
  if (p.splits > 1 || b_colsum) {
    size_t need = 0;
    rec_gemm_f32_workspace_bytes(desc, &need);
    REC_REQUIRE(workspace && workspace_bytes >= need, REC_EWORKSPACE, "workspace %zu < %zu",
                workspace_bytes, need);
    size_t off = 0;
    if (p.splits > 1) {
      partial = (float*)workspace;
      off = align_up((size_t)p.splits * desc->m * desc->ldc * sizeof(float), 256);
    }
    if (b_colsum) cpart = (float*)((char*)workspace + off);
  }
#define REC_EPI_CASE(E)                                                   \
  case E:                                                                 \
    if (partial) {                                                        \
      launch_epi<REC_EPI_NONE>(desc, p, A, B, C, e, partial, cpart, st);  \
      launch_reduce<E>(desc, p, partial, C, e, st);                       \
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
  if (b_colsum)
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
  hipLaunchKernelGGL(colsum_final_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st,
                     nblk, n, (const float*)workspace, out);
  return check_launch("rec_colsum");
}
