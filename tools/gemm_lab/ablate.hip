// Where does the time of gemm_f32_kernel go?  (measurement tool, not part of librecengine.so)
// The steady-state loop of paddlerec_amd/csrc/gemm_f32.hip — same tile loader (branch-free interior form), same LDS
// images, same fragment reads / MFMA order, same sched_barrier placement — with ONE piece removed per variant.  The
// results of the ablated variants are wrong on purpose; only ABL 0 is checked (it must match the engine's rate).
//   ABL 0  nothing removed
//   ABL 1  every block loads tile (0, 0) of A and B: the memory system serves everything from cache
//   ABL 2  no global loads after the first tile (the staged registers keep their values; LDS stores stay)
//   ABL 3  no LDS stores after the first tile (loads stay: their values are folded into a sink)
//   ABL 4  no barrier in the loop
//   ABL 5  no LDS fragment reads in the loop (the first tile's fragments are multiplied every time)
//   ABL 6  2 + 3 + 4: only fragment reads and MFMAs are left
//   ABL 7  2 + 3 + 4 + 5: MFMAs only
//   8 / 9  candidates with correct results: pipe_kernel PIPE 0 / 1 (see there)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared tools/gemm_lab/ablate.hip -o tools/gemm_lab/_build/libgemmablate.so
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace abl {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int kBK = 16;

template <int R, int C, bool MEMT, int NTHR>
struct TileLoader {   // interior tiles only (MODE 0 of the engine's loader)
  static constexpr int INNER = MEMT ? R : C, OUTER = MEMT ? C : R;
  static constexpr int kVecs = R * C / 4;
  static constexpr int kPerThread = (kVecs + NTHR - 1) / NTHR;
  float4 stage[kPerThread];
  __device__ __forceinline__ void load(const float* __restrict__ p, int64_t ld, int64_t r0, int64_t c0, int tid) {
#pragma unroll
    for (int it = 0; it < kPerThread; ++it) {
      const int v0 = tid + it * NTHR;
      const int v = (kVecs % NTHR == 0 || v0 < kVecs) ? v0 : kVecs - 1;
      const int o = v / (INNER / 4), i4 = (v % (INNER / 4)) * 4;
      const int64_t go = (MEMT ? c0 : r0) + o, gi = (MEMT ? r0 : c0) + i4;
      stage[it] = *reinterpret_cast<const float4*>(p + go * ld + gi);
    }
  }
  template <bool TRANSPOSE>
  __device__ __forceinline__ void store(float* __restrict__ lds, int tid) const {
#pragma unroll
    for (int it = 0; it < kPerThread; ++it) {
      const int v = tid + it * NTHR;
      if (kVecs % NTHR == 0 || v < kVecs) {
        const int o = v / (INNER / 4), i4 = (v % (INNER / 4)) * 4;
        if (!TRANSPOSE) {
          *reinterpret_cast<float4*>(lds + o * (INNER + 4) + i4) = stage[it];
        } else {
          lds[(i4 + 0) * (OUTER + 4) + o] = stage[it].x;
          lds[(i4 + 1) * (OUTER + 4) + o] = stage[it].y;
          lds[(i4 + 2) * (OUTER + 4) + o] = stage[it].z;
          lds[(i4 + 3) * (OUTER + 4) + o] = stage[it].w;
        }
      }
    }
  }
  __device__ __forceinline__ float fold() const {
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < kPerThread; ++it) s += stage[it].x + stage[it].y + stage[it].z + stage[it].w;
    return s;
  }
};

template <int BM, int BN, int WAVES_M, int WAVES_N, int OCC, bool TA, bool TB, int ABL>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, (OCC * WAVES_M * WAVES_N + 3) / 4) void ablate_kernel(
    int64_t M, int N, int K, const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
    float* __restrict__ C, int64_t ldc, int tiles_n, int64_t tiles_total, int k_chunk, float* __restrict__ partial,
    int splits_in_x) {
  constexpr bool NO_LOAD = ABL == 2 || ABL == 6 || ABL == 7, NO_STORE = ABL == 3 || ABL == 6 || ABL == 7;
  constexpr bool NO_BARRIER = ABL == 4 || ABL == 6 || ABL == 7, NO_FRAG = ABL == 5 || ABL == 7;
  constexpr int NTHR = WAVES_M * WAVES_N * 64;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MT = WTM / 16, NT = WTN / 16;
  constexpr int LDA_S = TA ? BM + 4 : kBK + 4, LDB_S = BN + 4;
  constexpr int A_ELEMS = TA ? kBK * LDA_S : BM * LDA_S, B_ELEMS = kBK * LDB_S;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * A_ELEMS;
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
  const int64_t lm0 = ABL == 1 ? 0 : m0;      // where the LOADS come from
  const int ln0 = ABL == 1 ? 0 : n0;
  const int k_begin = ABL == 1 ? 0 : kz * k_chunk;
  const int nkt = k_chunk / kBK;
  const int tid = threadIdx.x;
  const int lane = tid % 64, wave = tid / 64;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int li = lane & 15, g = lane >> 4;
  TileLoader<BM, kBK, TA, NTHR> la;
  TileLoader<kBK, BN, TB, NTHR> lb;
  f32x4_t acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float sink = 0.f;
  float af_keep[MT][4], bf_keep[NT][4];      // ABL 5 / 7 only: the first tile's fragments, multiplied every time
  auto frags = [&](int cur, float (&af)[MT][4], float (&bf)[NT][4]) {
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
  auto mfmas = [&](float (&af)[MT][4], float (&bf)[NT][4]) {
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a][s_], bf[b][s_], acc[a][b], 0, 0, 0);
  };
  auto compute = [&](int cur) {
    if constexpr (NO_FRAG) {
      mfmas(af_keep, bf_keep);
    } else {                                   // fragments local to the tile, as in the engine's loop
      float af[MT][4], bf[NT][4];
      frags(cur, af, bf);
      mfmas(af, bf);
    }
  };
  la.load(A, lda, lm0, k_begin, tid);
  lb.load(B, ldb, k_begin, ln0, tid);
  la.template store<false>(As, tid);
  lb.template store<TB>(Bs, tid);
  __syncthreads();
  if (NO_FRAG) frags(0, af_keep, bf_keep);
  int kt = 0;
  for (; kt + 1 < nkt; ++kt) {
    if (!NO_LOAD) {
      const int k0 = k_begin + (kt + 1) * kBK;
      la.load(A, lda, lm0, k0, tid);
      lb.load(B, ldb, k0, ln0, tid);
    }
    __builtin_amdgcn_sched_barrier(0);
    compute(kt & 1);
    __builtin_amdgcn_sched_barrier(0);
    if (!NO_STORE) {
      la.template store<false>(As + ((kt + 1) & 1) * A_ELEMS, tid);
      lb.template store<TB>(Bs + ((kt + 1) & 1) * B_ELEMS, tid);
    } else if (!NO_LOAD) {
      sink += la.fold() + lb.fold();
    }
    if (!NO_BARRIER) __syncthreads();
  }
  compute(kt & 1);
  float* out = partial ? partial + (int64_t)kz * M * ldc : C;
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t i = m0 + wm * WTM + a * 16 + g * 4 + r;
#pragma unroll
      for (int b = 0; b < NT; ++b) {
        const int j = n0 + wn * WTN + b * 16 + li;
        if (i < M && j < N) out[i * ldc + j] = acc[a][b][r] + (sink == 12345.678f ? 1.f : 0.f);
      }
    }
}

template <int BM, int BN, int WM, int WN, int OCC, bool TA, bool TB, int ABL>
static int launch(int64_t M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                  int splits, float* partial, hipStream_t st) {
  if (M % BM || N % BN || K % (kBK * splits)) return 1;     // interior tiles and whole K tiles only
  const int tiles_n = N / BN;
  const int64_t tiles_total = (M / BM) * tiles_n;
  const bool fold = splits >= 8 && splits % 8 == 0;
  dim3 grid(fold ? (unsigned)(tiles_total * splits) : (unsigned)tiles_total, fold ? 1u : (unsigned)splits);
  constexpr int A_ELEMS = TA ? kBK * (BM + 4) : BM * (kBK + 4), B_ELEMS = kBK * (BN + 4);
  constexpr size_t shmem = 2 * (size_t)(A_ELEMS + B_ELEMS) * sizeof(float);
  hipLaunchKernelGGL((ablate_kernel<BM, BN, WM, WN, OCC, TA, TB, ABL>), grid, dim3(WM * WN * 64), shmem, st, M, N, K, A,
                     lda, B, ldb, C, ldc, tiles_n, tiles_total, K / splits, splits > 1 ? partial : nullptr,
                     fold ? splits : 1);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

// ---- candidates (results are CORRECT, checked): two tiles of loads in flight (named register sets, loop unrolled by
// two) with the LDS stores of tile kt+1 either fenced behind the MFMAs of tile kt (PIPE 0: what
// tools/gemm_lab/generic_prefetch2.patch measured) or interleaved INTO them, one ds_write per four MFMAs
// (PIPE 1: the stored data was loaded a whole tile ago, so its s_waitcnt vmcnt does not stall the MFMA stream).
template <int BM, int BN, int WAVES_M, int WAVES_N, int OCC, bool TA, bool TB, int PIPE>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, (OCC * WAVES_M * WAVES_N + 3) / 4) void pipe_kernel(
    int64_t M, int N, int K, const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
    float* __restrict__ C, int64_t ldc, int tiles_n, int64_t tiles_total, int k_chunk, float* __restrict__ partial,
    int splits_in_x) {
  constexpr int NTHR = WAVES_M * WAVES_N * 64;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MT = WTM / 16, NT = WTN / 16;
  constexpr int LDA_S = TA ? BM + 4 : kBK + 4, LDB_S = BN + 4;
  constexpr int A_ELEMS = TA ? kBK * LDA_S : BM * LDA_S, B_ELEMS = kBK * LDB_S;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * A_ELEMS;
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
  const int nkt = k_chunk / kBK;
  const int tid = threadIdx.x;
  const int lane = tid % 64, wave = tid / 64;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int li = lane & 15, g = lane >> 4;
  f32x4_t acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // per-thread float4s of the A tile [BM x 16] and the B tile [16 x BN]; threads past the last one duplicate it
  constexpr int A_IN = TA ? BM : kBK, B_IN = TB ? kBK : BN;            // contiguous extent of a tile row in memory
  constexpr int A_VECS = BM * kBK / 4, B_VECS = kBK * BN / 4;
  constexpr int PA = (A_VECS + NTHR - 1) / NTHR, PB = (B_VECS + NTHR - 1) / NTHR;
  static_assert(PA <= 2 && PB <= 2, "at most two float4s of each operand per thread");
  auto vidx = [&](int j, int nvec) { const int v0 = tid + j * NTHR; return v0 < nvec ? v0 : nvec - 1; };
  const int va0 = vidx(0, A_VECS), va1 = vidx(PA - 1, A_VECS), vb0 = vidx(0, B_VECS), vb1 = vidx(PB - 1, B_VECS);
  auto goff = [&](int v, int inner, int64_t ld) { return (uint32_t)((v / (inner / 4)) * ld + (v % (inner / 4)) * 4); };
  const uint32_t oa0 = goff(va0, A_IN, lda), oa1 = goff(va1, A_IN, lda), ob0 = goff(vb0, B_IN, ldb), ob1 = goff(vb1, B_IN, ldb);
  // LDS: A image in memory order [outer][inner+4]; B image [k][BN+4] (transposed on the store when B is [N, K])
  auto lds_a = [&](int v) { return (v / (A_IN / 4)) * (A_IN + 4) + (v % (A_IN / 4)) * 4; };
  const int la0 = lds_a(va0), la1 = lds_a(va1);
  const int bo0 = vb0 / (B_IN / 4), bi0 = (vb0 % (B_IN / 4)) * 4, bo1 = vb1 / (B_IN / 4), bi1 = (vb1 % (B_IN / 4)) * 4;
  const float* a_base = TA ? A + (int64_t)k_begin * lda + m0 : A + m0 * lda + k_begin;
  const float* b_base = TB ? B + (int64_t)n0 * ldb + k_begin : B + (int64_t)k_begin * ldb + n0;
  const int64_t a_step = TA ? (int64_t)kBK * lda : kBK, b_step = TB ? kBK : (int64_t)kBK * ldb;
  float4 p0a0, p0a1, p0b0, p0b1, p1a0, p1a1, p1b0, p1b1;
#define PIPE_LOAD(S, T)                                                   \
  {                                                                       \
    const float* ap = a_base + (T) * a_step;                              \
    const float* bp = b_base + (T) * b_step;                              \
    S##a0 = *reinterpret_cast<const float4*>(ap + oa0);                   \
    if (PA > 1) S##a1 = *reinterpret_cast<const float4*>(ap + oa1);       \
    S##b0 = *reinterpret_cast<const float4*>(bp + ob0);                   \
    if (PB > 1) S##b1 = *reinterpret_cast<const float4*>(bp + ob1);       \
  }
#define PIPE_STORE_B(DST, O, I4, X)                                       \
  if (!TB) {                                                              \
    *reinterpret_cast<float4*>((DST) + (O) * LDB_S + (I4)) = X;           \
  } else {                                                                \
    (DST)[((I4) + 0) * LDB_S + (O)] = X.x;                                \
    (DST)[((I4) + 1) * LDB_S + (O)] = X.y;                                \
    (DST)[((I4) + 2) * LDB_S + (O)] = X.z;                                \
    (DST)[((I4) + 3) * LDB_S + (O)] = X.w;                                \
  }
#define PIPE_STORE(S, BUF)                                                \
  {                                                                       \
    float* ad = As + (BUF) * A_ELEMS;                                     \
    float* bd = Bs + (BUF) * B_ELEMS;                                     \
    *reinterpret_cast<float4*>(ad + la0) = S##a0;                         \
    if (PA > 1) *reinterpret_cast<float4*>(ad + la1) = S##a1;             \
    PIPE_STORE_B(bd, bo0, bi0, S##b0)                                     \
    if (PB > 1) PIPE_STORE_B(bd, bo1, bi1, S##b1)                         \
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
  // one tile: fragments of LDS[CUR], its MFMAs, and the LDS stores of register set S (the NEXT tile) into LDS[CUR ^ 1]
#define PIPE_TILE(CUR, S)                                                                  \
  frags(CUR);                                                                              \
  if (PIPE == 0) {                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    mfmas();                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    PIPE_STORE(S, (CUR) ^ 1)                                                               \
  } else {                                                                                 \
    PIPE_STORE(S, (CUR) ^ 1)                                                               \
    mfmas();                                                                               \
    _Pragma("unroll") for (int i_ = 0; i_ < kDsWrites; ++i_) {                             \
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                   \
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                   \
    }                                                                                      \
  }
  PIPE_LOAD(p0, 0)
  PIPE_STORE(p0, 0)
  if (nkt > 1) PIPE_LOAD(p1, 1)
  __syncthreads();
  int t = 0;
  for (; t + 3 < nkt; t += 2) {
    PIPE_LOAD(p0, t + 2)
    __builtin_amdgcn_sched_barrier(0);
    PIPE_TILE(0, p1)
    __syncthreads();
    PIPE_LOAD(p1, t + 3)
    __builtin_amdgcn_sched_barrier(0);
    PIPE_TILE(1, p0)
    __syncthreads();
  }
  // drain (nkt - t in 1..3 tiles left; LDS[0] = tile t, set 1 = tile t+1 if it exists)
  if (t + 1 < nkt) {
    const bool more = t + 2 < nkt;
    if (more) PIPE_LOAD(p0, t + 2)
    frags(0);
    mfmas();
    PIPE_STORE(p1, 1)
    __syncthreads();
    ++t;
    if (more) {
      frags(1);
      mfmas();
      PIPE_STORE(p0, 0)
      __syncthreads();
      ++t;
    }
  }
  frags(t & 1);
  mfmas();
#undef PIPE_LOAD
#undef PIPE_STORE
#undef PIPE_STORE_B
#undef PIPE_TILE
  float* out = partial ? partial + (int64_t)kz * M * ldc : C;
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t i = m0 + wm * WTM + a * 16 + g * 4 + r;
#pragma unroll
      for (int b = 0; b < NT; ++b) {
        const int j = n0 + wn * WTN + b * 16 + li;
        if (i < M && j < N) out[i * ldc + j] = acc[a][b][r];
      }
    }
}

template <int BM, int BN, int WM, int WN, int OCC, bool TA, bool TB, int PIPE>
static int launch_pipe(int64_t M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb, float* C,
                       int64_t ldc, int splits, float* partial, hipStream_t st) {
  if (M % BM || N % BN || K % (kBK * splits)) return 1;
  const int tiles_n = N / BN;
  const int64_t tiles_total = (M / BM) * tiles_n;
  const bool fold = splits >= 8 && splits % 8 == 0;
  dim3 grid(fold ? (unsigned)(tiles_total * splits) : (unsigned)tiles_total, fold ? 1u : (unsigned)splits);
  constexpr int A_ELEMS = TA ? kBK * (BM + 4) : BM * (kBK + 4), B_ELEMS = kBK * (BN + 4);
  constexpr size_t shmem = 2 * (size_t)(A_ELEMS + B_ELEMS) * sizeof(float);
  hipLaunchKernelGGL((pipe_kernel<BM, BN, WM, WN, OCC, TA, TB, PIPE>), grid, dim3(WM * WN * 64), shmem, st, M, N, K, A, lda,
                     B, ldb, C, ldc, tiles_n, tiles_total, K / splits, splits > 1 ? partial : nullptr, fold ? splits : 1);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

template <int BM, int BN, int WM, int WN, int OCC, bool TA, bool TB>
static int by_abl(int ablate, int64_t M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb, float* C,
                  int64_t ldc, int splits, float* partial, hipStream_t st) {
  switch (ablate) {
#define ABL_CASE(X) case X: return launch<BM, BN, WM, WN, OCC, TA, TB, X>(M, N, K, A, lda, B, ldb, C, ldc, splits, partial, st);
    ABL_CASE(0) ABL_CASE(1) ABL_CASE(2) ABL_CASE(3) ABL_CASE(4) ABL_CASE(5) ABL_CASE(6) ABL_CASE(7)
#undef ABL_CASE
    case 8: return launch_pipe<BM, BN, WM, WN, OCC, TA, TB, 0>(M, N, K, A, lda, B, ldb, C, ldc, splits, partial, st);
    case 9: return launch_pipe<BM, BN, WM, WN, OCC, TA, TB, 1>(M, N, K, A, lda, B, ldb, C, ldc, splits, partial, st);
  }
  return 1;
}

}  // namespace abl

// cfg: 0 = 256x80 (8 waves, 2 blocks/CU), 1 = 80x80 (5 waves, 4 blocks/CU), 2 = 256x128 (4x2 waves, 2 blocks/CU),
//      3 = 128x128 (2x2 waves, 4 blocks/CU).  form: 0 = A @ B, 1 = A @ B^T (B as [N, K]), 2 = A^T @ B (A as [K, M]).
// Split-K partials are written but not reduced (timing of the GEMM kernel itself).  Returns 0 ok, 1 unsupported, 2 HIP error.
extern "C" int ablate_gemm(int cfg, int form, int ablate, int64_t M, int N, int K, const float* A, int64_t lda,
                           const float* B, int64_t ldb, float* C, int64_t ldc, int splits, float* partial, void* stream) {
  using namespace abl;
  hipStream_t st = (hipStream_t)stream;
#define ABL_FORMS(BM, BN, WM, WN, OCC)                                                                                  \
  if (form == 0) return by_abl<BM, BN, WM, WN, OCC, false, false>(ablate, M, N, K, A, lda, B, ldb, C, ldc, splits, partial, st); \
  if (form == 1) return by_abl<BM, BN, WM, WN, OCC, false, true>(ablate, M, N, K, A, lda, B, ldb, C, ldc, splits, partial, st);  \
  if (form == 2) return by_abl<BM, BN, WM, WN, OCC, true, false>(ablate, M, N, K, A, lda, B, ldb, C, ldc, splits, partial, st);  \
  return 1;
  if (cfg == 0) { ABL_FORMS(256, 80, 8, 1, 2) }
  if (cfg == 1) { ABL_FORMS(80, 80, 5, 1, 4) }
  if (cfg == 2) { ABL_FORMS(256, 128, 4, 2, 2) }
  if (cfg == 3) { ABL_FORMS(128, 128, 2, 2, 4) }
#undef ABL_FORMS
  return 1;
}
