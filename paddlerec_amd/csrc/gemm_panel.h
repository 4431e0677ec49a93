// f32 GEMM on whole row panels — included by gemm_f32.hip after the epilogue definitions.
//
//   C[M,N] = epi(A[M,K] @ op(B)),   A row-major;   op(B) = B [K,N]  or  B^T with B [N,K];   N = 16 NT <= 448
//
// Why: the MLP GEMMs of the CTR towers are tall and NARROW (65536 x 400 x 400, deepfm/net.py:150-174).  On 256x80 tiles
// that is 1280 blocks for 512 resident slots: 2.5 rounds that cost three, and N = 400 = 25 x 16 has
// no tile width but 16, 80 and 400 that wastes no column.  Here a block owns a PANEL of 64 rows and ALL N columns:
// 65536 rows = 1024 blocks = exactly the 2 x 256 x 2 slots of one resident round, no tail, no tile scheduler; the big
// operand (A, 105 MB) is read from HBM exactly once and the small one (the weights, <= 700 KB) streams from L2.
//
// Block: 4 waves stacked along M, wave tile 16 x N: NT accumulator tiles of v_mfma_f32_16x16x4_f32 (100 VGPRs at
// N = 400), two blocks per CU (2 waves per SIMD, 256 VGPRs each).  Per k-step of 16: A tile [64][16], B tile [16][N],
// global -> registers -> LDS one k-step ahead, double-buffered LDS, one barrier per k-step.
// Column permutation: accumulator tile b = 4 q + c of lane (li, g) holds column j = 64 q + 4 li + c (the MFMA does not
// care which 16 columns form a tile), so a lane's four tiles of one quad are 4 CONSECUTIVE columns: the B fragment of a
// quad is one ds_read_b128 (7 LDS reads per 25 MFMAs instead of 25) and the epilogue stores float4s, 256 contiguous bytes
// per 16 lanes (the 80-wide tiles store 64).  Tiles past the last whole quad (NT % 4) keep the plain mapping
// j = 64 NQ + 16 t + li.
// MFMA step s of a k-tile multiplies k = 4 g + s of both operands, k-tiles ascending: the same order of additions per
// output element as gemm_f32_kernel / gemm_f32_pipe_kernel -> bit-identical results.
// Whole panels only (M % 64 == 0, K % 16 == 0, 16-B aligned rows of A, B, C and aux0): the launcher falls back otherwise.
#pragma once

namespace rec {

constexpr int kPanelRows = 64;

template <int NT, bool TB, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_f32_panel_kernel(int64_t M, int K, const float* __restrict__ A,
                                                                int64_t lda, const float* __restrict__ B, int64_t ldb,
                                                                float* __restrict__ C, int64_t ldc, EpiArgs epi) {
  constexpr int N = NT * 16, NQ = NT / 4, NTAIL = NT % 4;
  constexpr int LDA_S = kBK + 4, LDB_S = N + 4;
  constexpr int A_ELEMS = kPanelRows * LDA_S, B_ELEMS = kBK * LDB_S;
  constexpr int B_VECS = kBK * N / 4, PB = (B_VECS + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float panel_smem[];
  float* As = panel_smem;
  float* Bs = panel_smem + 2 * A_ELEMS;
  const int tid = threadIdx.x;
  const int lane = tid % kWave, wave = tid / kWave;
  const int li = lane & 15, g = lane >> 4;
  const int nkt = K / kBK;
  const int64_t npanels = M / kPanelRows;
  int64_t panel = blockIdx.x;
  f32x4_t acc[NT];

  // this thread's float4 of the A tile and its PB float4s of the B tile (a thread past the last one re-loads it)
  const float* a_src = A + (panel * kPanelRows + tid / 4) * lda + (tid % 4) * 4;
  const int a_dst = (tid / 4) * LDA_S + (tid % 4) * 4;
  constexpr int B_IN = TB ? kBK : N;                       // contiguous extent of a tile row in memory
  uint32_t b_off[PB];
  int b_lds[PB];
#pragma unroll
  for (int it = 0; it < PB; ++it) {
    const int v0 = tid + it * 256, v = v0 < B_VECS ? v0 : B_VECS - 1;
    const int o = v / (B_IN / 4), i4 = (v % (B_IN / 4)) * 4;
    b_off[it] = (uint32_t)(o * ldb + i4);
    b_lds[it] = TB ? i4 * LDB_S + o : o * LDB_S + i4;
  }
  const bool last_ok = tid + (PB - 1) * 256 < B_VECS;      // the last float4 of this thread is inside the tile
  const int64_t b_step = TB ? kBK : (int64_t)kBK * ldb;
  static_assert(PB == 7, "the staging registers below are seven named float4s");
  float4 sa, sb0, sb1, sb2, sb3, sb4, sb5, sb6;
  // (named registers and macros: as an array captured by lambdas — or carried around the panel loop — the compiler put the
  // staging registers into scratch and waited for every global load before issuing the next: 67 TF)
#define REC_PANEL_EACH(X) X(0, sb0) X(1, sb1) X(2, sb2) X(3, sb3) X(4, sb4) X(5, sb5) X(6, sb6)
#define REC_PANEL_LD1(IT, R) R = *reinterpret_cast<const float4*>(bp_ + b_off[IT]);
#define REC_PANEL_LOAD(KT)                                                                        \
  {                                                                                               \
    sa = *reinterpret_cast<const float4*>(a_src + (KT) * kBK);                                    \
    const float* bp_ = B + (KT) * b_step;                                                         \
    REC_PANEL_EACH(REC_PANEL_LD1)                                                                 \
  }
#define REC_PANEL_ST1(IT, R)                                                                      \
  if (B_VECS % 256 == 0 || IT + 1 < PB || last_ok) {                                              \
    if (!TB) {                                                                                    \
      *reinterpret_cast<float4*>(bd_ + b_lds[IT]) = R;                                            \
    } else { /* memory [n][k] -> LDS [k][n] */                                                    \
      bd_[b_lds[IT] + 0 * LDB_S] = R.x;                                                           \
      bd_[b_lds[IT] + 1 * LDB_S] = R.y;                                                           \
      bd_[b_lds[IT] + 2 * LDB_S] = R.z;                                                           \
      bd_[b_lds[IT] + 3 * LDB_S] = R.w;                                                           \
    }                                                                                             \
  }
#define REC_PANEL_STORE(BUF)                                                                      \
  {                                                                                               \
    *reinterpret_cast<float4*>(As + (BUF) * A_ELEMS + a_dst) = sa;                                \
    float* bd_ = Bs + (BUF) * B_ELEMS;                                                            \
    REC_PANEL_EACH(REC_PANEL_ST1)                                                                 \
  }
  const int a_frag = (wave * 16 + li) * LDA_S + g * 4;
#define REC_PANEL_COMPUTE(BUF)                                                                    \
  {                                                                                               \
    const float4 a4 = *reinterpret_cast<const float4*>(As + (BUF) * A_ELEMS + a_frag);            \
    const float av[4] = {a4.x, a4.y, a4.z, a4.w};                                                 \
    const float* bs_ = Bs + (BUF) * B_ELEMS + g * 4 * LDB_S;                                      \
    _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) {                                            \
      const float* br = bs_ + s_ * LDB_S;                                                         \
      float bf[NT];                                                                               \
      _Pragma("unroll") for (int q = 0; q < NQ; ++q) {                                            \
        const float4 t = *reinterpret_cast<const float4*>(br + q * 64 + li * 4);                  \
        bf[4 * q + 0] = t.x; bf[4 * q + 1] = t.y; bf[4 * q + 2] = t.z; bf[4 * q + 3] = t.w;       \
      }                                                                                           \
      _Pragma("unroll") for (int t = 0; t < NTAIL; ++t) bf[4 * NQ + t] = br[NQ * 64 + t * 16 + li]; \
      _Pragma("unroll") for (int b = 0; b < NT; ++b)                                              \
          acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s_], bf[b], acc[b], 0, 0, 0);          \
    }                                                                                             \
  }

  // Persistent over panels (grid = the resident slots when M has more panels than that): the stores of a
  // panel's epilogue drain under the next panel's MFMAs — with one
  // block per panel every block of a round reaches its (HBM-bound) epilogue at the same moment and the matrix pipes idle.
  REC_PANEL_LOAD(0)
  for (; panel < npanels; panel += gridDim.x) {
    const int64_t m0 = panel * kPanelRows;
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    REC_PANEL_STORE(0)
    __syncthreads();
    for (int kt = 0; kt + 1 < nkt; ++kt) {
      REC_PANEL_LOAD(kt + 1)
      REC_PANEL_COMPUTE(kt & 1)
      REC_PANEL_STORE((kt & 1) ^ 1)
      __syncthreads();
    }
    REC_PANEL_COMPUTE((nkt - 1) & 1)
    __syncthreads();                       // the next panel's first store reuses buffer 0
  // epilogue: row i = m0 + 16 wave + 4 g + r; quad q -> columns 64 q + 4 li .. + 3 (one float4), tail tile t -> column
  // 64 NQ + 16 t + li.  Every aux load of a row group is issued before its stores (see apply_epi).
  float bq[NQ > 0 ? NQ : 1][4], bt[NTAIL > 0 ? NTAIL : 1];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int c = 0; c < 4; ++c) bq[q][c] = load_bias<EPI>(q * 64 + li * 4 + c, epi);
#pragma unroll
  for (int t = 0; t < NTAIL; ++t) bt[t] = load_bias<EPI>(NQ * 64 + t * 16 + li, epi);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t i = m0 + wave * 16 + g * 4 + r;
    float4 x0q[NQ > 0 ? NQ : 1];
    float x0t[NTAIL > 0 ? NTAIL : 1], x1q[NQ > 0 ? NQ : 1][4], x1t[NTAIL > 0 ? NTAIL : 1];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      x0q[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (EpiUses<EPI>::aux0 && (EPI != REC_EPI_ADD || epi.aux0))
        x0q[q] = *reinterpret_cast<const float4*>(epi.aux0 + i * epi.ld0 + q * 64 + li * 4);
#pragma unroll
      for (int c = 0; c < 4; ++c) x1q[q][c] = load_aux1<EPI>(i, q * 64 + li * 4 + c, epi);
    }
#pragma unroll
    for (int t = 0; t < NTAIL; ++t) {
      x0t[t] = load_aux0<EPI>(i, NQ * 64 + t * 16 + li, epi);
      x1t[t] = load_aux1<EPI>(i, NQ * 64 + t * 16 + li, epi);
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      float4 o;
      o.x = apply_epi<EPI>(acc[4 * q + 0][r], x0q[q].x, x1q[q][0], bq[q][0], i, epi);
      o.y = apply_epi<EPI>(acc[4 * q + 1][r], x0q[q].y, x1q[q][1], bq[q][1], i, epi);
      o.z = apply_epi<EPI>(acc[4 * q + 2][r], x0q[q].z, x1q[q][2], bq[q][2], i, epi);
      o.w = apply_epi<EPI>(acc[4 * q + 3][r], x0q[q].w, x1q[q][3], bq[q][3], i, epi);
      *reinterpret_cast<float4*>(C + i * ldc + q * 64 + li * 4) = o;
    }
#pragma unroll
    for (int t = 0; t < NTAIL; ++t)
      C[i * ldc + NQ * 64 + t * 16 + li] = apply_epi<EPI>(acc[4 * NQ + t][r], x0t[t], x1t[t], bt[t], i, epi);
  }
    if (panel + gridDim.x < npanels) {     // (requested AFTER the epilogue: in front of it the staging registers spill)
      a_src += (int64_t)gridDim.x * kPanelRows * lda;
      REC_PANEL_LOAD(0)
    }
  }   // panels
#undef REC_PANEL_LOAD
#undef REC_PANEL_EACH
#undef REC_PANEL_LD1
#undef REC_PANEL_ST1
#undef REC_PANEL_STORE
#undef REC_PANEL_COMPUTE
}

template <int NT, bool TB>
static bool launch_panel_nt(const rec_gemm_desc* d, const float* A, const float* B, float* C, const EpiArgs& e,
                            hipStream_t st, int num_cus) {
  constexpr size_t shmem = 2 * (size_t)(kPanelRows * (kBK + 4) + kBK * (NT * 16 + 4)) * sizeof(float);
  static const int grid_env = [] { const char* v = getenv("REC_GEMM_PANEL_GRID"); return v && *v ? atoi(v) : 0; }();
  const int64_t panels = d->m / kPanelRows, slots = grid_env > 0 ? grid_env : 2 * (int64_t)num_cus;
  const dim3 grid((unsigned)(panels < slots ? panels : slots));
#define REC_PANEL_CASE(E)                                                                                          \
  case E: {                                                                                                        \
    auto kern = gemm_f32_panel_kernel<NT, TB, E>;                                                                  \
    if (shmem > 64 * 1024) {                                                                                       \
      static const hipError_t attr =                                                                               \
          hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);          \
      if (attr != hipSuccess) return false;                                                                        \
    }                                                                                                              \
    hipLaunchKernelGGL(kern, grid, dim3(256), shmem, st, d->m, d->k, A, (int64_t)d->lda, B, (int64_t)d->ldb, C,    \
                       (int64_t)d->ldc, e);                                                                        \
    return true;                                                                                                   \
  }
  switch (d->epilogue) {
    REC_PANEL_CASE(REC_EPI_NONE)
    REC_PANEL_CASE(REC_EPI_BIAS)
    REC_PANEL_CASE(REC_EPI_BIAS_RELU)
    REC_PANEL_CASE(REC_EPI_RELU_MASK)
  }
#undef REC_PANEL_CASE
  return false;
}

// -> false: not eligible (the caller takes the tiled kernels).  REC_GEMM_PANEL=0 switches the panel kernel off.
static bool launch_panel(const rec_gemm_desc* d, const float* A, const float* B, float* C, const EpiArgs& e,
                         hipStream_t st, int num_cus) {
  // opt-in (REC_GEMM_PANEL=1 every eligible GEMM, =nn only B [K,N]); read per call: tests flip it inside one process
  const char* env = getenv("REC_GEMM_PANEL");
  if (!env || !(*env == '1' || *env == 'n')) return false;
  if (*env == 'n' && d->trans_b) return false;
  if (d->trans_a || d->split_k > 1 || d->m % kPanelRows || d->k % kBK || d->k < 2 * kBK) return false;
  if (d->n != 400 && d->n != 432) return false;         // instantiated widths: the CTR towers' 400, 432 = 27 x 16
  if (d->lda % 4 || d->ldb % 4 || d->ldc % 4 || ((uintptr_t)A) % 16 || ((uintptr_t)B) % 16 || ((uintptr_t)C) % 16)
    return false;
  if (d->lda >= (1 << 23) || d->ldb >= (1 << 23)) return false;
  const int ep = d->epilogue;
  if (!(ep == REC_EPI_NONE || ep == REC_EPI_BIAS || ep == REC_EPI_BIAS_RELU || ep == REC_EPI_RELU_MASK)) return false;
  if (ep == REC_EPI_RELU_MASK && (e.ld0 % 4 || ((uintptr_t)e.aux0) % 16)) return false;
  // worth it when the panels fill whole resident rounds (2 blocks per CU): >= 85 % of the slots of the last round
  const int64_t panels = d->m / kPanelRows, slots = 2 * (int64_t)num_cus;
  const int64_t rounds = (panels + slots - 1) / slots;
  if (panels < slots || panels * 100 < rounds * slots * 85) return false;
  if (d->n == 400)
    return d->trans_b ? launch_panel_nt<25, true>(d, A, B, C, e, st, num_cus) : launch_panel_nt<25, false>(d, A, B, C, e, st, num_cus);
  return d->trans_b ? launch_panel_nt<27, true>(d, A, B, C, e, st, num_cus) : launch_panel_nt<27, false>(d, A, B, C, e, st, num_cus);
}

}  // namespace rec
