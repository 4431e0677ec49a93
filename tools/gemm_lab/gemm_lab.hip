// GEMM lab (measurement tool, not part of librecengine.so): the f32-MFMA GEMM of csrc/gemm_f32.hip with every
// tiling decision a template parameter, so that ONE GPU call can rank block tile / K step / MFMA shape / occupancy
// on the shapes of the hot path.  The winners are folded back into csrc/gemm_f32.hip.
//   C[M,N] = op(A) @ op(B), row-major; TA: A given as [K,M]; TB: B given as [N,K]; split-K partials summed in order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <type_traits>

namespace lab {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

// Tile T[R][C] of a matrix; MEMT: memory is contiguous along R (element (r,c) at p[c*ld + r]).
// LDS image = the memory order: [R][C+4] (!MEMT) or [C][R+4] (MEMT).  Full float4s, no edge checks except rows/cols
// beyond rmax/cmax load zeros.
template <int R, int C, bool MEMT, int NTHR>
struct Loader {
  static constexpr int INNER = MEMT ? R : C, OUTER = MEMT ? C : R;
  static constexpr int LD = INNER + 4;
  static constexpr int kVecs = R * C / 4, kPer = (kVecs + NTHR - 1) / NTHR;
  float4 st[kPer];
  __device__ __forceinline__ void load(const float* __restrict__ p, int64_t ld, int64_t r0, int64_t c0, int64_t rmax,
                                       int64_t cmax, int tid) {
#pragma unroll
    for (int it = 0; it < kPer; ++it) {
      const int v = tid + it * NTHR;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kVecs % NTHR == 0 || v < kVecs) {
        const int o = v / (INNER / 4), i4 = (v % (INNER / 4)) * 4;
        const int64_t go = (MEMT ? c0 : r0) + o, gi = (MEMT ? r0 : c0) + i4;
        const int64_t omax = MEMT ? cmax : rmax, imax = MEMT ? rmax : cmax;
        if (go < omax && gi + 3 < imax) x = *reinterpret_cast<const float4*>(p + go * ld + gi);
        else if (go < omax) {
          const float* q = p + go * ld + gi;
          if (gi + 0 < imax) x.x = q[0];
          if (gi + 1 < imax) x.y = q[1];
          if (gi + 2 < imax) x.z = q[2];
        }
      }
      st[it] = x;
    }
  }
  // MODE 0: the memory order, rows padded by 4 floats;  MODE 1: rows of exactly 16 floats, the four float4 chunks of
  // row o XOR-swizzled with h[(o >> 2) & 3], h = {0,2,3,1} (conflict-free ds_read_b128 fragment reads for the
  // 16x16x4 lane map: see the bank table of MI355X_MICROARCH.md);  MODE 2: transposed ([INNER][OUTER+4]).
  template <int MODE>
  __device__ __forceinline__ void store(float* __restrict__ lds, int tid) const {
#pragma unroll
    for (int it = 0; it < kPer; ++it) {
      const int v = tid + it * NTHR;
      if (kVecs % NTHR == 0 || v < kVecs) {
        const int o = v / (INNER / 4), i4 = (v % (INNER / 4)) * 4;
        if (MODE == 0) {
          *reinterpret_cast<float4*>(lds + o * LD + i4) = st[it];
        } else if (MODE == 1) {
          *reinterpret_cast<float4*>(lds + o * 16 + (((i4 >> 2) ^ swz(o)) << 2)) = st[it];
        } else {
          lds[(i4 + 0) * (OUTER + 4) + o] = st[it].x;
          lds[(i4 + 1) * (OUTER + 4) + o] = st[it].y;
          lds[(i4 + 2) * (OUTER + 4) + o] = st[it].z;
          lds[(i4 + 3) * (OUTER + 4) + o] = st[it].w;
        }
      }
    }
  }
  static __device__ __forceinline__ int swz(int row) {   // h[(row >> 2) & 3], h = {0, 2, 3, 1}
    return (0x78 >> (((row >> 2) & 3) * 2)) & 3;         // 0b01'11'10'00
  }
};

// AL / BL: LDS image of a k-contiguous operand (A when !TA, B when TB): 0 padded rows + b128 reads, 1 swizzled rows +
// b128 reads (BK == 16, MF == 16 only), 2 transposed on the store + b32 reads.  Ignored for m/n-contiguous operands.
template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, int MF, bool TA, bool TB, int OCC, int AL, int BL>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, (OCC * WAVES_M * WAVES_N + 3) / 4) void gemm_lab_kernel(
    int64_t M, int N, int K, const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
    float* __restrict__ C, int64_t ldc, int tiles_n, int64_t tiles_total, int k_chunk, float* __restrict__ partial) {
  constexpr int NTHR = WAVES_M * WAVES_N * 64;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MT = WTM / MF, NT = WTN / MF;
  constexpr int G = 64 / MF;           // k-groups per MFMA step set: lane (i = l % MF, g = l / MF)
  constexpr int KC = G * 4;            // k consumed by one fragment read (4 MFMA steps)
  constexpr int ACC = MF * MF / 64;    // accumulator registers per MFMA tile (4 or 16)
  static_assert(WTM % MF == 0 && WTN % MF == 0 && BK % KC == 0, "tile shape");
  using LA = Loader<BM, BK, TA, NTHR>;   // logical [BM rows][BK k]; memory order k-contiguous unless TA
  using LB = Loader<BK, BN, TB, NTHR>;   // logical [BK k][BN cols]; memory order n-contiguous unless TB
  // LDS images: A !TA: [BM][BK+4] (k contiguous, "KC" reads);  A TA: [BK][BM+4] ("MC" reads)
  //             B !TB: [BK][BN+4] ("MC");                      B TB: [BN][BK+4] ("KC")
  constexpr bool A_MC = TA || AL == 2, B_MC = !TB || BL == 2;      // m/n-contiguous LDS image -> b32 fragment reads
  constexpr int A_MODE = TA ? 0 : AL, B_MODE = TB ? BL : 0;
  static_assert((A_MODE != 1 && B_MODE != 1) || (BK == 16 && MF == 16), "swizzled image needs BK = 16 on 16x16x4");
  constexpr int A_EL = A_MC ? BK * (BM + 4) : (A_MODE == 1 ? BM * 16 : BM * (BK + 4));
  constexpr int B_EL = B_MC ? BK * (BN + 4) : (B_MODE == 1 ? BN * 16 : BN * (BK + 4));
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                 // [2][A_EL]
  float* Bs = smem + 2 * A_EL;      // [2][B_EL]

  int64_t w = blockIdx.x;
  const int kz = blockIdx.y;
  {
    const int64_t per = tiles_total / 8;   // XCD-aware: contiguous tile range per XCD
    if (w < per * 8) w = (w % 8) * per + w / 8;
  }
  const int64_t tm = w / tiles_n;
  const int tn = (int)(w % tiles_n);
  const int64_t m0 = tm * BM;
  const int n0 = tn * BN;
  const int k_begin = kz * k_chunk;
  const int k_end = (k_begin + k_chunk < K) ? k_begin + k_chunk : K;
  const int tid = threadIdx.x, lane = tid % 64, wave = tid / 64;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int li = lane % MF, g = lane / MF;

  LA la;
  LB lb;
  using AccT = std::conditional_t<MF == 16, f32x4_t, f32x16_t>;
  AccT acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < ACC; ++r) acc[a][b][r] = 0.f;

  const int nkt = (k_end - k_begin + BK - 1) / BK;
  if (nkt > 0) {
    la.load(A, lda, m0, k_begin, M, k_end, tid);
    lb.load(B, ldb, k_begin, n0, k_end, N, tid);
    la.template store<A_MODE>(As, tid);
    lb.template store<B_MODE>(Bs, tid);
  }
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nkt) {
      const int k0 = k_begin + (kt + 1) * BK;
      la.load(A, lda, m0, k0, M, k_end, tid);
      lb.load(B, ldb, k0, n0, k_end, N, tid);
    }
    const float* as = As + cur * A_EL;
    const float* bs = Bs + cur * B_EL;
#pragma unroll
    for (int kc = 0; kc < BK / KC; ++kc) {
      float af[MT][4], bf[NT][4];
#pragma unroll
      for (int a = 0; a < MT; ++a) {
        const int row = wm * WTM + a * MF + li;
        if (!A_MC) {
          const float* q = A_MODE == 1 ? as + row * 16 + ((g ^ LA::swz(row)) << 2) : as + row * (BK + 4) + kc * KC + g * 4;
          const float4 t = *reinterpret_cast<const float4*>(q);
          af[a][0] = t.x; af[a][1] = t.y; af[a][2] = t.z; af[a][3] = t.w;
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) af[a][s] = as[(kc * KC + g * 4 + s) * (BM + 4) + row];
        }
      }
#pragma unroll
      for (int b = 0; b < NT; ++b) {
        const int col = wn * WTN + b * MF + li;
        if (!B_MC) {
          const float* q = B_MODE == 1 ? bs + col * 16 + ((g ^ LB::swz(col)) << 2) : bs + col * (BK + 4) + kc * KC + g * 4;
          const float4 t = *reinterpret_cast<const float4*>(q);
          bf[b][0] = t.x; bf[b][1] = t.y; bf[b][2] = t.z; bf[b][3] = t.w;
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) bf[b][s] = bs[(kc * KC + g * 4 + s) * (BN + 4) + col];
        }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
          for (int b = 0; b < NT; ++b) {
            if constexpr (MF == 16)
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a][s], bf[b][s], acc[a][b], 0, 0, 0);
            else
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][s], bf[b][s], acc[a][b], 0, 0, 0);
          }
    }
    if (kt + 1 < nkt) {
      la.template store<A_MODE>(As + (cur ^ 1) * A_EL, tid);
      lb.template store<B_MODE>(Bs + (cur ^ 1) * B_EL, tid);
    }
    __syncthreads();
  }
  // C/D layouts: 16x16x4: col = lane & 15, row = (lane >> 4) * 4 + r;  32x32x2: col = lane & 31,
  // row = 8 * (r / 4) + 4 * (lane >> 5) + r % 4
  float* out = partial ? partial + (int64_t)kz * M * ldc : C;
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int r = 0; r < ACC; ++r) {
      const int rr = MF == 16 ? g * 4 + r : 8 * (r / 4) + 4 * g + r % 4;
      const int64_t i = m0 + wm * WTM + a * MF + rr;
      if (i < M) {
#pragma unroll
        for (int b = 0; b < NT; ++b) {
          const int j = n0 + wn * WTN + b * MF + li;
          if (j < N) out[i * ldc + j] = acc[a][b][r];
        }
      }
    }
}

__global__ void reduce_kernel(int64_t M, int N, int64_t ldc, int splits, const float* __restrict__ partial,
                              float* __restrict__ C) {
  const int64_t total = M * N;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t i = e / N;
    const int j = (int)(e % N);
    float t = 0.f;
    for (int z = 0; z < splits; ++z) t += partial[(int64_t)z * M * ldc + i * ldc + j];
    C[i * ldc + j] = t;
  }
}

struct Cfg {
  const char* name;
  int bm, bn, bk, threads, occ;
  size_t lds[2][2];   // [ta][tb]
  void (*launch[2][2])(dim3, size_t, hipStream_t, int64_t, int, int, const float*, int64_t, const float*, int64_t,
                       float*, int64_t, int, int64_t, int, float*);
};

template <int BM, int BN, int BK, int WM, int WN, int MF, bool TA, bool TB, int OCC, int AL, int BL>
static void launch(dim3 grid, size_t shmem, hipStream_t st, int64_t M, int N, int K, const float* A, int64_t lda,
                   const float* B, int64_t ldb, float* C, int64_t ldc, int tiles_n, int64_t tiles_total, int k_chunk,
                   float* partial) {
  auto kern = gemm_lab_kernel<BM, BN, BK, WM, WN, MF, TA, TB, OCC, AL, BL>;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), shmem, st, M, N, K, A, lda, B, ldb, C, ldc, tiles_n,
                     tiles_total, k_chunk, partial);
}

template <int BM, int BN, int BK, bool TA, bool TB, int AL, int BL>
constexpr size_t lds_bytes() {
  const size_t a = (TA || AL == 2) ? BK * (BM + 4) : (AL == 1 ? BM * 16 : BM * (BK + 4));
  const size_t b = (!TB || BL == 2) ? BK * (BN + 4) : (BL == 1 ? BN * 16 : BN * (BK + 4));
  return 2 * (a + b) * 4;
}

#define LAB_CFG(BM, BN, BK, WM, WN, MF, OCC, AL, BL)                                                           \
  {#BM "x" #BN "x" #BK " w" #WM "x" #WN " mf" #MF " occ" #OCC " a" #AL "b" #BL, BM, BN, BK, WM * WN * 64, OCC,  \
   {{lds_bytes<BM, BN, BK, false, false, AL, BL>(), lds_bytes<BM, BN, BK, false, true, AL, BL>()},             \
    {lds_bytes<BM, BN, BK, true, false, AL, BL>(), lds_bytes<BM, BN, BK, true, true, AL, BL>()}},              \
   {{launch<BM, BN, BK, WM, WN, MF, false, false, OCC, AL, BL>,                                                \
     launch<BM, BN, BK, WM, WN, MF, false, true, OCC, AL, BL>},                                                \
    {launch<BM, BN, BK, WM, WN, MF, true, false, OCC, AL, BL>, nullptr}}}

static const Cfg kCfgs[] = {
    LAB_CFG(128, 80, 16, 4, 1, 16, 5, 0, 0),    // 0: the engine's N=400 config, padded k-contiguous images
    LAB_CFG(128, 80, 16, 4, 1, 16, 5, 1, 1),    // 1: swizzled
    LAB_CFG(128, 80, 16, 4, 1, 16, 5, 2, 2),    // 2: transposed on the store
    LAB_CFG(128, 80, 16, 4, 1, 16, 5, 1, 2),    // 3: A swizzled, B transposed
    LAB_CFG(128, 128, 16, 2, 2, 16, 4, 0, 0),   // 4
    LAB_CFG(128, 128, 16, 2, 2, 16, 4, 1, 1),   // 5
    LAB_CFG(128, 128, 16, 2, 2, 16, 4, 2, 2),   // 6
    LAB_CFG(256, 128, 16, 4, 2, 16, 2, 0, 0),   // 7: 8 waves
    LAB_CFG(256, 128, 16, 4, 2, 16, 2, 1, 1),   // 8
    LAB_CFG(256, 128, 16, 4, 2, 16, 2, 2, 2),   // 9
    LAB_CFG(256, 128, 16, 4, 2, 16, 2, 1, 2),   // 10
    LAB_CFG(80, 80, 16, 5, 1, 16, 4, 0, 0),     // 11
    LAB_CFG(80, 80, 16, 5, 1, 16, 4, 1, 1),     // 12
    LAB_CFG(64, 80, 16, 4, 1, 16, 6, 0, 0),     // 13
    LAB_CFG(64, 80, 16, 4, 1, 16, 6, 1, 2),     // 14
    LAB_CFG(256, 80, 16, 8, 1, 16, 2, 1, 2),    // 15: 8 waves, narrow
    LAB_CFG(128, 160, 16, 4, 2, 16, 2, 1, 2),   // 16: 8 waves, wave tile 32x80
    LAB_CFG(256, 160, 16, 4, 2, 16, 1, 1, 2),   // 17: 8 waves, wave tile 64x80
};
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);

}  // namespace lab

extern "C" int lab_num_configs() { return lab::kNumCfgs; }
extern "C" const char* lab_config_name(int i) { return i >= 0 && i < lab::kNumCfgs ? lab::kCfgs[i].name : ""; }
extern "C" int lab_workspace_floats(int64_t M, int64_t ldc, int splits, int64_t* out) {
  *out = splits > 1 ? (int64_t)splits * M * ldc : 0;
  return 0;
}

// returns 0 ok, 1 unsupported combination, 2 HIP error
extern "C" int lab_gemm(int cfg, int ta, int tb, int64_t M, int N, int K, const float* A, int64_t lda, const float* B,
                        int64_t ldb, float* C, int64_t ldc, int splits, float* workspace, void* stream) {
  using namespace lab;
  if (cfg < 0 || cfg >= kNumCfgs || (ta && tb)) return 1;
  const Cfg& c = kCfgs[cfg];
  auto fn = c.launch[ta ? 1 : 0][tb ? 1 : 0];
  if (!fn) return 1;
  if (lda % 4 || ldb % 4 || ((uintptr_t)A) % 16 || ((uintptr_t)B) % 16) return 1;
  const size_t shmem = c.lds[ta ? 1 : 0][tb ? 1 : 0];
  if (shmem > 160 * 1024) return 1;
  const int tiles_n = (N + c.bn - 1) / c.bn;
  const int64_t tiles_m = (M + c.bm - 1) / c.bm;
  const int64_t tiles_total = tiles_m * tiles_n;
  if (splits < 1) splits = 1;
  const int nkt = (K + c.bk - 1) / c.bk;
  if (splits > nkt) splits = nkt;
  const int kt_per = (nkt + splits - 1) / splits;
  const int k_chunk = kt_per * c.bk;
  splits = (nkt + kt_per - 1) / kt_per;
  hipStream_t st = (hipStream_t)stream;
  float* partial = splits > 1 ? workspace : nullptr;
  fn(dim3((unsigned)tiles_total, (unsigned)splits), shmem, st, M, N, K, A, lda, B, ldb, C, ldc, tiles_n, tiles_total,
     k_chunk, partial);
  if (splits > 1) {
    int64_t grid = (M * N + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(reduce_kernel, dim3((unsigned)grid), dim3(256), 0, st, M, N, ldc, splits, partial, C);
  }
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
