// f32 GEMM with LDS-DMA staging (gfx950 `global_load_lds_dwordx4`) and a three-stage LDS ring — included by
// gemm_f32.hip after the epilogue definitions.
//
//   C[M,N] = epi(A[M,K] @ op(B)),   A row-major (k contiguous);   op(B) = B [K,N]  or  B^T with B [N,K]
//
// Why: gemm_f32_kernel / gemm_f32_pipe_kernel stage every tile global -> registers -> LDS, so a tile's loads have ONE
// k-step to arrive and the loader's registers cap the prefetch depth; PMC put the matrix pipe of those kernels at
// 60-75 % busy (profiles/r02d_gemm_pmc.txt, r02f_gemm_ablate.txt).  Here the tiles go global -> LDS directly (no staging
// VGPRs, no ds_write pass), TWO k-steps ahead, and cross the barrier in flight: the only waits in the loop are a
// COUNTED `s_waitcnt vmcnt(n)` (this wave's loads of the NEXT tile have landed; the tile after it stays in flight)
// and one raw `s_barrier` per k-step (cdna_hip_programming.md "Pipelining across barriers").
//
// Block: 8 waves stacked along M, wave tile (BM/8) x BN = MT x NT MFMA tiles of 16x16 (v_mfma_f32_16x16x4_f32).
//   256 x 80  (MT 2, NT 5)  : N = 400, the DeepFM MLP width (5 column tiles, no waste)
//   128 x 208 (MT 1, NT 13) : N = 416 (layer-0 dX)
// LDS stage = A tile [BM][16] + B tile, 21 KB; three stages = 63 KB -> two blocks per CU.
// LDS images (an LDS-DMA instruction writes 1 KB lane-linear: lane i -> bytes [16 i, 16 i + 16) of the piece, so the
// layout is chosen through each lane's SOURCE address, cdna_hip_programming.md rule 21):
//   k-contiguous operand (A; B given [N,K]): pieces of 16 rows x 16 k; slot (r, c) of a piece holds the row's k-chunk
//     c ^ f(r), f(r) = {0,3,2,1}[r / 4].  The fragment read of lane (i = lane & 15, g = lane >> 4) is ONE
//     ds_read_b128 of slot (i, g ^ f(i)) = k 4g..4g+3 — conflict-free in all four 16-lane groups of a b128 read — and
//     MFMA step s multiplies the k = 4g + s slices of both operands (every k once, fixed order).
//   B given [K,N]: the tile's 16 k-rows of BN floats, row r rotated left by 16 floats when (r >> 2) is odd, so that the
//     two k-rows a 32-lane ds_read_b32 phase touches (4g + s, g = 0 / 1) fall into opposite bank halves.
// A wave loads exactly the A rows it multiplies (no cross-wave dependency on A); the B pieces are dealt round-robin.
// Whole tiles only (M % BM == 0, N % BN == 0, K % 16 == 0, 16-B aligned rows): the launcher falls back otherwise.
#pragma once

namespace rec {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

__device__ __forceinline__ void glds16(const float* src, const char* lds_dst) {
  __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)lds_dst, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int kGldsStages = 3;

template <int BM, int BN, bool TB, int EPI>
__global__ __launch_bounds__(512, 4) void gemm_f32_glds_kernel(
    int64_t M, int N, int K, const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
    float* __restrict__ C, int64_t ldc, EpiArgs epi, int tiles_n, int64_t tiles_total, int skew) {
  constexpr int NW = 8, WTM = BM / NW, MT = WTM / 16, NT = BN / 16;
  static_assert(WTM % 16 == 0 && BN % 16 == 0, "wave tile must be a multiple of the 16x16 MFMA tile");
  constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, STAGE = A_BYTES + B_BYTES;
  constexpr int NB_LO = NT / NW, NB_REM = NT % NW;          // B pieces per wave: NB_LO (+1 for waves < NB_REM)
  constexpr int NB_MAX = NB_LO + (NB_REM ? 1 : 0);
  extern __shared__ __attribute__((aligned(1024))) char glds_smem[];

  const int lane = threadIdx.x % kWave;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int li = lane & 15, g = lane >> 4;
  const int nkt = K / kBK;

  // PERSISTENT block: output tiles blockIdx.x, + gridDim.x, ... (the launcher sizes the grid to the resident blocks).
  // The ring does not drain between tiles: the loads of the next tile's first k-steps are issued during the last
  // k-steps of this one, and a tile's C stores drain underneath the next tile's MFMAs — the per-tile fill / drain
  // that cost a non-persistent launch ~15 us per round of blocks (profiles/r03_gemm_shape_probe.txt) is paid once.
  // XCD-aware numbering (block b and tile t = b + j * grid run on XCD b % 8, grid % 8 == 0): every XCD gets a
  // contiguous range of tiles, so the N-tiles that share an A tile run on one XCD at the same time.
  const int64_t per = tiles_total / 8;
  auto tile_origin = [&](int64_t t, int64_t& m0, int& n0) {
    const int64_t w = t < per * 8 ? (t % 8) * per + t / 8 : t;
    m0 = (w / tiles_n) * BM;
    n0 = (int)(w % tiles_n) * BN;
  };
  const int64_t first = blockIdx.x, stride = gridDim.x;
  if (first >= tiles_total) return;
  const int64_t my_tiles = (tiles_total - first + stride - 1) / stride;
  const int64_t total_steps = my_tiles * nkt;

  // ---- issue side: per-lane source pointers of the tile being loaded (k-step 0); a k-step adds 16 floats
  // (k-contiguous operand) or 16 rows ([K,N] operand)
  const int pr = lane >> 2;                                   // row of the lane's slot inside a 16 x 16 piece
  const int pc = (lane & 3) ^ ((4 - (pr >> 2)) & 3);          // k-chunk stored there: c ^ f(r), f = {0,3,2,1}[r/4]
  const float* a_src[MT];
  const float* b_src[NB_MAX > 0 ? NB_MAX : 1];
  auto set_sources = [&](int64_t t) {
    int64_t m0;
    int n0;
    tile_origin(t, m0, n0);
#pragma unroll
    for (int a = 0; a < MT; ++a) a_src[a] = A + (m0 + wave * WTM + a * 16 + pr) * lda + pc * 4;
#pragma unroll
    for (int j = 0; j < NB_MAX; ++j) {
      const int p = wave + j * NW;                             // piece index (wave-uniform)
      if (TB) {
        b_src[j] = B + (int64_t)(n0 + (p < NT ? p : 0) * 16 + pr) * ldb + pc * 4;
      } else {
        constexpr int NCH = BN / 4;                            // 16-B chunks per k-row
        const int q = (p < NT ? p : 0) * 64 + lane;
        const int r = q / NCH, cc = q % NCH;
        const int gc = (cc + 4 * ((r >> 2) & 1)) % NCH;
        b_src[j] = B + (int64_t)r * ldb + n0 + gc * 4;
      }
    }
  };
  const int64_t b_step = TB ? (int64_t)kBK : (int64_t)kBK * ldb;
  const bool b_hi = wave < NB_REM;                             // this wave loads NB_LO + 1 pieces of B
  int64_t tile_i = first;                                      // issue cursor: (tile, k-step)
  int kt_i = 0;
  set_sources(tile_i);
  auto issue_next = [&](int stage) {
    const char* s = glds_smem + stage * STAGE;
#pragma unroll
    for (int a = 0; a < MT; ++a) glds16(a_src[a] + (int64_t)kt_i * kBK, s + (wave * MT + a) * 1024);
#pragma unroll
    for (int j = 0; j < NB_MAX; ++j) {
      if (j < NB_LO || b_hi) glds16(b_src[j] + (int64_t)kt_i * b_step, s + A_BYTES + (wave + j * NW) * 1024);
    }
    if (++kt_i == nkt) {                                       // block-uniform
      kt_i = 0;
      tile_i += stride;
      if (tile_i < tiles_total) set_sources(tile_i);
    }
  };
  // this wave's loads of one k-step; waiting down to that count leaves exactly the newest k-step in flight
  auto wait_keep_one_step = [&]() {
    if (b_hi) wait_vmcnt<MT + NB_LO + 1>(); else wait_vmcnt<MT + NB_LO>();
  };

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // ---- per-lane fragment addresses (byte offsets inside a stage)
  const int frag_slot = (li * 4 + (g ^ ((4 - (li >> 2)) & 3))) * 16;
  int b_off[NT];                                               // [K,N] image: byte offset of (k-row 4g, column block b)
#pragma unroll
  for (int b = 0; b < NT; ++b) b_off[b] = A_BYTES + ((4 * g) * BN + ((b + ((g & 1) ? NT - 1 : 0)) % NT) * 16 + li) * 4;

  // The MFMAs are issued with the operands SWAPPED (B fragment first): the accumulator tile is C^T, i.e. lane
  // (i = lane & 15, g) holds C[row i][columns 4g .. 4g+3] of its 16x16 tile — four CONSECUTIVE columns, so the tile
  // leaves as global_store_dwordx4 (64 B per row, 1 KB per wave instruction) and bias / aux operands come in as float4.
  // (Measured and dropped, profiles/r03_gemm_glds.txt: fragments double-buffered in registers with the next step's
  // ds_reads pinned into this step's MFMA stream — 124 -> 104-113 TF at K 1600, and 152 VGPRs at 256x80.)
  auto compute = [&](int stage) {
    const char* s = glds_smem + stage * STAGE;
    f32x4_t af[MT];
#pragma unroll
    for (int a = 0; a < MT; ++a) af[a] = *reinterpret_cast<const f32x4_t*>(s + (wave * MT + a) * 1024 + frag_slot);
    if constexpr (TB) {
      f32x4_t bf[NT];
#pragma unroll
      for (int b = 0; b < NT; ++b) bf[b] = *reinterpret_cast<const f32x4_t*>(s + A_BYTES + b * 1024 + frag_slot);
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
          for (int b = 0; b < NT; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[b][s_], af[a][s_], acc[a][b], 0, 0, 0);
    } else {
      float bf[NT][4];
#pragma unroll
      for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) bf[b][s_] = *reinterpret_cast<const float*>(s + b_off[b] + s_ * BN * 4);
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
          for (int b = 0; b < NT; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[b][s_], af[a][s_], acc[a][b], 0, 0, 0);
    }
  };

  // epilogue of the tile whose last k-step was just accumulated; resets the accumulators
  auto finish_tile = [&](int64_t t) {
    int64_t m0;
    int n0;
    tile_origin(t, m0, n0);
    f32x4_t bj[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      const int j = n0 + b * 16 + g * 4;
#pragma unroll
      for (int c = 0; c < 4; ++c) bj[b][c] = load_bias<EPI>(j + c, epi);
    }
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      const int64_t i = m0 + wave * WTM + a * 16 + li;
      f32x4_t x0[NT], x1[NT];
#pragma unroll
      for (int b = 0; b < NT; ++b) {
        const int j = n0 + b * 16 + g * 4;
        if constexpr (EpiUses<EPI>::aux0) x0[b] = *reinterpret_cast<const f32x4_t*>(epi.aux0 + i * epi.ld0 + j);
        else x0[b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if constexpr (EpiUses<EPI>::aux1) x1[b] = *reinterpret_cast<const f32x4_t*>(epi.aux1 + i * epi.ld1 + j);
        else x1[b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int b = 0; b < NT; ++b) {
        const int j = n0 + b * 16 + g * 4;
        f32x4_t v;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = apply_epi<EPI>(acc[a][b][c], x0[b][c], x1[b][c], bj[b][c], i, epi);
        *reinterpret_cast<f32x4_t*>(C + i * ldc + j) = v;
        acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
    }
  };

  // experiment knob: the second block of a CU starts `skew` x 64 cycles late, so that the two co-resident blocks sit in
  // opposite phases of the k-step (one reads LDS / waits at its barrier while the other feeds the matrix pipe)
  if (skew > 0 && (blockIdx.x / 256) % 2 == 1) {
    for (int i = 0; i < skew; ++i) __builtin_amdgcn_s_sleep(1);
  }
  // ---- prologue: steps 0 and 1 in flight, step 0 landed and published
  issue_next(0);
  if (total_steps > 1) {
    issue_next(1);
    wait_keep_one_step();
  } else {
    wait_vmcnt<0>();
  }
  __builtin_amdgcn_s_barrier();
  // ---- main loop over the k-steps of ALL of this block's tiles: one barrier per step; at its top step q is in LDS,
  // step q+1 is in flight
  int stage = 0, kt_c = 0;
  int64_t tile_c = first;
  for (int64_t q = 0; q < total_steps; ++q) {
    const int s2 = stage == 0 ? 2 : stage - 1;        // (stage + 2) % 3: last read during step q-1 (barrier passed)
    const bool more = q + 2 < total_steps;
    if (more) issue_next(s2);
    __builtin_amdgcn_sched_barrier(0);
    compute(stage);
    __builtin_amdgcn_sched_barrier(0);
    if (++kt_c == nkt) {                               // tile complete: its stores drain under the next tile's MFMAs
      finish_tile(tile_c);
      kt_c = 0;
      tile_c += stride;
    }
    if (q + 1 < total_steps) {
      if (more) wait_keep_one_step(); else wait_vmcnt<0>();   // my loads of step q+1 have landed
      __builtin_amdgcn_s_barrier();                           // ... and everybody else's; stage `stage` is free again
    }
    stage = stage == 2 ? 0 : stage + 1;
  }
}

}  // namespace rec
