// DIN attention-pool, fused (gfx950): gather -> attention MLP -> masked softmax over T -> weighted sum.
//
// Replaces /root/reference/models/rank/din/net.py:141-173
//   4 Embedding lookups -> concat [h, q, h-q, h*q] ([B,T,4E] in HBM) -> Linear/Sigmoid x2 -> Linear ->
//   + mask -> transpose -> scale(E^-0.5) -> softmax -> matmul(weight, h)
// with ONE kernel that never materialises the [B,T,4E] concat or the [B,T,80]/[B,T,40] activations:
// a block owns a sample and walks its history in tiles of 32 positions, flash-attention style
// (running max / running sum / running weighted sum of h, rescaled per tile), so the softmax over a
// variable-length history needs no second pass over HBM.  Per tile:
//   gather  h = [item_emb(hist_item), cat_emb(hist_cat)], q = [.. target seq ..] into LDS (float4, coalesced per row)
//   layer 1 thread (j, pg) owns hidden unit j for positions pg, pg+PG, ..:   acc += h*w_a + q*w_b + (h-q)*w_c + (h*q)*w_d
//           (the four E-row blocks of W1 are walked together, so the concat exists only as 4 FMAs)
//   layer 2 / layer 3 from LDS, sigmoid in registers
//   logits  s = (s + mask) * E^-0.5  kept in LDS for the whole history (softmax weights for the backward)
// Padded positions (mask = -1e9) get weight exp(-8.8e7 - m) = 0 exactly in fp32, as in the reference
// (SURVEY App. B-11); they are computed, not skipped, so an all-padding history still reproduces it.
#include <stdlib.h>
#include <string.h>

#include "rec_common.h"
#include "tail_roles.h"

namespace rec {

constexpr int kDinTP = 32;       // positions per tile
constexpr int kDinNP1 = 16;      // max positions per thread in layer 1 (PG >= 2)
constexpr int kDinNP2 = 16;
constexpr size_t kDinCtLdsMax = 160 * 1024;   // LDS of a gfx950 CU; two blocks are resident below half of it

struct DinArgs {
  int64_t B;
  int T, Ei, Ec, H1, H2;
  int64_t n_item, n_cat;
  int ld_item, ld_cat;
  const int64_t *hist_item, *hist_cat, *tgt_item, *tgt_cat, *mask;
  const float *w_hist_item, *w_hist_cat, *w_tgt_item, *w_tgt_cat;
  const float *w1, *b1, *w2, *b2, *w3, *b3;
  float *out, *att_weight;
  float* act1;             // [B,T,H1] layer-1 activations, saved for the backward when non-null
  int32_t* status;
  float* part;             // tile-split forward (few samples): [B * tiles][E + 2] = pooled sum / l, m, l of every tile
  unsigned int* ticket;    // large batches with a workspace: zeroed counter the blocks draw their next sample from
};

// -DREC_DIN_PHASE_TIMING: wave 0 of every block sums the shader cycles it spends in each phase of a tile into
// din_phase_dbg (tools/din_phase_probe.py reads it through rec_din_debug_phases) — a measurement build, never shipped.
#ifdef REC_DIN_PHASE_TIMING          // = 1: the forward kernel, = 2: the backward kernel
__device__ unsigned long long din_phase_dbg[16];
#define DIN_PH_DECL_ unsigned long long ph_t = __builtin_amdgcn_s_memtime(), ph_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define DIN_PH_(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); ph_acc[k] += n_ - ph_t; ph_t = n_; } while (0)
#define DIN_PH_FLUSH_ do { if (threadIdx.x == 0) for (int k_ = 0; k_ < 10; ++k_) atomicAdd(&din_phase_dbg[k_], ph_acc[k_]); } while (0)
#endif
#if defined(REC_DIN_PHASE_TIMING) && REC_DIN_PHASE_TIMING == 1
#define DIN_PH_DECL DIN_PH_DECL_
#define DIN_PH(k) DIN_PH_(k)
#define DIN_PH_FLUSH DIN_PH_FLUSH_
#else
#define DIN_PH_DECL
#define DIN_PH(k)
#define DIN_PH_FLUSH
#endif
#if defined(REC_DIN_PHASE_TIMING) && REC_DIN_PHASE_TIMING == 2
#define DIN_PHB_DECL DIN_PH_DECL_
#define DIN_PHB(k) DIN_PH_(k)
#define DIN_PHB_FLUSH DIN_PH_FLUSH_
#else
#define DIN_PHB_DECL
#define DIN_PHB(k)
#define DIN_PHB_FLUSH
#endif

// Block barrier that orders LDS traffic only: __syncthreads() also drains the wave's outstanding GLOBAL loads
// (s_waitcnt vmcnt(0)), which would end a prefetch issued in front of it.  Use only where no wave of the block reads
// global memory another wave of the block wrote before the barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float sigmoidf_(float z) { return 1.f / (1.f + expf(-z)); }
// sum over the 16 lanes of a DPP row (lane & 15), every lane gets it: the pairs of the xor-8/4/2/1 butterfly (rotations by
// 8 and 4 inside the row meet the same partner values), on the VALU instead of four ds_bpermute round trips
__device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, false));   // row_ror:8
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xF, 0xF, false));   // row_ror:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
  return v;
}
__device__ __forceinline__ float dout_k_of(const float* dout, int64_t b, int E, int k) { return dout[b * E + k]; }

typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int kDinNT1 = 5;   // hidden1 <= 80 on the MFMA path (5 column tiles of 16)

// Attention layer 1 on the matrix cores: z1[32, H1] = [h, q, h-q, h*q][32, 4E] @ W1[4E, H1] + b1, sigmoid.
// The block's 4 waves split K by SEGMENT of the concat (wave w owns rows w*E..(w+1)*E of W1), so every wave
// builds its A fragments with one fixed formula and the concat never exists anywhere; v_mfma_f32_16x16x4_f32
// with 2 x 5 accumulator tiles per wave.  The four partial sums are added into a1 in wave order (fixed
// order -> deterministic), then the sigmoid is applied in place.  Needs E % 16 == 0, H1 % 16 == 0, H1 <= 80.
__device__ __forceinline__ void din_layer1_mfma(const float* __restrict__ hs, const float* __restrict__ qs,
                                                int EP, const float* __restrict__ w1, const float* __restrict__ b1s,
                                                float* __restrict__ a1, int E, int H1) {
  const int lane = threadIdx.x % kWave, seg = threadIdx.x / kWave;
  const int li = lane & 15, g = lane >> 4;
  const int nt = H1 / 16;
  f32x4_t acc[2][kDinNT1];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < kDinNT1; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // W1 fragments come from L2 (the 4E x H1 matrix does not fit beside the tiles in LDS): the loads of the
  // next K block are issued before this block's MFMAs (register double buffering) to cover their latency.
  const int nkb = E / 16;
  float bf[2][kDinNT1][4];
  auto load_b = [&](int kb, float (&dst)[kDinNT1][4]) {
    const float* wrow = w1 + ((int64_t)(seg * E + kb * 16 + 4 * g)) * H1 + li;
#pragma unroll
    for (int b = 0; b < kDinNT1; ++b)
#pragma unroll
      for (int s = 0; s < 4; ++s) dst[b][s] = (b < nt) ? wrow[(int64_t)s * H1 + b * 16] : 0.f;
  };
  load_b(0, bf[0]);
  for (int kb0 = 0; kb0 < nkb; kb0 += 2) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int kb = kb0 + half;
      if (kb < nkb) {
        if (kb + 1 < nkb) load_b(kb + 1, bf[half ^ 1]);
        float av[2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int off = (a * 16 + li) * EP + kb * 16 + 4 * g;
          const float4 hv = *reinterpret_cast<const float4*>(hs + off);
          const float4 qv = *reinterpret_cast<const float4*>(qs + off);
          const float h4[4] = {hv.x, hv.y, hv.z, hv.w}, q4[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
          for (int s = 0; s < 4; ++s)
            av[a][s] = seg == 0 ? h4[s] : seg == 1 ? q4[s] : seg == 2 ? h4[s] - q4[s] : h4[s] * q4[s];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < kDinNT1; ++b)
              if (b < nt)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a][s], bf[half][b][s], acc[a][b], 0, 0, 0);
      }
    }
  }
  // C/D layout: col = lane & 15, row = (lane >> 4) * 4 + reg
  for (int w0 = 0; w0 < kBlock / kWave; ++w0) {
    if (seg == w0) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < kDinNT1; ++b)
          if (b < nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int p = a * 16 + g * 4 + r, j = b * 16 + li;
              const float base = (w0 == 0) ? b1s[j] : a1[p * H1 + j];
              a1[p * H1 + j] = base + acc[a][b][r];
            }
          }
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < kDinTP * H1; i += kBlock) a1[i] = sigmoidf_(a1[i]);
}

// Attention layer 2 on the matrix cores: a2[32, H2] = sigmoid(a1[32, H1] @ W2[H1, H2] + b2) — 2 row tiles x
// ceil(H2/16) column tiles of 16x16, dealt round-robin to the block's 4 waves (whole K per tile: no partial sums to
// combine); A fragments by ds_read_b128 from a1, B fragments from the LDS copy of W2.  On the VALU this layer cost
// 5 k instructions per wave and tile (runtime-shaped loops over the position groups) against 320 MFMAs of layer 1.
// Needs H1 % 16 == 0; columns >= H2 of the last tile are computed on zeros and dropped.
__device__ __forceinline__ void din_layer2_mfma(const float* __restrict__ a1, const float* __restrict__ w2s,
                                                const float* __restrict__ b2s, float* __restrict__ a2, int H1,
                                                int H2) {
  const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
  const int li = lane & 15, g = lane >> 4;
  const int nt = (H2 + 15) / 16;
  const int nkb = H1 / 16;
  for (int t = wave; t < 2 * nt; t += kBlock / kWave) {
    const int m = t & 1, n = t >> 1;
    const int col = n * 16 + li;
    const bool cok = col < H2;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int kb = 0; kb < nkb; ++kb) {
      const int k0 = kb * 16 + 4 * g;
      const float4 av = *reinterpret_cast<const float4*>(a1 + (m * 16 + li) * H1 + k0);
      const float a4[4] = {av.x, av.y, av.z, av.w};
      float b4[4];
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) b4[s_] = cok ? w2s[(k0 + s_) * H2 + col] : 0.f;
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s_], b4[s_], acc, 0, 0, 0);
    }
    if (cok) {
      const float bias = b2s[col];
#pragma unroll
      for (int r = 0; r < 4; ++r) a2[(m * 16 + g * 4 + r) * H2 + col] = sigmoidf_(acc[r] + bias);
    }
  }
}

__global__ __launch_bounds__(kBlock) void din_attention_fwd_kernel(DinArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int E = a.Ei + a.Ec, H1 = a.H1, H2 = a.H2, T = a.T;
  const int EP = E + 4;                // LDS row stride of h/q: +4 floats so the 16 rows of an MFMA
                                       // A-fragment read (one b128 per lane) fall on different banks
  float* hs = smem;                    // [TP][EP]
  float* qs = hs + kDinTP * EP;        // [TP][EP]
  float* a1 = qs + kDinTP * EP;        // [TP][H1]
  float* a2 = a1 + kDinTP * H1;        // [TP][H2]
  float* w2s = a2 + kDinTP * H2;       // [H1][H2]
  float* w3s = w2s + H1 * H2;          // [H2]
  float* b1s = w3s + H2;               // [H1]
  float* b2s = b1s + H1;               // [H2]
  float* sall = b2s + H2;              // [T] scaled logits of the whole history
  const int tid = threadIdx.x;
  for (int i = tid; i < H1 * H2; i += kBlock) w2s[i] = a.w2[i];
  for (int i = tid; i < H2; i += kBlock) { w3s[i] = a.w3[i]; b2s[i] = a.b2[i]; }
  for (int i = tid; i < H1; i += kBlock) b1s[i] = a.b1[i];
  const float b3 = a.b3[0];
  const float scale = 1.f / sqrtf((float)E);          // net.py:168  firInDim ** -0.5
  const int j1 = tid % H1, pg1 = tid / H1, PG1 = kBlock / H1;
  const int NP1 = (kDinTP + PG1 - 1) / PG1;
  const bool on1 = pg1 < PG1;
  const int j2 = tid % H2, pg2 = tid / H2, PG2 = kBlock / H2;
  const int NP2 = (kDinTP + PG2 - 1) / PG2;
  const bool on2 = pg2 < PG2;
  const int e4 = E / 4;
  const bool use_mfma = (E % 16 == 0) && (H1 % 16 == 0) && H1 <= 16 * kDinNT1;   // block-uniform
  const bool use_mfma2 = H1 % 16 == 0;
  int oob = 0;
  __syncthreads();

  for (int64_t b = blockIdx.x; b < a.B; b += gridDim.x) {
    float m_run = -INFINITY, l_run = 0.f, acc = 0.f;   // acc: thread d < E owns out[b][d]
    for (int t0 = 0; t0 < T; t0 += kDinTP) {
      // ---- gather h and q rows of this tile into LDS (net.py:141-151)
      for (int v = tid; v < kDinTP * e4; v += kBlock) {
        const int p = v / e4, c4 = (v % e4) * 4;
        const int t = t0 + p;
        float4 hv = make_float4(0.f, 0.f, 0.f, 0.f), qv = hv;
        if (t < T) {
          const bool item = c4 < a.Ei;
          const int64_t hid = item ? a.hist_item[b * T + t] : a.hist_cat[b * T + t];
          const int64_t qid = item ? a.tgt_item[b * T + t] : a.tgt_cat[b * T + t];
          const int64_t nrow = item ? a.n_item : a.n_cat;
          const int ld = item ? a.ld_item : a.ld_cat;
          const int c = item ? c4 : c4 - a.Ei;
          const float* wh = item ? a.w_hist_item : a.w_hist_cat;
          const float* wq = item ? a.w_tgt_item : a.w_tgt_cat;
          if (hid >= 0 && hid < nrow) hv = *reinterpret_cast<const float4*>(wh + hid * ld + c); else oob = 1;
          if (qid >= 0 && qid < nrow) qv = *reinterpret_cast<const float4*>(wq + qid * ld + c); else oob = 1;
        }
        *reinterpret_cast<float4*>(hs + p * EP + c4) = hv;
        *reinterpret_cast<float4*>(qs + p * EP + c4) = qv;
      }
      __syncthreads();
      // ---- attention layer 1: [h, q, h-q, h*q] @ W1 + b1, sigmoid (net.py:155-164)
      if (use_mfma) {
        din_layer1_mfma(hs, qs, EP, a.w1, b1s, a1, E, H1);
      } else if (on1) {
        float s1[kDinNP1];
#pragma unroll
        for (int i = 0; i < kDinNP1; ++i) s1[i] = 0.f;
        const float* w = a.w1 + j1;
        for (int kk = 0; kk < E; ++kk) {
          const float wa = w[(int64_t)kk * H1], wb = w[(int64_t)(E + kk) * H1];
          const float wc = w[(int64_t)(2 * E + kk) * H1], wd = w[(int64_t)(3 * E + kk) * H1];
#pragma unroll
          for (int i = 0; i < kDinNP1; ++i) {
            const int p = pg1 + i * PG1;
            if (i < NP1 && p < kDinTP) {
              const float hv = hs[p * EP + kk], qv = qs[p * EP + kk];
              s1[i] += hv * wa + qv * wb + (hv - qv) * wc + (hv * qv) * wd;
            }
          }
        }
#pragma unroll
        for (int i = 0; i < kDinNP1; ++i) {
          const int p = pg1 + i * PG1;
          if (i < NP1 && p < kDinTP) a1[p * H1 + j1] = sigmoidf_(s1[i] + b1s[j1]);
        }
      }
      __syncthreads();
      // ---- layer 2
      if (use_mfma2) {
        din_layer2_mfma(a1, w2s, b2s, a2, H1, H2);
      } else if (on2) {
        float s2[kDinNP2];
#pragma unroll
        for (int i = 0; i < kDinNP2; ++i) s2[i] = 0.f;
        for (int k = 0; k < H1; ++k) {
          const float wv = w2s[k * H2 + j2];
#pragma unroll
          for (int i = 0; i < kDinNP2; ++i) {
            const int p = pg2 + i * PG2;
            if (i < NP2 && p < kDinTP) s2[i] += a1[p * H1 + k] * wv;
          }
        }
#pragma unroll
        for (int i = 0; i < kDinNP2; ++i) {
          const int p = pg2 + i * PG2;
          if (i < NP2 && p < kDinTP) a2[p * H2 + j2] = sigmoidf_(s2[i] + b2s[j2]);
        }
      }
      __syncthreads();
      // ---- layer 3 + mask + scale (net.py:166-168)
      if (tid < kDinTP) {
        const int t = t0 + tid;
        float s = -INFINITY;
        if (t < T) {
          s = b3;
          for (int k = 0; k < H2; ++k) s += a2[tid * H2 + k] * w3s[k];
          s = (s + (float)a.mask[b * T + t]) * scale;
          sall[t] = s;
        }
        a2[tid * H2] = s;   // a2[p][0] doubles as the tile's logit slot (layer 2 output no longer needed)
      }
      __syncthreads();
      // ---- online softmax + weighted sum of h (net.py:169-171).  The 32 exponentials of the tile are
      // computed once (threads 0..31) and shared through LDS instead of once per thread.
      float mt = -INFINITY;
      for (int p = 0; p < kDinTP; ++p) mt = fmaxf(mt, a2[p * H2]);
      const float m_new = fmaxf(m_run, mt);
      const float f = (m_run == -INFINITY) ? 0.f : expf(m_run - m_new);
      __syncthreads();                       // everyone has read the logits before they are replaced
      if (tid < kDinTP) {
        const float sp = a2[tid * H2];
        a2[tid * H2] = (sp == -INFINITY) ? 0.f : expf(sp - m_new);
      }
      __syncthreads();
      float lsum = 0.f, asum = 0.f;
      for (int p = 0; p < kDinTP; ++p) {
        const float e = a2[p * H2];
        lsum += e;
        if (tid < E) asum += e * hs[p * EP + tid];
      }
      l_run = l_run * f + lsum;
      acc = acc * f + asum;
      m_run = m_new;
      __syncthreads();   // hs/qs/a2 are rewritten by the next tile
    }
    if (tid < E) a.out[b * E + tid] = acc / l_run;
    if (a.att_weight) {
      for (int t = tid; t < T; t += kBlock) a.att_weight[b * T + t] = expf(sall[t] - m_run) / l_run;
    }
    __syncthreads();
  }
  if (oob) atomicOr(a.status, REC_FLAG_INDEX_OOB);
}

// ------------------------------------------------------------------------- forward, compile-time shapes
// The same attention-pool for shapes known at compile time (the reference net: E = 128, attention MLP 80-40-1,
// din/net.py:60-75).  What changes against the runtime-shaped kernel above, all of it to keep the matrix cores fed:
//   * the wave's K-segment of W1 ([E, H1] of the [4E, H1] matrix) lives in registers for the life of the block
//     (E/16 x H1/16 x 4 = 160 VGPRs at 128/80): no W1 traffic and no address arithmetic inside the tile loop;
//   * <= 256 VGPRs and < 80 KB LDS, so TWO blocks are resident per CU and one block's gathers / barriers overlap
//     the other's MFMAs (the runtime-shaped kernel sits at one wave per SIMD and waits 63 % of its cycles);
//   * the ids and mask of tile k+1 are fetched during tile k's layer 1, so the row gather at the end of tile k is one
//     HBM round trip, not a dependent id -> row chain (fetching the rows themselves a phase earlier, template
//     parameter PF, costs 40 more spilled registers than it hides latency: measured slower);
//   * the four K-segment partial sums are combined as a two-level tree (2 barriers instead of 4), layer 3 is folded
//     into layer 2's epilogue (a 16-lane reduction of sigmoid(z2) * w3), and the softmax bookkeeping is done once by
//     wave 0 instead of by every thread.
// Summation order differs from the kernel above only inside layer 3 and the final pool (both within fp32 rounding
// of the same values); every order is fixed, so results are reproducible run to run.
template <int E, int H1, int H2>
struct DinCt {
  static_assert(E % 32 == 0 && kBlock % E == 0 && H1 % 16 == 0 && H1 % 4 == 0, "unsupported compile-time shape");
  static constexpr int EP = E + 4;                       // row stride of hs / qs   (== 4 mod 8: conflict-free b128)
  static constexpr int H1P = H1 + 4;                     // row stride of X / Y
  static constexpr int H2C = ((H2 + 15) / 16) * 16;      // layer-2 columns padded to whole MFMA tiles (zeros)
  static constexpr int W2P = H2C + 4;                    // row stride of the LDS copy of W2
  static constexpr int NKB = E / 16, NT1 = H1 / 16, NT2 = H2C / 16;
  static constexpr int E4 = E / 4, NIT = kDinTP * E4 / kBlock, PSTEP = kBlock / E4;
  static constexpr int NPART = kBlock / E, PPART = kDinTP / NPART;
  static constexpr int kHs = 0;
  static constexpr int kQs = kHs + kDinTP * EP;
  static constexpr int kX = kQs + kDinTP * EP;
  static constexpr int kY = kX + kDinTP * H1P;
  static constexpr int kW2 = kY + kDinTP * H1P;
  static constexpr int kW3 = kW2 + H1 * W2P;
  static constexpr int kB1 = kW3 + H2C;
  static constexpr int kB2 = kB1 + H1;
  static constexpr int kLp = kB2 + H2C;                  // [NT2][32] layer-3 partial dots
  static constexpr int kEs = kLp + NT2 * kDinTP;         // [32] exp weights of the tile, [32] f, [33] l_run, [34] m_run
  static constexpr int kRed = kEs + 44;                  // (es[35] skip flag, [36..39] mask scan, [40..41] sample tickets)  [NPART][E] end-of-sample combine
  static constexpr int kIds = (kRed + NPART * E + 1) & ~1;   // int64 [2][5][32]: ids + mask of this / the next tile
  static constexpr int kSall = kIds + 2 * 2 * 5 * kDinTP;
  static size_t lds_bytes(int T) { return sizeof(float) * ((size_t)kSall + (size_t)T); }
};

template <int SEG, int E, int H1>
__device__ __forceinline__ void din_l1_segment(const float* __restrict__ hs, const float* __restrict__ qs,
                                               const float (&bf)[E / 16][H1 / 16][4],
                                               f32x4_t (&acc)[2][H1 / 16], int li, int g) {
  constexpr int EP = E + 4, NKB = E / 16;
  // A fragments: lane (li, g) holds row a*16 + li, k = kb*16 + 4g .. +3 of the segment's operand
  auto load_a = [&](int kb, float (&av)[2][4]) {
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int off = (a * 16 + li) * EP + kb * 16 + 4 * g;
      float4 hv, qv;
      if (SEG != 1) hv = *reinterpret_cast<const float4*>(hs + off);
      if (SEG != 0) qv = *reinterpret_cast<const float4*>(qs + off);
      if (SEG == 0) { av[a][0] = hv.x; av[a][1] = hv.y; av[a][2] = hv.z; av[a][3] = hv.w; }
      if (SEG == 1) { av[a][0] = qv.x; av[a][1] = qv.y; av[a][2] = qv.z; av[a][3] = qv.w; }
      if (SEG == 2) { av[a][0] = hv.x - qv.x; av[a][1] = hv.y - qv.y; av[a][2] = hv.z - qv.z; av[a][3] = hv.w - qv.w; }
      if (SEG == 3) { av[a][0] = hv.x * qv.x; av[a][1] = hv.y * qv.y; av[a][2] = hv.z * qv.z; av[a][3] = hv.w * qv.w; }
    }
  };
  // The operands are SWAPPED (W1 fragment as A, activations as B): the tile comes out transposed — lane (li, g) holds
  // columns b*16 + 4g .. + 3 of position a*16 + li — so a lane's four values are contiguous in the row-major [p][c]
  // buffers and the partial sums move as 16-byte LDS accesses instead of four 4-byte ones (same products, same order).
  // one K block ahead in registers; the scheduling fences keep the compiler from hoisting all E/16 blocks' LDS
  // reads to the top (64 more live registers - the W1 fragments would spill)
  float av[2][2][4];
  load_a(0, av[0]);
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    if (kb + 1 < NKB) load_a(kb + 1, av[(kb + 1) & 1]);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < H1 / 16; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[kb][b][s], av[kb & 1][a][s], acc[a][b], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Layer 2 (+ layer 3 in the epilogue) for J of a wave's (m, n) tiles n = n0, n0 + 2, ...: lp[n][p] = sum_c a2[p][c] w3[c].
// J is a compile-time count so that the k loop has no branch in it (a wave-uniform `if` per MFMA compiled to one scalar
// branch, one LDS read and one s_waitcnt per MFMA: 40 serial LDS round trips per tile).
template <int J, class S>
__device__ __forceinline__ void din_l2_tiles(const float* __restrict__ X, const float* __restrict__ w2s,
                                             const float* __restrict__ b2s, const float* __restrict__ w3s,
                                             float* __restrict__ lp, int m, int n0, int li, int g) {
  if constexpr (J > 0) {
    f32x4_t z[J];
#pragma unroll
    for (int j = 0; j < J; ++j) z[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < S::NT1; ++kb) {
      const int k0 = kb * 16 + 4 * g;
      const float4 av = *reinterpret_cast<const float4*>(X + (m * 16 + li) * S::H1P + k0);
      const float a4[4] = {av.x, av.y, av.z, av.w};
      float bv[4][J];
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < J; ++j) bv[s][j] = w2s[(k0 + s) * S::W2P + (n0 + 2 * j) * 16 + li];
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < J; ++j) z[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s], bv[s][j], z[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int n = n0 + 2 * j, c = n * 16 + li;
      const float bias = b2s[c], w3c = w3s[c];      // columns >= H2 carry w3 = 0
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = row16_sum(sigmoidf_(z[j][r] + bias) * w3c);
        if (li == 0) lp[n * kDinTP + m * 16 + g * 4 + r] = v;
      }
    }
  }
}

template <int E, int H1, int H2, bool PF, int OCC>
__global__ __launch_bounds__(kBlock, OCC) void din_attention_fwd_ct_kernel(DinArgs a) {
  using S = DinCt<E, H1, H2>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* hs = smem + S::kHs;
  float* qs = smem + S::kQs;
  float* X = smem + S::kX;
  float* Y = smem + S::kY;
  float* w2s = smem + S::kW2;
  float* w3s = smem + S::kW3;
  float* b1s = smem + S::kB1;
  float* b2s = smem + S::kB2;
  float* lp = smem + S::kLp;
  float* es = smem + S::kEs;
  float* red = smem + S::kRed;
  int64_t* idbuf = reinterpret_cast<int64_t*>(smem + S::kIds);
  float* sall = smem + S::kSall;
  const int tid = threadIdx.x, lane = tid % kWave;
  const int seg = __builtin_amdgcn_readfirstlane(tid / kWave);   // wave index == K segment of the concat
  const int li = lane & 15, g = lane >> 4;
  const int T = a.T;
  for (int i = tid; i < H1 * S::W2P; i += kBlock) {
    const int k = i / S::W2P, c = i % S::W2P;
    w2s[i] = c < H2 ? a.w2[k * H2 + c] : 0.f;
  }
  for (int i = tid; i < S::H2C; i += kBlock) {
    w3s[i] = i < H2 ? a.w3[i] : 0.f;
    b2s[i] = i < H2 ? a.b2[i] : 0.f;
  }
  for (int i = tid; i < H1; i += kBlock) b1s[i] = a.b1[i];
  const float b3 = a.b3[0];
  const float scale = 1.f / sqrtf((float)E);          // net.py:168  firInDim ** -0.5
  // this wave's segment of W1 as MFMA B fragments: bf[kb][b][s] = W1[seg*E + kb*16 + 4g + s][b*16 + li]
  float bf[S::NKB][S::NT1][4];
#pragma unroll
  for (int kb = 0; kb < S::NKB; ++kb)
#pragma unroll
    for (int b = 0; b < S::NT1; ++b)
#pragma unroll
      for (int s = 0; s < 4; ++s)
        bf[kb][b][s] = a.w1[(int64_t)(seg * E + kb * 16 + 4 * g + s) * H1 + b * 16 + li];

  // gather geometry: a thread always fetches the same 4 columns, of rows p0, p0 + PSTEP, ...
  const int c4 = (tid % S::E4) * 4, p0 = tid / S::E4;
  const bool item = c4 < a.Ei;
  const float* wh = item ? a.w_hist_item : a.w_hist_cat;
  const float* wq = item ? a.w_tgt_item : a.w_tgt_cat;
  const int64_t nrow = item ? a.n_item : a.n_cat;
  const int ld = item ? a.ld_item : a.ld_cat;
  const int col = item ? c4 : c4 - a.Ei;
  const int ih = item ? 0 : 1, iq = item ? 2 : 3;
  // id prefetch geometry: threads 0..159 = 5 arrays x 32 positions
  const int id_arr = tid >> 5, id_p = tid & 31;
  const int64_t* id_ptr = id_arr == 0 ? a.hist_item : id_arr == 1 ? a.hist_cat : id_arr == 2 ? a.tgt_item
                          : id_arr == 3 ? a.tgt_cat : a.mask;
  int oob = 0;

  auto ids_issue = [&](int64_t b, int t0) -> int64_t {
    const int t = t0 + id_p;
    return (id_arr < 5 && t < T) ? id_ptr[b * T + t] : (int64_t)0;
  };
  auto ids_store = [&](int buf, int64_t v) {
    if (id_arr < 5) idbuf[(buf * 5 + id_arr) * kDinTP + id_p] = v;
  };
  // The gathered rows of the NEXT tile live in the registers of this tile's layer-1 accumulators (dead once the partial
  // sums are in LDS): row it of h in acc[0][it], of q in acc[1][it].  Issued right behind the partial-sum tree, in flight
  // through sigmoid / layer 2 / softmax / pool, stored to LDS at the end of the tile: the gather latency (10.6 k of a
  // tile's 57 k cycles when issued at the end, tools/din_phase_probe.py) is hidden and costs no register.
  static_assert(S::NIT <= S::NT1, "row prefetch aliases the accumulator tiles");
  f32x4_t acc[2][S::NT1];
  // Branch-free: a lookup that must read as zero (position >= T, id out of range) fetches row 0 and is zeroed when the
  // rows are stored — a load inside a divergent branch is followed by s_waitcnt vmcnt(0) at the join, which serialised
  // the eight row fetches of a thread into eight memory round trips (10.6 k of a tile's 57 k cycles).
  unsigned rows_ok = 0;                                  // bit it: h row valid, bit 8 + it: q row valid
  auto rows_issue = [&](int buf, int t0) {
    rows_ok = 0;
#pragma unroll
    for (int it = 0; it < S::NIT; ++it) {
      const int p = p0 + it * S::PSTEP;
      const bool inb = t0 + p < T;
      const int64_t hid = idbuf[(buf * 5 + ih) * kDinTP + p], qid = idbuf[(buf * 5 + iq) * kDinTP + p];
      const bool hok = hid >= 0 && hid < nrow, qok = qid >= 0 && qid < nrow;
      if (inb && !(hok && qok)) oob = 1;
      acc[0][it] = *reinterpret_cast<const f32x4_t*>(wh + (hok ? hid : 0) * ld + col);
      acc[1][it] = *reinterpret_cast<const f32x4_t*>(wq + (qok ? qid : 0) * ld + col);
      rows_ok |= (inb && hok ? 1u : 0u) << it | (inb && qok ? 1u : 0u) << (8 + it);
    }
  };
  auto rows_store = [&]() {
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < S::NIT; ++it) {
      const int p = p0 + it * S::PSTEP;
      *reinterpret_cast<f32x4_t*>(hs + p * S::EP + c4) = (rows_ok >> it & 1u) ? acc[0][it] : zero;
      *reinterpret_cast<f32x4_t*>(qs + p * S::EP + c4) = (rows_ok >> (8 + it) & 1u) ? acc[1][it] : zero;
    }
  };

  // Tile split (a.part != null: fewer samples than the chip has block slots — the reference's batch size 32): a work
  // item is ONE 32-position tile of a sample, softmax-normalised over its own positions; the per-tile (pooled, m, l) go
  // to a.part and din_combine_kernel rescales them to the softmax over the whole history.  Otherwise a block walks the
  // tiles of a sample one after the other with a running (online) softmax, as at large batches.
  const bool split = a.part != nullptr;
  const int NTL = (T + kDinTP - 1) / kDinTP;
  const int64_t W = split ? a.B * NTL : a.B;
  // Samples cost what their history holds (1 .. T/32 walked tiles): dealt round-robin, the slowest block of a B 4096
  // batch of uniform lengths works on 26 tiles while the average works on 17.  With a ticket counter (a.ticket, zeroed by
  // the launcher in the caller's workspace) a block draws its next sample when it finishes one — the draw is issued at
  // the top of a sample's last tile, one sample ahead, so its latency never shows.  Which block computes a sample does
  // not change the sample's arithmetic.  Without a workspace: round-robin as before.
  const bool dyn = !split && a.ticket != nullptr;
  int* ies = reinterpret_cast<int*>(es);
  int64_t w = blockIdx.x, wn = w + gridDim.x;
  if (dyn) {
    if (tid == 0) {
      ies[40] = (int)atomicAdd(a.ticket, 1u);
      ies[41] = (int)atomicAdd(a.ticket, 1u);
    }
    __syncthreads();
    w = ies[40];
    wn = ies[41];
    __syncthreads();
  }
  if (w >= W) return;
  int64_t b = split ? w / NTL : w;
  int t0 = split ? (int)(w % NTL) * kDinTP : 0, buf = 0;
  ids_store(0, ids_issue(b, t0));
  __syncthreads();
  rows_issue(0, t0);
  rows_store();
  __syncthreads();
  float m_run = -INFINITY, l_run = 0.f;     // live in wave 0
  float pool = 0.f;                         // thread (part, d): partial of out[b][d] over its positions
  const int pd = tid % E, part = tid / E;
  // A tile whose 32 positions are ALL padding (mask <= -1e6: the reader's -1e9, din/dinReader.py:81-84) behind a tile that
  // already produced a finite running maximum contributes exactly nothing: its logits are <= -8.8e4 + O(1), so its
  // weights exp(s - m) underflow to 0.0f and the online-softmax state (m, l, pool) does not move.  Such tiles are not
  // computed (block-uniform flag, decided by wave 0 for the NEXT tile of the same sample): their weights are written as
  // 0, their layer-1 activations are not saved — the backward skips a tile whose 32 saved weights are all zero.  Padding
  // in front of the first valid position, and a history that is padding only (uniform weights, App. B-11), are computed
  // as before.  The batches the reference's reader builds (lengths 1..152 padded to the batch maximum) are mostly such
  // tiles.
  bool skip = false;
  // ... and the block does not even walk the padded TAIL of a history: Teff = 1 + the last position with mask > -1e6 (T
  // when there is none) ends the sample's tile loop.  The scan of the next sample's mask row is issued at the top of the
  // current sample's last tile and folded at its end (es[36..39]: one partial per wave).
  //  (walked: everything up to the last position with mask > -1e6; skipped only when some position has mask > -1e5 —
  //   a gap of >= 9e5 * E^-0.5 between a kept logit and the dropped ones, far beyond the 104 at which expf underflows)
  auto scan_issue = [&](int64_t bb) -> int {
    int last = 0, any = 0;
    for (int t = tid; t < T; t += kBlock) {
      const int64_t mv = a.mask[bb * T + t];
      if (mv > (int64_t)-1000000) last = t + 1;                   // ascending t: the last hit wins
      if (mv > (int64_t)-100000) any = 1 << 20;
    }
    return last | any;
  };
  auto scan_fold = [&](int sc) {            // wave maximum of the position, OR of the flag -> es[36 + wave]
    int last = sc & 0xFFFFF, any = sc >> 20;
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
      last = max(last, __shfl_xor(last, o, kWave));
      any |= __shfl_xor(any, o, kWave);
    }
    if (lane == 0) es[36 + seg] = (float)(last | (any << 20));
  };
  auto scan_read = [&]() -> int {
    int last = 0, any = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int v = (int)es[36 + k];
      last = max(last, v & 0xFFFFF);
      any |= v >> 20;
    }
    return (any && last > 0) ? last : T;
  };
  int Teff = T;
  if (!split) {
    scan_fold(scan_issue(b));
    for (int t = tid; t < T; t += kBlock) sall[t] = -INFINITY;
    __syncthreads();
    Teff = scan_read();
  }

  DIN_PH_DECL;
  while (true) {
    DIN_PH(9);
    int64_t nb = b;
    int nt0 = t0 + kDinTP;
    const bool last_tile = split || nt0 >= Teff;
    const int64_t nw = split ? w + gridDim.x : wn;
    int tk = 0;
    if (dyn && last_tile && tid == 0) tk = (int)atomicAdd(a.ticket, 1u);     // the sample after the next one
    if (split) { nb = nw / NTL; nt0 = (int)(nw % NTL) * kDinTP; }
    else if (last_tile) { nb = nw; nt0 = 0; }
    const bool has_next = split ? nw < W : nb < a.B;
    int64_t idv = 0;
    if (has_next) idv = ids_issue(nb, nt0);
    int scan_next = 0;
    if (!split && last_tile && has_next) scan_next = scan_issue(nb);

    // ---- layer 1: this wave's K segment of [h, q, h-q, h*q] @ W1 (net.py:155-164)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < S::NT1; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (!skip) {
      if (seg == 0) din_l1_segment<0, E, H1>(hs, qs, bf, acc, li, g);
      else if (seg == 1) din_l1_segment<1, E, H1>(hs, qs, bf, acc, li, g);
      else if (seg == 2) din_l1_segment<2, E, H1>(hs, qs, bf, acc, li, g);
      else din_l1_segment<3, E, H1>(hs, qs, bf, acc, li, g);
    }
    DIN_PH(0);
    if (has_next) ids_store(buf ^ 1, idv);
    // partial sums, two-level tree in a fixed order: X = (b1 + P0) + P2, Y = P1 + P3, z1 = X + Y.
    // C/D layout of the swapped product: position = a*16 + (lane & 15), columns = j*16 + (lane >> 4) * 4 + reg
    {
      float* dst = (seg & 1) ? Y : X;
      if (seg < 2 && !skip) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < S::NT1; ++j) {
            const int p = i * 16 + li, c = j * 16 + g * 4;
            f32x4_t v = acc[i][j];
            if (seg == 0) {
              const f32x4_t bb = *reinterpret_cast<const f32x4_t*>(b1s + c);
              v = f32x4_t{bb[0] + v[0], bb[1] + v[1], bb[2] + v[2], bb[3] + v[3]};
            }
            *reinterpret_cast<f32x4_t*>(dst + p * S::H1P + c) = v;
          }
      }
      __syncthreads();
      if (seg >= 2 && !skip) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < S::NT1; ++j) {
            const int p = i * 16 + li, c = j * 16 + g * 4;
            f32x4_t* q = reinterpret_cast<f32x4_t*>(dst + p * S::H1P + c);
            const f32x4_t o = *q, v = acc[i][j];
            *q = f32x4_t{o[0] + v[0], o[1] + v[1], o[2] + v[2], o[3] + v[3]};
          }
      }
      __syncthreads();
    }
    DIN_PH(1);
    // embedding rows of the next tile: in flight during layers 2/3 and the softmax of this one
    if (PF && has_next) rows_issue(buf ^ 1, nt0);
    for (int v = tid; v < kDinTP * (H1 / 4) && !skip; v += kBlock) {
      const int p = v / (H1 / 4), c = (v % (H1 / 4)) * 4;
      const float4 x = *reinterpret_cast<const float4*>(X + p * S::H1P + c);
      const float4 y = *reinterpret_cast<const float4*>(Y + p * S::H1P + c);
      float4 z;
      z.x = sigmoidf_(x.x + y.x); z.y = sigmoidf_(x.y + y.y); z.z = sigmoidf_(x.z + y.z); z.w = sigmoidf_(x.w + y.w);
      *reinterpret_cast<float4*>(X + p * S::H1P + c) = z;
      if (a.act1 && t0 + p < T) *reinterpret_cast<float4*>(a.act1 + ((int64_t)b * T + t0 + p) * H1 + c) = z;
    }
    lds_barrier();     // the next tile's rows stay in flight
    DIN_PH(2);
    // ---- layer 2 on the matrix cores, layer 3 folded into its epilogue: lp[n][p] = sum_{c in tile n} a2[p][c] w3[c]
    // Wave `seg` owns the (m, n) tiles seg, seg + 4, ...: the same 16 rows m = seg & 1 for all of them, so the A fragments
    // are read once and the tiles' accumulator chains (20 dependent MFMAs each) are issued interleaved.
    if (!skip) {
      constexpr int MAXJ = (2 * S::NT2 + 3) / 4;
      const int m = seg & 1, n0 = seg >> 1;
      if (n0 + 2 * (MAXJ - 1) < S::NT2) din_l2_tiles<MAXJ, S>(X, w2s, b2s, w3s, lp, m, n0, li, g);
      else din_l2_tiles<MAXJ - 1, S>(X, w2s, b2s, w3s, lp, m, n0, li, g);
    }
    lds_barrier();     // the next tile's rows stay in flight
    DIN_PH(3);
    // ---- logits + online softmax bookkeeping, wave 0 (net.py:166-170)
    if (seg == 0) {
      const int p = lane & 31, t = t0 + p;
      if (!skip) {
        float s = -INFINITY;
        if (t < T) {
          s = b3;
#pragma unroll
          for (int n = 0; n < S::NT2; ++n) s += lp[n * kDinTP + p];
          s = (s + (float)idbuf[(buf * 5 + 4) * kDinTP + p]) * scale;
          if (lane < 32) sall[t] = s;
        }
        float mt = s;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) mt = fmaxf(mt, __shfl_xor(mt, o, kWave));
        const float m_new = fmaxf(m_run, mt);
        const float f = (m_run == -INFINITY) ? 0.f : expf(m_run - m_new);
        const float e = (s == -INFINITY) ? 0.f : expf(s - m_new);
        float lsum = e;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o, kWave);
        l_run = l_run * f + lsum;
        m_run = m_new;
        if (lane < 32) es[p] = e;
        if (lane == 0) { es[32] = f; es[33] = l_run; es[34] = m_run; }
      } else if (lane < 32 && t < T) {
        sall[t] = -INFINITY;             // weight exp(-inf - m) / l = 0 at the end of the sample; es[33] / es[34] stand
      }
      // the NEXT tile of this sample: all padding behind a finite maximum -> skipped (see the top of the loop)
      bool padn = true;
      if (has_next && !last_tile) {
        const int tn = nt0 + p;
        padn = tn >= T || idbuf[((buf ^ 1) * 5 + 4) * kDinTP + p] <= (int64_t)-1000000;
      }
      const bool all_pad = __builtin_amdgcn_ballot_w64(padn) == ~0ull;
      if (lane == 0) es[35] = (has_next && !last_tile && m_run > -1e5f && all_pad) ? 1.f : 0.f;
      if (last_tile) { m_run = -INFINITY; l_run = 0.f; }
    }
    lds_barrier();     // the next tile's rows stay in flight
    DIN_PH(4);
    // ---- weighted sum of h (net.py:171): thread (part, d) covers PPART positions of column d
    {
      if (!skip) {
        const float f = es[32];
        float asum = 0.f;
#pragma unroll
        for (int i = 0; i < S::PPART; ++i) {
          const int p = part * S::PPART + i;
          asum += es[p] * hs[p * S::EP + pd];
        }
        pool = pool * f + asum;
      }
      if (last_tile) { red[part * E + pd] = pool; pool = 0.f; }
    }
    lds_barrier();     // the next tile's rows stay in flight
    if (last_tile) {
      const float l_fin = es[33], m_fin = es[34];
      if (tid < E) {
        float o = 0.f;
#pragma unroll
        for (int q = 0; q < S::NPART; ++q) o += red[q * E + tid];
        if (split) a.part[w * (E + 2) + tid] = o / l_fin;
        else a.out[b * E + tid] = o / l_fin;
      }
      if (split) {
        if (tid == 0) { a.part[w * (E + 2) + E] = m_fin; a.part[w * (E + 2) + E + 1] = l_fin; }
        const int t = t0 + tid;
        if (a.att_weight && tid < kDinTP && t < T) a.att_weight[b * T + t] = expf(sall[t] - m_fin) / l_fin;
      } else {
        for (int t = tid; t < T; t += kBlock) {
          if (a.att_weight) a.att_weight[b * T + t] = expf(sall[t] - m_fin) / l_fin;   // never-walked tail: exp(-inf) = 0
          sall[t] = -INFINITY;                                                          // for the next sample
        }
      }
    }
    DIN_PH(5);
    if (!has_next) break;
    if (!PF) rows_issue(buf ^ 1, nt0);
    DIN_PH(6);
    rows_store();
    if (!split && last_tile) scan_fold(scan_next);
    if (dyn && last_tile && tid == 0) ies[40] = tk;
    __syncthreads();
    if (!split && last_tile) Teff = scan_read();
    skip = es[35] != 0.f && !split;
    if (last_tile && !split) wn = dyn ? (int64_t)ies[40] : wn + gridDim.x;
    if (last_tile) w = nw;
    b = nb; t0 = nt0; buf ^= 1;
    DIN_PH(7);
  }
  DIN_PH_FLUSH;
  if (oob) atomicOr(a.status, REC_FLAG_INDEX_OOB);
}

// softmax over the whole history from the per-tile pieces of the tile-split forward: with M = max_j m_j and
// L = sum_j l_j e^(m_j - M), tile j weighs c_j = l_j e^(m_j - M) / L:  out[b] = sum_j c_j part_j (ascending j),
// att_weight[b, t] *= c_(t / 32).  One block per sample.
// Blocks behind the B samples': gathers that only wait for the step's inputs (tail_roles.h, GatherJobs: DIN's target rows at
// the shipped batch size — one launch less on a launch-bound step).
__global__ __launch_bounds__(kBlock) void din_combine_kernel(int64_t B, int T, int NTL, int E,
                                                             const float* __restrict__ part, float* __restrict__ out,
                                                             float* __restrict__ att_weight, GatherJobs riders,
                                                             int32_t* __restrict__ rider_status) {
  if ((int64_t)blockIdx.x >= B) {
    gather_role((int)((int64_t)blockIdx.x - B), threadIdx.x, riders, rider_status);
    return;
  }
  __shared__ float cj[64];
  const int64_t b = blockIdx.x;
  const float* pb = part + b * NTL * (E + 2);
  if (threadIdx.x == 0) {
    float M = -INFINITY;
    for (int j = 0; j < NTL; ++j) M = fmaxf(M, pb[j * (E + 2) + E]);
    float L = 0.f;
    for (int j = 0; j < NTL; ++j) L += pb[j * (E + 2) + E + 1] * expf(pb[j * (E + 2) + E] - M);
    for (int j = 0; j < NTL; ++j) cj[j] = pb[j * (E + 2) + E + 1] * expf(pb[j * (E + 2) + E] - M) / L;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < E; e += kBlock) {
    float o = 0.f;
    for (int j = 0; j < NTL; ++j) o += cj[j] * pb[j * (E + 2) + e];
    out[b * E + e] = o;
  }
  if (att_weight)
    for (int t = threadIdx.x; t < T; t += kBlock) att_weight[b * T + t] *= cj[t / kDinTP];
}


// ------------------------------------------------------------------------------------------ backward
// Gradient of the attention-pool w.r.t. the gathered rows: dh [B,T,E] (history item|cat), dq [B,T,E]
// (target seq item|cat) — the per-position values whose row-wise merge is the embedding gradient.
// Same tiling as the forward; the hidden activations are recomputed per tile instead of stored.
//   pass 1 over the history: dp_t = dout . h_t,  sdp = sum_t p_t dp_t        (softmax backward needs it)
//   pass 2 per tile: a1, a2 (recomputed) -> dl_t = p_t (dp_t - sdp) E^-0.5 -> dz2 = dl w3 * a2(1-a2)
//                    -> dz1 = (dz2 W2^T) * a1(1-a1) -> dx = dz1 W1^T (W1 passed transposed: coalesced)
//                    dh = p_t dout + dx_a + dx_c + dx_d*q ;  dq = dx_b - dx_c + dx_d*h
// The attention MLP's own weight gradients are not produced: in the dygraph mode tools/trainer.py runs,
// those layers are not registered parameters and stay frozen.
struct DinBwdArgs {
  DinArgs f;
  const float* w1t;        // [H1][4E]  (att_w1 transposed)
  const float* att_weight; // [B,T] softmax weights saved by the forward
  const float* dout;       // [B,E]
  const float* out_saved;  // [B,E]    forward output        } both non-null: the compile-time-shaped kernel runs
  const float* act1;       // [B,T,H1] forward layer-1 acts  } on saved activations instead of recomputing them
  float *dh, *dq;          // [B,T,E]
  unsigned int* ticket;    // zeroed counter in the caller's workspace (or null): blocks draw whole samples from it
};

__global__ __launch_bounds__(kBlock) void din_attention_bwd_kernel(DinBwdArgs g) {
  const DinArgs& a = g.f;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int E = a.Ei + a.Ec, H1 = a.H1, H2 = a.H2, T = a.T;
  float* hs = smem;                    // [TP][E]
  float* qs = hs + kDinTP * E;         // [TP][E]
  float* a1 = qs + kDinTP * E;         // [TP][H1]   activations, then dz1 in place
  float* a2 = a1 + kDinTP * H1;        // [TP][H2]   activations, then dz2 in place
  float* w2s = a2 + kDinTP * H2;       // [H1][H2]
  float* w3s = w2s + H1 * H2;          // [H2]
  float* b1s = w3s + H2;               // [H1]
  float* b2s = b1s + H1;               // [H2]
  float* dls = b2s + H2;               // [TP] dl of the tile
  float* red = dls + kDinTP;           // [kBlock/64] block reduction scratch
  float* dps = red + kBlock / kWave;   // [T] dp_t of the whole history
  const int tid = threadIdx.x;
  for (int i = tid; i < H1 * H2; i += kBlock) w2s[i] = a.w2[i];
  for (int i = tid; i < H2; i += kBlock) { w3s[i] = a.w3[i]; b2s[i] = a.b2[i]; }
  for (int i = tid; i < H1; i += kBlock) b1s[i] = a.b1[i];
  const float scale = 1.f / sqrtf((float)E);
  const int j1 = tid % H1, pg1 = tid / H1, PG1 = kBlock / H1;
  const int NP1 = (kDinTP + PG1 - 1) / PG1;
  const bool on1 = pg1 < PG1;
  const int j2 = tid % H2, pg2 = tid / H2, PG2 = kBlock / H2;
  const int NP2 = (kDinTP + PG2 - 1) / PG2;
  const bool on2 = pg2 < PG2;
  const int e4 = E / 4;
  // dx mapping: thread (kk, pgx) owns embedding dim kk for positions pgx, pgx+PGX, ...
  const int kkx = tid % E, pgx = tid / E, PGX = kBlock / E;
  const int NPX = (kDinTP + PGX - 1) / PGX;
  const bool onx = pgx < PGX;
  const bool use_mfma = (E % 16 == 0) && (H1 % 16 == 0) && H1 <= 16 * kDinNT1;
  const bool use_mfma2 = H1 % 16 == 0;
  __syncthreads();

  auto gather_tile = [&](int64_t b, int t0, bool with_q) {
    for (int v = tid; v < kDinTP * e4; v += kBlock) {
      const int p = v / e4, c4 = (v % e4) * 4;
      const int t = t0 + p;
      float4 hv = make_float4(0.f, 0.f, 0.f, 0.f), qv = hv;
      if (t < T) {
        const bool item = c4 < a.Ei;
        const int64_t hid = item ? a.hist_item[b * T + t] : a.hist_cat[b * T + t];
        const int64_t nrow = item ? a.n_item : a.n_cat;
        const int ld = item ? a.ld_item : a.ld_cat;
        const int c = item ? c4 : c4 - a.Ei;
        const float* wh = item ? a.w_hist_item : a.w_hist_cat;
        if (hid >= 0 && hid < nrow) hv = *reinterpret_cast<const float4*>(wh + hid * ld + c);
        if (with_q) {
          const int64_t qid = item ? a.tgt_item[b * T + t] : a.tgt_cat[b * T + t];
          const float* wq = item ? a.w_tgt_item : a.w_tgt_cat;
          if (qid >= 0 && qid < nrow) qv = *reinterpret_cast<const float4*>(wq + qid * ld + c);
        }
      }
      *reinterpret_cast<float4*>(hs + p * E + c4) = hv;
      if (with_q) *reinterpret_cast<float4*>(qs + p * E + c4) = qv;
    }
  };

  for (int64_t b = blockIdx.x; b < a.B; b += gridDim.x) {
    const float dout_k = (tid < E) ? g.dout[b * E + tid] : 0.f;
    // ---- pass 1: dp_t and sdp = sum_t p_t dp_t
    float sdp_part = 0.f;
    for (int t0 = 0; t0 < T; t0 += kDinTP) {
      gather_tile(b, t0, false);
      __syncthreads();
      if (tid < kDinTP && t0 + tid < T) {
        float dp = 0.f;
        for (int k = 0; k < E; ++k) dp += g.dout[b * E + k] * hs[tid * E + k];
        dps[t0 + tid] = dp;
        sdp_part += g.att_weight[b * T + t0 + tid] * dp;
      }
      __syncthreads();
    }
    // block reduction of sdp_part (only threads < TP hold something): fixed order
    {
      float v = sdp_part;
#pragma unroll
      for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
      if (tid % kWave == 0) red[tid / kWave] = v;
      __syncthreads();
    }
    float sdp = 0.f;
    for (int w = 0; w < kBlock / kWave; ++w) sdp += red[w];
    __syncthreads();

    // ---- pass 2
    for (int t0 = 0; t0 < T; t0 += kDinTP) {
      gather_tile(b, t0, true);
      __syncthreads();
      if (use_mfma) {   // recompute layer 1 on the matrix cores (same routine as the forward)
        din_layer1_mfma(hs, qs, E, a.w1, b1s, a1, E, H1);
      } else if (on1) {   // recompute layer 1
        float s1[kDinNP1];
#pragma unroll
        for (int i = 0; i < kDinNP1; ++i) s1[i] = 0.f;
        const float* w = a.w1 + j1;
        for (int kk = 0; kk < E; ++kk) {
          const float wa = w[(int64_t)kk * H1], wb = w[(int64_t)(E + kk) * H1];
          const float wc = w[(int64_t)(2 * E + kk) * H1], wd = w[(int64_t)(3 * E + kk) * H1];
#pragma unroll
          for (int i = 0; i < kDinNP1; ++i) {
            const int p = pg1 + i * PG1;
            if (i < NP1 && p < kDinTP) {
              const float hv = hs[p * E + kk], qv = qs[p * E + kk];
              s1[i] += hv * wa + qv * wb + (hv - qv) * wc + (hv * qv) * wd;
            }
          }
        }
#pragma unroll
        for (int i = 0; i < kDinNP1; ++i) {
          const int p = pg1 + i * PG1;
          if (i < NP1 && p < kDinTP) a1[p * H1 + j1] = sigmoidf_(s1[i] + b1s[j1]);
        }
      }
      __syncthreads();
      if (use_mfma2) {   // recompute layer 2
        din_layer2_mfma(a1, w2s, b2s, a2, H1, H2);
      } else if (on2) {
        float s2[kDinNP2];
#pragma unroll
        for (int i = 0; i < kDinNP2; ++i) s2[i] = 0.f;
        for (int k = 0; k < H1; ++k) {
          const float wv = w2s[k * H2 + j2];
#pragma unroll
          for (int i = 0; i < kDinNP2; ++i) {
            const int p = pg2 + i * PG2;
            if (i < NP2 && p < kDinTP) s2[i] += a1[p * H1 + k] * wv;
          }
        }
#pragma unroll
        for (int i = 0; i < kDinNP2; ++i) {
          const int p = pg2 + i * PG2;
          if (i < NP2 && p < kDinTP) a2[p * H2 + j2] = sigmoidf_(s2[i] + b2s[j2]);
        }
      }
      if (tid < kDinTP) {   // dl_t (softmax backward + scale)
        const int t = t0 + tid;
        dls[tid] = (t < T) ? g.att_weight[b * T + t] * (dps[t] - sdp) * scale : 0.f;
      }
      __syncthreads();
      if (on2) {   // dz2 = dl * w3 * a2 (1 - a2), in place
#pragma unroll
        for (int i = 0; i < kDinNP2; ++i) {
          const int p = pg2 + i * PG2;
          if (i < NP2 && p < kDinTP) {
            const float av = a2[p * H2 + j2];
            a2[p * H2 + j2] = dls[p] * w3s[j2] * av * (1.f - av);
          }
        }
      }
      __syncthreads();
      if (on1) {   // dz1 = (dz2 W2^T) * a1 (1 - a1), in place (each thread rewrites only its own slots)
        float d1[kDinNP1];
#pragma unroll
        for (int i = 0; i < kDinNP1; ++i) d1[i] = 0.f;
        for (int k = 0; k < H2; ++k) {
          const float wv = w2s[j1 * H2 + k];
#pragma unroll
          for (int i = 0; i < kDinNP1; ++i) {
            const int p = pg1 + i * PG1;
            if (i < NP1 && p < kDinTP) d1[i] += a2[p * H2 + k] * wv;
          }
        }
#pragma unroll
        for (int i = 0; i < kDinNP1; ++i) {
          const int p = pg1 + i * PG1;
          if (i < NP1 && p < kDinTP) {
            const float av = a1[p * H1 + j1];
            a1[p * H1 + j1] = d1[i] * av * (1.f - av);
          }
        }
      }
      __syncthreads();
      if (onx) {   // dx = dz1 W1^T, folded straight into dh / dq
        float xa[kDinNP1], xb[kDinNP1], xc[kDinNP1], xd[kDinNP1];
#pragma unroll
        for (int i = 0; i < kDinNP1; ++i) xa[i] = xb[i] = xc[i] = xd[i] = 0.f;
        for (int j = 0; j < H1; ++j) {
          const float* wt = g.w1t + (int64_t)j * 4 * E + kkx;
          const float wa = wt[0], wb = wt[E], wc = wt[2 * E], wd = wt[3 * E];
#pragma unroll
          for (int i = 0; i < kDinNP1; ++i) {
            const int p = pgx + i * PGX;
            if (i < NPX && p < kDinTP) {
              const float dz = a1[p * H1 + j];
              xa[i] += dz * wa; xb[i] += dz * wb; xc[i] += dz * wc; xd[i] += dz * wd;
            }
          }
        }
#pragma unroll
        for (int i = 0; i < kDinNP1; ++i) {
          const int p = pgx + i * PGX;
          const int t = t0 + p;
          if (i < NPX && p < kDinTP && t < T) {
            const float hv = hs[p * E + kkx], qv = qs[p * E + kkx];
            const float pw = g.att_weight[b * T + t];
            g.dh[(b * T + t) * (int64_t)E + kkx] = pw * dout_k_of(g.dout, b, E, kkx) + xa[i] + xc[i] + xd[i] * qv;
            g.dq[(b * T + t) * (int64_t)E + kkx] = xb[i] - xc[i] + xd[i] * hv;
          }
        }
      }
      __syncthreads();
    }
    (void)dout_k;
  }
}

// ------------------------------------------------------------------------ backward, compile-time shapes
// Backward on the activations the forward saved (act1 [B,T,H1], out [B,E]) instead of recomputing layer 1:
//   * sdp = sum_t p_t (dout . h_t) = dout . out — the first pass over the history of the kernel above disappears;
//   * a2 is recomputed from a1 (one small MFMA GEMM), dz2 / dz1 are MFMA GEMMs against the LDS copy of W2;
//   * dx = dz1 W1^T — as large as layer 1 — runs on the matrix cores with W1^T fragments resident in registers.
//     Wave w owns columns [w*E/4, (w+1)*E/4) of ALL FOUR concat segments, so dx_a, dx_b, dx_c, dx_d of an (p, e)
//     sit in the same lane and   dh = p_t dout + dx_a + dx_c + dx_d*q,   dq = dx_b - dx_c + dx_d*h
//     are formed in registers and stored straight to HBM: no partial sums, no LDS round trip for the result.
// Saving a1 costs B*T*H1*4 bytes written once and read once (0.67 GB at B 4096, T 512: ~0.3 ms of HBM time)
// against 2.7 ms of recomputed layer-1 MFMA work.
template <int E, int H1, int H2>
struct DinBwdCt {
  static_assert(E % 64 == 0 && kBlock % E == 0 && H1 % 16 == 0, "unsupported compile-time shape");
  static constexpr int EP = E + 4, H1P = H1 + 4;
  static constexpr int H2C = ((H2 + 15) / 16) * 16, W2P = H2C + 4, Z2P = H2C + 4;
  static constexpr int NJB = H1 / 16, NCB = H2C / 16;
  static constexpr int EW = E / 4, NH = EW / 16;         // columns of each segment owned by a wave, in MFMA tiles
  static constexpr int E4 = E / 4, NIT = kDinTP * E4 / kBlock, PSTEP = kBlock / E4;
  static constexpr int kHs = 0;
  static constexpr int kQs = kHs + kDinTP * EP;
  static constexpr int kX = kQs + kDinTP * EP;           // a1 of the tile
  static constexpr int kY = kX + kDinTP * H1P;           // dz1
  static constexpr int kZ2 = kY + kDinTP * H1P;          // dz2
  static constexpr int kW2 = kZ2 + kDinTP * Z2P;
  static constexpr int kW3 = kW2 + H1 * W2P;
  static constexpr int kB2 = kW3 + H2C;
  static constexpr int kDl = kB2 + H2C;                  // [32] dl of the tile
  static constexpr int kPw = kDl + kDinTP;               // [2][32] softmax weights of this / the next tile
  static constexpr int kDout = kPw + 2 * kDinTP;         // [E] dout of the sample, [E] sdp
  static constexpr int kIds = (kDout + E + 8 + 1) & ~1;  // (douts[E+1]: skip flag, [E+2..E+3]: sample tickets, [E+4..E+7]: tail scan)  int64 [4][32]: ids of the NEXT tile (this tile's are spent)
  static constexpr int kEnd = kIds + 2 * 4 * kDinTP;
  static_assert(2 * sizeof(float) * kEnd <= 160 * 1024 || E > 128, "two blocks per CU no longer fit the LDS");
  static constexpr size_t lds_bytes = sizeof(float) * (size_t)kEnd;
};

template <int E, int H1, int H2>
__global__ __launch_bounds__(kBlock, 2) void din_attention_bwd_ct_kernel(DinBwdArgs gb) {
  using S = DinBwdCt<E, H1, H2>;
  const DinArgs& a = gb.f;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* hs = smem + S::kHs;
  float* qs = smem + S::kQs;
  float* X = smem + S::kX;
  float* Y = smem + S::kY;
  float* Z2 = smem + S::kZ2;
  float* w2s = smem + S::kW2;
  float* w3s = smem + S::kW3;
  float* b2s = smem + S::kB2;
  float* dls = smem + S::kDl;
  float* pws = smem + S::kPw;
  float* douts = smem + S::kDout;
  int64_t* idbuf = reinterpret_cast<int64_t*>(smem + S::kIds);
  const int tid = threadIdx.x, lane = tid % kWave;
  const int wv = __builtin_amdgcn_readfirstlane(tid / kWave);
  const int li = lane & 15, g = lane >> 4;
  const int T = a.T;
  for (int i = tid; i < H1 * S::W2P; i += kBlock) {
    const int k = i / S::W2P, c = i % S::W2P;
    w2s[i] = c < H2 ? a.w2[k * H2 + c] : 0.f;
  }
  for (int i = tid; i < S::H2C; i += kBlock) {
    w3s[i] = i < H2 ? a.w3[i] : 0.f;
    b2s[i] = i < H2 ? a.b2[i] : 0.f;
  }
  const float scale = 1.f / sqrtf((float)E);
  // W1^T as MFMA B fragments of dx[p][n] = sum_j dz1[p][j] W1[n][j]:
  //   bt[sg][nh][jb][s] = W1[sg*E + wv*EW + nh*16 + li][jb*16 + 4g + s]
  float bt[4][S::NH][S::NJB][4];
#pragma unroll
  for (int sg = 0; sg < 4; ++sg)
#pragma unroll
    for (int nh = 0; nh < S::NH; ++nh)
#pragma unroll
      for (int jb = 0; jb < S::NJB; ++jb) {
        const float4 w = *reinterpret_cast<const float4*>(
            a.w1 + (int64_t)(sg * E + wv * S::EW + nh * 16 + li) * H1 + jb * 16 + 4 * g);
        bt[sg][nh][jb][0] = w.x; bt[sg][nh][jb][1] = w.y; bt[sg][nh][jb][2] = w.z; bt[sg][nh][jb][3] = w.w;
      }

  const int c4 = (tid % S::E4) * 4, p0 = tid / S::E4;
  const bool item = c4 < a.Ei;
  const float* wh = item ? a.w_hist_item : a.w_hist_cat;
  const float* wq = item ? a.w_tgt_item : a.w_tgt_cat;
  const int64_t nrow = item ? a.n_item : a.n_cat;
  const int ld = item ? a.ld_item : a.ld_cat;
  const int col = item ? c4 : c4 - a.Ei;
  const int ih = item ? 0 : 1, iq = item ? 2 : 3;
  const int id_arr = tid >> 5, id_p = tid & 31;     // threads 0..127: ids; 128..159: softmax weights
  const int64_t* id_ptr = id_arr == 0 ? a.hist_item : id_arr == 1 ? a.hist_cat : id_arr == 2 ? a.tgt_item : a.tgt_cat;

  auto ids_issue = [&](int64_t b, int t0, int64_t& idv, float& pwv) {
    const int t = t0 + id_p;
    idv = 0; pwv = 0.f;
    if (t < T) {
      if (id_arr < 4) idv = id_ptr[b * T + t];
      else if (id_arr == 4) pwv = gb.att_weight[b * T + t];
    }
  };
  auto ids_store = [&](int buf, int64_t idv, float pwv) {
    if (id_arr < 4) idbuf[id_arr * kDinTP + id_p] = idv;
    else if (id_arr == 4) pws[buf * kDinTP + id_p] = pwv;
  };
  // gathers h, q rows and the saved a1 rows of a tile into LDS.  Branch-free (see the forward's rows_issue): every fetch
  // is issued unconditionally on a clamped address and zeroed on the way into LDS, so the thread's loads are in flight
  // together instead of one memory round trip per divergent branch.
  auto tile_load = [&](int buf, int64_t b, int t0) {
    f32x4_t ph[S::NIT], pq[S::NIT];
    unsigned ok = 0;
#pragma unroll
    for (int it = 0; it < S::NIT; ++it) {
      const int p = p0 + it * S::PSTEP;
      const bool inb = t0 + p < T;
      const int64_t hid = idbuf[ih * kDinTP + p], qid = idbuf[iq * kDinTP + p];
      const bool hok = hid >= 0 && hid < nrow, qok = qid >= 0 && qid < nrow;
      ph[it] = *reinterpret_cast<const f32x4_t*>(wh + (hok ? hid : 0) * ld + col);
      pq[it] = *reinterpret_cast<const f32x4_t*>(wq + (qok ? qid : 0) * ld + col);
      ok |= (inb && hok ? 1u : 0u) << it | (inb && qok ? 1u : 0u) << (8 + it);
    }
    constexpr int NV = kDinTP * (H1 / 4), NA = (NV + kBlock - 1) / kBlock;
    f32x4_t pa[NA];
#pragma unroll
    for (int it = 0; it < NA; ++it) {
      const int v = tid + it * kBlock;
      const int p = v / (H1 / 4), c = (v % (H1 / 4)) * 4;
      const bool live = v < NV && t0 + p < T;
      pa[it] = *reinterpret_cast<const f32x4_t*>(gb.act1 + (b * T + (live ? t0 + p : 0)) * H1 + (live ? c : 0));
      ok |= (live ? 1u : 0u) << (16 + it);
    }
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < S::NIT; ++it) {
      const int p = p0 + it * S::PSTEP;
      *reinterpret_cast<f32x4_t*>(hs + p * S::EP + c4) = (ok >> it & 1u) ? ph[it] : zero;
      *reinterpret_cast<f32x4_t*>(qs + p * S::EP + c4) = (ok >> (8 + it) & 1u) ? pq[it] : zero;
    }
#pragma unroll
    for (int it = 0; it < NA; ++it) {
      const int v = tid + it * kBlock;
      const int p = v / (H1 / 4), c = (v % (H1 / 4)) * 4;
      if (v < NV) *reinterpret_cast<f32x4_t*>(X + p * S::H1P + c) = (ok >> (16 + it) & 1u) ? pa[it] : zero;
    }
  };
  // dout of a sample and sdp = dout . out (softmax backward), by wave 0 and threads < E
  auto sample_load = [&](int64_t b) {
    if (tid < E) douts[tid] = gb.dout[b * E + tid];
    if (wv == 0) {
      float v = 0.f;
      for (int k = lane; k < E; k += kWave) v += gb.dout[b * E + k] * gb.out_saved[b * E + k];
#pragma unroll
      for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
      if (lane == 0) douts[E] = v;
    }
  };

  // Work items are (sample, 32-position tile) pairs — with the saved softmax weights and sdp = dout . out no tile of a
  // sample depends on another one — dealt to the blocks as contiguous ranges of the flattened list: a block still walks
  // the tiles of one sample after the other at large batches, and at the reference's batch size (32 samples x 5 tiles
  // on 256 CUs) every tile has a block of its own instead of 32 blocks walking five tiles each (68 -> ~20 us).
  const int NTL = (T + kDinTP - 1) / kDinTP;
  const int64_t W = a.B * NTL, per = (W + gridDim.x - 1) / gridDim.x;
  // With a ticket counter (gb.ticket: the _ws entry point at large batches) a block draws whole SAMPLES — skipped tiles
  // make a sample cost anything between one and T / 32 tiles — one sample ahead, the draw after that issued while the
  // current sample is worked on; otherwise contiguous ranges of the flattened list as before.
  const bool dyn = gb.ticket != nullptr;
  int* itk = reinterpret_cast<int*>(douts + E + 2);
  int64_t w = (int64_t)blockIdx.x * per;
  int64_t w_end = (w + per < W) ? w + per : W;
  int64_t next_sample = a.B;
  // The padded TAIL of a sample (weights exactly 0.0f behind the last valid position: the forward does not even walk it)
  // is not visited tile by tile — every skipped tile still cost the ids / weights round trip of the tile behind it, two
  // block barriers and its share of the loop (17 k cycles each, as much as a computed tile's non-MFMA part): a block
  // that owns a whole sample (ticket mode) scans the sample's weight row once, zero-fills the gradient rows behind the
  // last non-zero weight in one streaming pass and walks only the tiles in front of it.  -> tiles to walk (>= 1)
  auto tail_cut = [&](int64_t bs) -> int {          // block-uniform; two barriers
    int last = 0;
    for (int t = tid; t < T; t += kBlock)
      if (gb.att_weight[bs * T + t] != 0.f) last = t + 1;             // ascending t: the last hit wins
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) last = max(last, __shfl_xor(last, o, kWave));
    __syncthreads();
    if (lane == 0) itk[2 + wv] = last;
    __syncthreads();
    last = max(max(itk[2], itk[3]), max(itk[4], itk[5]));
    const int nt = last > 0 ? (last + kDinTP - 1) / kDinTP : 1;
    const int tz = nt * kDinTP;
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    for (int v = tid; v < (T - tz) * S::E4; v += kBlock) {
      const int64_t o = (bs * T + tz + v / S::E4) * (int64_t)E + (v % S::E4) * 4;
      __builtin_nontemporal_store(zero, reinterpret_cast<f32x4_t*>(gb.dh + o));
      __builtin_nontemporal_store(zero, reinterpret_cast<f32x4_t*>(gb.dq + o));
    }
    return nt;
  };
  if (dyn) {
    if (tid == 0) {
      itk[0] = (int)atomicAdd(gb.ticket, 1u);
      itk[1] = (int)atomicAdd(gb.ticket, 1u);
    }
    __syncthreads();
    w = (int64_t)itk[0] * NTL;
    next_sample = itk[1];
    __syncthreads();
    if (w >= W) return;
    w_end = w + tail_cut(w / NTL);
  }
  if (w >= W || w >= w_end) return;
  int64_t b = w / NTL;
  int t0 = (int)(w % NTL) * kDinTP, buf = 0;
  {
    int64_t idv; float pwv;
    ids_issue(b, t0, idv, pwv);
    ids_store(0, idv, pwv);
    sample_load(b);
  }
  // A tile whose 32 saved softmax weights are all 0.0f (padding behind the valid positions: the forward writes exact
  // zeros there, and skips such tiles itself) has dl = p (..) = 0, hence dz2 = dz1 = dx = 0 and dh = p dout = 0: its
  // gradient rows are written as zeros without gathering anything (block-uniform flag from a ballot over the weights).
  // (a flag in the dynamic LDS block, not __syncthreads_and: its static LDS word would push the kernel past the
  //  160 KB dynamic-LDS attribute the launcher sets)
  auto tile_is_zero = [&](int bufi) -> bool {      // block-uniform; two barriers
    __syncthreads();
    if (wv == 0) {
      const bool z = lane >= kDinTP || pws[bufi * kDinTP + lane] == 0.f;
      const bool all = __builtin_amdgcn_ballot_w64(z) == ~0ull;
      if (lane == 0) douts[E + 1] = all ? 1.f : 0.f;
    }
    __syncthreads();
    return douts[E + 1] != 0.f;
  };
  bool skip = tile_is_zero(0);
  if (!skip) tile_load(0, b, t0);
  __syncthreads();

  DIN_PHB_DECL;
  while (true) {
    DIN_PHB(9);
    int64_t nw = w + 1;
    const bool chunk_end = nw >= w_end;
    if (chunk_end && dyn) nw = next_sample * NTL;          // the sample drawn a sample ago
    const bool has_next = dyn ? (!chunk_end || next_sample < a.B) : !chunk_end;
    const int64_t nb = nw / NTL;
    const int nt0 = (int)(nw % NTL) * kDinTP;
    int tk = 0;
    if (dyn && chunk_end && has_next && tid == 0) tk = (int)atomicAdd(gb.ticket, 1u);   // the sample after the next one
    int64_t idv = 0;
    float pwv = 0.f;
    if (has_next) ids_issue(nb, nt0, idv, pwv);
    if (skip) {
      for (int v = tid; v < kDinTP * S::E4; v += kBlock) {
        const int p = v / S::E4, c = (v % S::E4) * 4, t = t0 + p;
        if (t < T) {
          *reinterpret_cast<float4*>(gb.dh + (b * T + t) * (int64_t)E + c) = make_float4(0.f, 0.f, 0.f, 0.f);
          *reinterpret_cast<float4*>(gb.dq + (b * T + t) * (int64_t)E + c) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      if (!has_next) break;
      ids_store(buf ^ 1, idv, pwv);
      if (dyn && chunk_end && tid == 0) itk[0] = tk;
      skip = tile_is_zero(buf ^ 1);
      if (dyn && chunk_end) { next_sample = itk[0]; w_end = nw + tail_cut(nb); }
      if (nb != b) sample_load(nb);
      if (!skip) tile_load(buf ^ 1, nb, nt0);
      __syncthreads();
      w = nw; b = nb; t0 = nt0; buf ^= 1;
      continue;
    }
    // ---- dl_t = p_t (dout . h_t - sdp) E^-0.5 : 8 threads per position
    {
      const int p = tid >> 3, sub = tid & 7;
      float dp = 0.f;
#pragma unroll
      for (int i = 0; i < E / 32; ++i) {
        const int k = (i * 8 + sub) * 4;
        const float4 hv = *reinterpret_cast<const float4*>(hs + p * S::EP + k);
        const float4 dv = *reinterpret_cast<const float4*>(douts + k);
        dp += hv.x * dv.x + hv.y * dv.y + hv.z * dv.z + hv.w * dv.w;
      }
#pragma unroll
      for (int o = 4; o > 0; o >>= 1) dp += __shfl_xor(dp, o, 8);
      if (sub == 0) dls[p] = pws[buf * kDinTP + p] * (dp - douts[E]) * scale;   // padded positions: p_t = 0
    }
    __syncthreads();
    DIN_PHB(0);
    // ---- a2 = sigmoid(a1 W2 + b2) recomputed on the matrix cores; dz2 = dl w3 a2 (1 - a2) in the epilogue
    for (int t = wv; t < 2 * S::NCB; t += kBlock / kWave) {
      const int m = t & 1, n = t >> 1;
      const int c = n * 16 + li;
      f32x4_t z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < S::NJB; ++kb) {
        const int k0 = kb * 16 + 4 * g;
        const float4 av = *reinterpret_cast<const float4*>(X + (m * 16 + li) * S::H1P + k0);
        const float a4[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
        for (int s = 0; s < 4; ++s)
          z = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s], w2s[(k0 + s) * S::W2P + c], z, 0, 0, 0);
      }
      const float bias = b2s[c], w3c = w3s[c];      // columns >= H2: w3 = 0 -> dz2 = 0
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int p = m * 16 + g * 4 + r;
        const float a2v = sigmoidf_(z[r] + bias);
        Z2[p * S::Z2P + c] = dls[p] * w3c * a2v * (1.f - a2v);
      }
    }
    __syncthreads();
    DIN_PHB(1);
    // ---- dz1 = (dz2 W2^T) a1 (1 - a1)
    for (int t = wv; t < 2 * S::NJB; t += kBlock / kWave) {
      const int m = t & 1, n = t >> 1;
      f32x4_t z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int cb = 0; cb < S::NCB; ++cb) {
        const int k0 = cb * 16 + 4 * g;
        const float4 av = *reinterpret_cast<const float4*>(Z2 + (m * 16 + li) * S::Z2P + k0);
        const float4 bv = *reinterpret_cast<const float4*>(w2s + (n * 16 + li) * S::W2P + k0);
        z = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, z, 0, 0, 0);
        z = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, z, 0, 0, 0);
        z = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, z, 0, 0, 0);
        z = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, z, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int p = m * 16 + g * 4 + r, j = n * 16 + li;
        const float a1v = X[p * S::H1P + j];
        Y[p * S::H1P + j] = z[r] * a1v * (1.f - a1v);
      }
    }
    if (has_next) ids_store(buf ^ 1, idv, pwv);
    __syncthreads();
    DIN_PHB(2);
    // ---- dx = dz1 W1^T for this wave's columns of the four segments, folded into dh / dq (16 positions a pass)
#pragma unroll 1
    for (int m = 0; m < 2; ++m) {
      f32x4_t acc[4][S::NH];
#pragma unroll
      for (int sg = 0; sg < 4; ++sg)
#pragma unroll
        for (int nh = 0; nh < S::NH; ++nh) acc[sg][nh] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int jb = 0; jb < S::NJB; ++jb) {
        const float4 av = *reinterpret_cast<const float4*>(Y + (m * 16 + li) * S::H1P + jb * 16 + 4 * g);
        const float a4[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int sg = 0; sg < 4; ++sg)
#pragma unroll
            for (int nh = 0; nh < S::NH; ++nh)
              acc[sg][nh] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[s], bt[sg][nh][jb][s], acc[sg][nh], 0, 0, 0);
      }
#pragma unroll
      for (int nh = 0; nh < S::NH; ++nh) {
        const int e = wv * S::EW + nh * 16 + li;
        const float de = douts[e];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int p = m * 16 + g * 4 + r, t = t0 + p;
          if (t < T) {
            const float hv = hs[p * S::EP + e], qv = qs[p * S::EP + e];
            const float xa = acc[0][nh][r], xb = acc[1][nh][r], xc = acc[2][nh][r], xd = acc[3][nh][r];
            gb.dh[(b * T + t) * (int64_t)E + e] = pws[buf * kDinTP + p] * de + xa + xc + xd * qv;
            gb.dq[(b * T + t) * (int64_t)E + e] = xb - xc + xd * hv;
          }
        }
      }
    }
    DIN_PHB(3);
    if (!has_next) break;
    if (dyn && chunk_end && tid == 0) itk[0] = tk;
    // hs / qs / X / douts are rewritten for the next tile (barrier); is the next tile all zero weights?
    skip = tile_is_zero(buf ^ 1);
    DIN_PHB(4);
    if (dyn && chunk_end) { next_sample = itk[0]; w_end = nw + tail_cut(nb); }
    if (nb != b) sample_load(nb);
    if (!skip) tile_load(buf ^ 1, nb, nt0);
    DIN_PHB(5);
    __syncthreads();
    w = nw; b = nb; t0 = nt0; buf ^= 1;
    DIN_PHB(6);
  }
  DIN_PHB_FLUSH;
}

}  // namespace rec

using namespace rec;

#ifdef REC_DIN_PHASE_TIMING
extern "C" __attribute__((visibility("default"))) int rec_din_debug_phases(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(din_phase_dbg), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(din_phase_dbg), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif

// 1 when the forward for this shape writes `act1` (and the backward can consume it), else 0
extern "C" int rec_din_saves_act1(const rec_din_desc* d) {
  if (!d) return 0;
  using Ct = DinCt<128, 80, 40>;
  return d->item_dim + d->cat_dim == 128 && d->hidden1 == 80 && d->hidden2 == 40 && d->max_len > 0 &&
         Ct::lds_bytes(d->max_len) <= kDinCtLdsMax && getenv("REC_DIN_FWD_GENERIC") == nullptr;
}

// workspace of the tile-split forward: B * ceil(T / 32) pieces of (E + 2) floats; 0 = the shape never splits
extern "C" int rec_din_attention_pool_fwd_workspace_bytes(const rec_din_desc* d, size_t* bytes) {
  REC_REQUIRE(d && bytes, REC_EINVAL, "null argument");
  const int64_t ntl = (d->max_len + kDinTP - 1) / kDinTP;
  const int64_t tiles = d->batch * ntl;
  *bytes = (ntl > 1 && ntl <= 64 && tiles <= 4096) ? align_up((size_t)tiles * (d->item_dim + d->cat_dim + 2) * sizeof(float), 256)
                                                  : 0;
  // larger batches of the compile-time-shaped kernel: one 32-bit ticket counter (dynamic sample distribution)
  if (*bytes == 0 && rec_din_saves_act1(d) == 1 && ntl > 1) *bytes = 256;
  return REC_OK;
}

static thread_local rec::GatherJobs t_combine_rider_jobs;
static thread_local const rec::GatherJobs* t_combine_rider = nullptr;
static thread_local int32_t* t_combine_rider_status = nullptr;
static thread_local bool t_combine_rider_taken = false;
void rec::din_combine_rider_set(const GatherJobs* jobs, int32_t* status) {
  t_combine_rider = nullptr;
  if (jobs && status && jobs->count > 0) {
    t_combine_rider_jobs = *jobs;
    t_combine_rider = &t_combine_rider_jobs;
  }
  t_combine_rider_status = status;
  t_combine_rider_taken = false;
}
bool rec::din_combine_rider_take() {
  const bool taken = t_combine_rider_taken;
  t_combine_rider = nullptr;
  t_combine_rider_taken = false;
  return taken;
}

extern "C" int rec_din_attention_pool_fwd_ws(const rec_din_desc* d, const int64_t* hist_item,
                                          const int64_t* hist_cat, const int64_t* tgt_item_seq,
                                          const int64_t* tgt_cat_seq, const int64_t* mask,
                                          const float* w_hist_item, const float* w_hist_cat,
                                          const float* w_tgt_item_seq, const float* w_tgt_cat_seq,
                                          const float* att_w1, const float* att_b1,
                                          const float* att_w2, const float* att_b2,
                                          const float* att_w3, const float* att_b3, float* out,
                                          float* att_weight, float* act1, int32_t* status, void* workspace,
                                          size_t workspace_bytes, void* stream) {
  REC_REQUIRE(d, REC_EINVAL, "desc is NULL");
  const int E = d->item_dim + d->cat_dim;
  REC_REQUIRE(d->batch >= 0 && d->max_len > 0 && d->item_dim > 0 && d->cat_dim > 0 && d->hidden1 > 0 &&
                  d->hidden2 > 0 && d->item_rows > 0 && d->cat_rows > 0, REC_EINVAL, "bad sizes");
  REC_REQUIRE(d->item_dim % 4 == 0 && d->cat_dim % 4 == 0 && E <= kBlock, REC_ESHAPE,
              "item_dim/cat_dim must be multiples of 4 with item_dim+cat_dim <= %d", kBlock);
  REC_REQUIRE(d->hidden1 <= kBlock / 2 && d->hidden2 <= kBlock / 2, REC_ESHAPE, "hidden sizes must be <= %d",
              kBlock / 2);
  REC_REQUIRE((kDinTP + kBlock / d->hidden1 - 1) / (kBlock / d->hidden1) <= kDinNP1 &&
                  (kDinTP + kBlock / d->hidden2 - 1) / (kBlock / d->hidden2) <= kDinNP2, REC_ESHAPE,
              "hidden sizes too large for the tile");
  REC_REQUIRE(d->item_stride >= d->item_dim && d->cat_stride >= d->cat_dim && d->item_stride % 4 == 0 &&
                  d->cat_stride % 4 == 0, REC_EINVAL, "table strides must be multiples of 4 and >= dims");
  if (d->batch == 0) return REC_OK;
  REC_REQUIRE(hist_item && hist_cat && tgt_item_seq && tgt_cat_seq && mask && w_hist_item && w_hist_cat &&
                  w_tgt_item_seq && w_tgt_cat_seq && att_w1 && att_b1 && att_w2 && att_b2 && att_w3 &&
                  att_b3 && out && status, REC_EINVAL, "null pointer argument");
  const int H1 = d->hidden1, H2 = d->hidden2;
  size_t shmem = sizeof(float) * ((size_t)2 * kDinTP * (E + 4) + (size_t)kDinTP * (H1 + H2) +
                                  (size_t)H1 * H2 + 2 * H2 + H1 + (size_t)d->max_len);
  // the reference net's own shape (E 128, attention MLP 80-40-1) runs the compile-time-shaped kernel
  using Ct = DinCt<128, 80, 40>;
  static const bool force_generic = getenv("REC_DIN_FWD_GENERIC") != nullptr;
  const bool ct = !force_generic && E == 128 && H1 == 80 && H2 == 40 && Ct::lds_bytes(d->max_len) <= kDinCtLdsMax;
  if (ct) shmem = Ct::lds_bytes(d->max_len);
  else REC_REQUIRE(shmem <= 64 * 1024, REC_ESHAPE, "history too long for the LDS logit buffer (%zu B)", shmem);
  DinArgs a;
  a.B = d->batch; a.T = d->max_len; a.Ei = d->item_dim; a.Ec = d->cat_dim; a.H1 = H1; a.H2 = H2;
  a.n_item = d->item_rows; a.n_cat = d->cat_rows; a.ld_item = d->item_stride; a.ld_cat = d->cat_stride;
  a.hist_item = hist_item; a.hist_cat = hist_cat; a.tgt_item = tgt_item_seq; a.tgt_cat = tgt_cat_seq;
  a.mask = mask; a.w_hist_item = w_hist_item; a.w_hist_cat = w_hist_cat; a.w_tgt_item = w_tgt_item_seq;
  a.w_tgt_cat = w_tgt_cat_seq; a.w1 = att_w1; a.b1 = att_b1; a.w2 = att_w2; a.b2 = att_b2; a.w3 = att_w3;
  a.b3 = att_b3; a.out = out; a.att_weight = att_weight; a.status = status;
  a.act1 = ct ? act1 : nullptr;       // only the compile-time-shaped pair saves / consumes layer-1 activations
  a.part = nullptr;
  a.ticket = nullptr;
  if (ct) {
    // REC_DIN_FWD_VARIANT: measurement knob.  Measured at B 4096, T 512 (profiles/r02_din_variants.txt):
    //   nopf2 (default; rows fetched at the end of the tile, 2 blocks/CU, 36 spilled VGPRs)  2.74 ms  68.0 TF
    //   pf2   (rows prefetched into registers during layers 2/3, 76 spilled VGPRs)            3.25 ms  57.3 TF
    //   pf1   (prefetch, 345 VGPRs, 1 block/CU, no spills)                                    3.36 ms  55.4 TF
    // the runtime-shaped kernel: 6.81 ms, 27.3 TF.
    static const int variant = [] {
      const char* v = getenv("REC_DIN_FWD_VARIANT");
      return !v ? 1 : !strcmp(v, "pf2") ? 0 : !strcmp(v, "pf1") ? 2 : 1;
    }();
    void (*kern)(DinArgs) = variant == 1   ? din_attention_fwd_ct_kernel<128, 80, 40, false, 2>
                            : variant == 2 ? din_attention_fwd_ct_kernel<128, 80, 40, true, 1>
                                           : din_attention_fwd_ct_kernel<128, 80, 40, true, 2>;
    static bool attr_set[3] = {false, false, false};
    if (!attr_set[variant]) {
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDinCtLdsMax);
      attr_set[variant] = true;
    }
    int64_t grid = resident_blocks(kern, kBlock, shmem);
    // few samples (the reference's batch size): one block per 32-position tile + a combine launch instead of one
    // block per sample walking its tiles (REC_DIN_TILE_SPLIT=0: off)
    const int ntl = (d->max_len + kDinTP - 1) / kDinTP;
    const int64_t tiles = d->batch * ntl;
    size_t need = 0;
    rec_din_attention_pool_fwd_workspace_bytes(d, &need);
    static const bool split_env = [] { const char* v = getenv("REC_DIN_TILE_SPLIT"); return !(v && *v == '0'); }();
    if (split_env && need > 0 && workspace && workspace_bytes >= need && tiles <= grid) {
      a.part = (float*)workspace;
      hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(kBlock), shmem, (hipStream_t)stream, a);
      GatherJobs riders;
      riders.count = riders.blocks = 0;
      int32_t* rider_status = nullptr;
      if (t_combine_rider) {                              // (din_combine_rider_set: this thread's next combine launch)
        riders = *t_combine_rider;
        rider_status = t_combine_rider_status;
        t_combine_rider = nullptr;
        t_combine_rider_taken = true;
      }
      hipLaunchKernelGGL(din_combine_kernel, dim3((unsigned)(d->batch + riders.blocks)), dim3(kBlock), 0,
                         (hipStream_t)stream, d->batch, d->max_len, ntl, E, (const float*)workspace, out, att_weight, riders,
                         rider_status);
      return check_launch("rec_din_attention_pool_fwd (tile split)");
    }
    if (grid > d->batch) grid = d->batch;
    static const bool ticket_env = [] { const char* v = getenv("REC_DIN_TICKETS"); return !(v && *v == '0'); }();
    if (ticket_env && workspace && workspace_bytes >= sizeof(unsigned int) && ntl > 1 && d->batch > grid &&
        d->batch < (1ll << 31) - 4 * grid) {
      a.ticket = (unsigned int*)workspace;
      REC_REQUIRE(hipMemsetAsync(workspace, 0, sizeof(unsigned int), (hipStream_t)stream) == hipSuccess, REC_EHIP,
                  "hipMemsetAsync failed");
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kBlock), shmem, (hipStream_t)stream, a);
    return check_launch("rec_din_attention_pool_fwd");
  }
  int64_t grid = resident_blocks(din_attention_fwd_kernel, kBlock, shmem);
  if (grid > d->batch) grid = d->batch;
  hipLaunchKernelGGL(din_attention_fwd_kernel, dim3((unsigned)grid), dim3(kBlock), shmem,
                     (hipStream_t)stream, a);
  return check_launch("rec_din_attention_pool_fwd");
}

extern "C" int rec_din_attention_pool_fwd(const rec_din_desc* d, const int64_t* hist_item,
                                          const int64_t* hist_cat, const int64_t* tgt_item_seq,
                                          const int64_t* tgt_cat_seq, const int64_t* mask,
                                          const float* w_hist_item, const float* w_hist_cat,
                                          const float* w_tgt_item_seq, const float* w_tgt_cat_seq,
                                          const float* att_w1, const float* att_b1,
                                          const float* att_w2, const float* att_b2,
                                          const float* att_w3, const float* att_b3, float* out,
                                          float* att_weight, float* act1, int32_t* status, void* stream) {
  return rec_din_attention_pool_fwd_ws(d, hist_item, hist_cat, tgt_item_seq, tgt_cat_seq, mask, w_hist_item, w_hist_cat,
                                       w_tgt_item_seq, w_tgt_cat_seq, att_w1, att_b1, att_w2, att_b2, att_w3, att_b3, out,
                                       att_weight, act1, status, nullptr, 0, stream);
}

extern "C" int rec_din_attention_pool_bwd(const rec_din_desc* d, const int64_t* hist_item,
                                          const int64_t* hist_cat, const int64_t* tgt_item_seq,
                                          const int64_t* tgt_cat_seq, const float* w_hist_item,
                                          const float* w_hist_cat, const float* w_tgt_item_seq,
                                          const float* w_tgt_cat_seq, const float* att_w1,
                                          const float* att_w1_t, const float* att_b1,
                                          const float* att_w2, const float* att_b2,
                                          const float* att_w3, const float* att_weight,
                                          const float* out_saved, const float* act1_saved,
                                          const float* d_out, float* d_hist, float* d_tgt_seq,
                                          void* stream) {
  return rec_din_attention_pool_bwd_ws(d, hist_item, hist_cat, tgt_item_seq, tgt_cat_seq, w_hist_item, w_hist_cat,
                                       w_tgt_item_seq, w_tgt_cat_seq, att_w1, att_w1_t, att_b1, att_w2, att_b2, att_w3,
                                       att_weight, out_saved, act1_saved, d_out, d_hist, d_tgt_seq, nullptr, 0, stream);
}

// one 32-bit ticket counter for the compile-time-shaped kernel at batches larger than the resident grid; else nothing
extern "C" int rec_din_attention_pool_bwd_workspace_bytes(const rec_din_desc* d, size_t* bytes) {
  REC_REQUIRE(d && bytes, REC_EINVAL, "null argument");
  *bytes = (rec_din_saves_act1(d) == 1 && d->max_len > kDinTP) ? 256 : 0;
  return REC_OK;
}

extern "C" int rec_din_attention_pool_bwd_ws(const rec_din_desc* d, const int64_t* hist_item,
                                             const int64_t* hist_cat, const int64_t* tgt_item_seq,
                                             const int64_t* tgt_cat_seq, const float* w_hist_item,
                                             const float* w_hist_cat, const float* w_tgt_item_seq,
                                             const float* w_tgt_cat_seq, const float* att_w1,
                                             const float* att_w1_t, const float* att_b1,
                                             const float* att_w2, const float* att_b2,
                                             const float* att_w3, const float* att_weight,
                                             const float* out_saved, const float* act1_saved,
                                             const float* d_out, float* d_hist, float* d_tgt_seq,
                                             void* workspace, size_t workspace_bytes, void* stream) {
  REC_REQUIRE(d, REC_EINVAL, "desc is NULL");
  const int E = d->item_dim + d->cat_dim;
  REC_REQUIRE(d->batch >= 0 && d->max_len > 0 && d->item_dim > 0 && d->cat_dim > 0 && d->hidden1 > 0 &&
                  d->hidden2 > 0 && d->item_rows > 0 && d->cat_rows > 0, REC_EINVAL, "bad sizes");
  REC_REQUIRE(d->item_dim % 4 == 0 && d->cat_dim % 4 == 0 && E <= kBlock / 2, REC_ESHAPE,
              "item_dim/cat_dim must be multiples of 4 with item_dim+cat_dim <= %d", kBlock / 2);
  REC_REQUIRE(d->hidden1 <= kBlock / 2 && d->hidden2 <= kBlock / 2, REC_ESHAPE, "hidden sizes must be <= %d",
              kBlock / 2);
  REC_REQUIRE(d->item_stride >= d->item_dim && d->cat_stride >= d->cat_dim && d->item_stride % 4 == 0 &&
                  d->cat_stride % 4 == 0, REC_EINVAL, "table strides must be multiples of 4 and >= dims");
  if (d->batch == 0) return REC_OK;
  REC_REQUIRE(hist_item && hist_cat && tgt_item_seq && tgt_cat_seq && w_hist_item && w_hist_cat &&
                  w_tgt_item_seq && w_tgt_cat_seq && att_w1 && att_w1_t && att_b1 && att_w2 && att_b2 &&
                  att_w3 && att_weight && d_out && d_hist && d_tgt_seq, REC_EINVAL, "null pointer argument");
  const int H1 = d->hidden1, H2 = d->hidden2;
  const size_t shmem = sizeof(float) * ((size_t)2 * kDinTP * E + (size_t)kDinTP * (H1 + H2) + (size_t)H1 * H2 +
                                        2 * H2 + H1 + kDinTP + kBlock / kWave + (size_t)d->max_len);
  DinBwdArgs g;
  DinArgs& a = g.f;
  a.B = d->batch; a.T = d->max_len; a.Ei = d->item_dim; a.Ec = d->cat_dim; a.H1 = H1; a.H2 = H2;
  a.n_item = d->item_rows; a.n_cat = d->cat_rows; a.ld_item = d->item_stride; a.ld_cat = d->cat_stride;
  a.hist_item = hist_item; a.hist_cat = hist_cat; a.tgt_item = tgt_item_seq; a.tgt_cat = tgt_cat_seq;
  a.mask = nullptr; a.w_hist_item = w_hist_item; a.w_hist_cat = w_hist_cat; a.w_tgt_item = w_tgt_item_seq;
  a.w_tgt_cat = w_tgt_cat_seq; a.w1 = att_w1; a.b1 = att_b1; a.w2 = att_w2; a.b2 = att_b2; a.w3 = att_w3;
  a.b3 = nullptr; a.out = nullptr; a.att_weight = nullptr; a.status = nullptr; a.act1 = nullptr;
  g.w1t = att_w1_t; g.att_weight = att_weight; g.dout = d_out; g.dh = d_hist; g.dq = d_tgt_seq;
  g.out_saved = out_saved; g.act1 = act1_saved;
  g.ticket = nullptr;
  static const bool force_generic = getenv("REC_DIN_BWD_GENERIC") != nullptr;
  if (!force_generic && out_saved && act1_saved && E == 128 && H1 == 80 && H2 == 40) {
    using Ct = DinBwdCt<128, 80, 40>;
    auto kern = din_attention_bwd_ct_kernel<128, 80, 40>;
    static const hipError_t attr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       (int)kDinCtLdsMax);
    (void)attr;
    int64_t grid = resident_blocks(kern, kBlock, Ct::lds_bytes);
    const int64_t tiles = d->batch * ((d->max_len + kDinTP - 1) / kDinTP);
    static const bool split_env = [] { const char* v = getenv("REC_DIN_TILE_SPLIT"); return !(v && *v == '0'); }();
    const int64_t items = split_env ? tiles : d->batch;      // REC_DIN_TILE_SPLIT=0: at most one block per sample
    if (grid > items) grid = items;
    static const bool ticket_env = [] { const char* v = getenv("REC_DIN_TICKETS"); return !(v && *v == '0'); }();
    if (ticket_env && split_env && workspace && workspace_bytes >= sizeof(unsigned int) && d->max_len > kDinTP &&
        d->batch > grid && d->batch < (1ll << 31) - 4 * grid) {
      g.ticket = (unsigned int*)workspace;
      REC_REQUIRE(hipMemsetAsync(workspace, 0, sizeof(unsigned int), (hipStream_t)stream) == hipSuccess, REC_EHIP,
                  "hipMemsetAsync failed");
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kBlock), Ct::lds_bytes, (hipStream_t)stream, g);
    return check_launch("rec_din_attention_pool_bwd");
  }
  REC_REQUIRE(shmem <= 64 * 1024, REC_ESHAPE, "history too long for the LDS buffers (%zu B)", shmem);
  int64_t grid = resident_blocks(din_attention_bwd_kernel, kBlock, shmem);
  if (grid > d->batch) grid = d->batch;
  hipLaunchKernelGGL(din_attention_bwd_kernel, dim3((unsigned)grid), dim3(kBlock), shmem,
                     (hipStream_t)stream, g);
  return check_launch("rec_din_attention_pool_bwd");
}
