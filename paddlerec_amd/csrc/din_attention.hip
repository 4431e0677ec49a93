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
#include "rec_common.h"

namespace rec {

constexpr int kDinTP = 32;       // positions per tile
constexpr int kDinNP1 = 16;      // max positions per thread in layer 1 (PG >= 2)
constexpr int kDinNP2 = 16;

struct DinArgs {
  int64_t B;
  int T, Ei, Ec, H1, H2;
  int64_t n_item, n_cat;
  int ld_item, ld_cat;
  const int64_t *hist_item, *hist_cat, *tgt_item, *tgt_cat, *mask;
  const float *w_hist_item, *w_hist_cat, *w_tgt_item, *w_tgt_cat;
  const float *w1, *b1, *w2, *b2, *w3, *b3;
  float *out, *att_weight;
  int32_t* status;
};

__device__ __forceinline__ float sigmoidf_(float z) { return 1.f / (1.f + expf(-z)); }
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
      if (on2) {
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
  float *dh, *dq;          // [B,T,E]
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
      if (on2) {   // recompute layer 2
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

}  // namespace rec

using namespace rec;

extern "C" int rec_din_attention_pool_fwd(const rec_din_desc* d, const int64_t* hist_item,
                                          const int64_t* hist_cat, const int64_t* tgt_item_seq,
                                          const int64_t* tgt_cat_seq, const int64_t* mask,
                                          const float* w_hist_item, const float* w_hist_cat,
                                          const float* w_tgt_item_seq, const float* w_tgt_cat_seq,
                                          const float* att_w1, const float* att_b1,
                                          const float* att_w2, const float* att_b2,
                                          const float* att_w3, const float* att_b3, float* out,
                                          float* att_weight, int32_t* status, void* stream) {
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
  const size_t shmem = sizeof(float) * ((size_t)2 * kDinTP * (E + 4) + (size_t)kDinTP * (H1 + H2) +
                                        (size_t)H1 * H2 + 2 * H2 + H1 + (size_t)d->max_len);
  REC_REQUIRE(shmem <= 64 * 1024, REC_ESHAPE, "history too long for the LDS logit buffer (%zu B)", shmem);
  DinArgs a;
  a.B = d->batch; a.T = d->max_len; a.Ei = d->item_dim; a.Ec = d->cat_dim; a.H1 = H1; a.H2 = H2;
  a.n_item = d->item_rows; a.n_cat = d->cat_rows; a.ld_item = d->item_stride; a.ld_cat = d->cat_stride;
  a.hist_item = hist_item; a.hist_cat = hist_cat; a.tgt_item = tgt_item_seq; a.tgt_cat = tgt_cat_seq;
  a.mask = mask; a.w_hist_item = w_hist_item; a.w_hist_cat = w_hist_cat; a.w_tgt_item = w_tgt_item_seq;
  a.w_tgt_cat = w_tgt_cat_seq; a.w1 = att_w1; a.b1 = att_b1; a.w2 = att_w2; a.b2 = att_b2; a.w3 = att_w3;
  a.b3 = att_b3; a.out = out; a.att_weight = att_weight; a.status = status;
  int64_t grid = resident_blocks(din_attention_fwd_kernel, kBlock, shmem);
  if (grid > d->batch) grid = d->batch;
  hipLaunchKernelGGL(din_attention_fwd_kernel, dim3((unsigned)grid), dim3(kBlock), shmem,
                     (hipStream_t)stream, a);
  return check_launch("rec_din_attention_pool_fwd");
}

extern "C" int rec_din_attention_pool_bwd(const rec_din_desc* d, const int64_t* hist_item,
                                          const int64_t* hist_cat, const int64_t* tgt_item_seq,
                                          const int64_t* tgt_cat_seq, const float* w_hist_item,
                                          const float* w_hist_cat, const float* w_tgt_item_seq,
                                          const float* w_tgt_cat_seq, const float* att_w1,
                                          const float* att_w1_t, const float* att_b1,
                                          const float* att_w2, const float* att_b2,
                                          const float* att_w3, const float* att_weight,
                                          const float* d_out, float* d_hist, float* d_tgt_seq,
                                          void* stream) {
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
  REC_REQUIRE(shmem <= 64 * 1024, REC_ESHAPE, "history too long for the LDS buffers (%zu B)", shmem);
  DinBwdArgs g;
  DinArgs& a = g.f;
  a.B = d->batch; a.T = d->max_len; a.Ei = d->item_dim; a.Ec = d->cat_dim; a.H1 = H1; a.H2 = H2;
  a.n_item = d->item_rows; a.n_cat = d->cat_rows; a.ld_item = d->item_stride; a.ld_cat = d->cat_stride;
  a.hist_item = hist_item; a.hist_cat = hist_cat; a.tgt_item = tgt_item_seq; a.tgt_cat = tgt_cat_seq;
  a.mask = nullptr; a.w_hist_item = w_hist_item; a.w_hist_cat = w_hist_cat; a.w_tgt_item = w_tgt_item_seq;
  a.w_tgt_cat = w_tgt_cat_seq; a.w1 = att_w1; a.b1 = att_b1; a.w2 = att_w2; a.b2 = att_b2; a.w3 = att_w3;
  a.b3 = nullptr; a.out = nullptr; a.att_weight = nullptr; a.status = nullptr;
  g.w1t = att_w1_t; g.att_weight = att_weight; g.dout = d_out; g.dh = d_hist; g.dq = d_tgt_seq;
  int64_t grid = resident_blocks(din_attention_bwd_kernel, kBlock, shmem);
  if (grid > d->batch) grid = d->batch;
  hipLaunchKernelGGL(din_attention_bwd_kernel, dim3((unsigned)grid), dim3(kBlock), shmem,
                     (hipStream_t)stream, g);
  return check_launch("rec_din_attention_pool_bwd");
}
