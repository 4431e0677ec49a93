// CrossNetV2 backward glue (gfx950): the two elementwise products between the GEMMs.
//
// Forward layer (/root/reference/models/rank/dcn_v2/net.py:222-226): X_{l+1} = X_l + X_0 * U_l,
// U_l = X_l W_l + b_l (rec_gemm_f32, REC_EPI_CROSS, U_l saved through out2).  Backward of one layer,
// given dX = d X_{l+1}:
//     dU       = dX * X_0            -> feeds  dW_l = X_l^T dU (b_colsum = db_l),  dX_l = dX + dU W_l^T
//     dX0_acc += dX * U_l            (gradient reaching X_0 through the Hadamard product)
// One streaming pass reads dX, X_0, U_l once and writes both results (HBM-bound, float4 per lane).
#include "rec_common.h"
#include "tail_roles.h"

namespace rec {

__global__ __launch_bounds__(kBlock) void cross_bwd_prep_kernel(
    int64_t M, int N, const float* __restrict__ dX, int64_t ld_dx, const float* __restrict__ X0,
    int64_t ld_x0, const float* __restrict__ U, int64_t ld_u, float* __restrict__ dU, int64_t ld_du,
    float* __restrict__ acc, int64_t ld_acc, int accumulate, bool vec) {
  if (vec) {
    const int n4 = N / 4;
    const int64_t total = M * n4;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * kBlock) {
      const int64_t i = e / n4;
      const int j = (int)(e % n4) * 4;
      const float4 g = *reinterpret_cast<const float4*>(dX + i * ld_dx + j);
      const float4 x = *reinterpret_cast<const float4*>(X0 + i * ld_x0 + j);
      const float4 u = *reinterpret_cast<const float4*>(U + i * ld_u + j);
      *reinterpret_cast<float4*>(dU + i * ld_du + j) = make_float4(g.x * x.x, g.y * x.y, g.z * x.z, g.w * x.w);
      float4 a = make_float4(g.x * u.x, g.y * u.y, g.z * u.z, g.w * u.w);
      float4* ap = reinterpret_cast<float4*>(acc + i * ld_acc + j);
      if (accumulate) {
        const float4 o = *ap;
        a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
      }
      *ap = a;
    }
  } else {
    const int64_t total = M * N;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * kBlock) {
      const int64_t i = e / N;
      const int j = (int)(e % N);
      const float g = dX[i * ld_dx + j];
      dU[i * ld_du + j] = g * X0[i * ld_x0 + j];
      const float a = g * U[i * ld_u + j];
      acc[i * ld_acc + j] = accumulate ? acc[i * ld_acc + j] + a : a;
    }
  }
}

// CrossNetMix backward glue for one expert (dcn_v2/net.py:301-317), one wave per row, float4 lanes:
//   du  = dX * x0 * p_e            (gradient of u_e = U_e v + bias)
//   acc = (accumulate ? acc : 0) + dX * p_e * u_e      (gradient reaching x_0 through the Hadamard product)
//   dp_e[i] = sum_j dX * x0 * u_e                       (gradient of the gate probability)
__global__ __launch_bounds__(kBlock) void moe_bwd_prep_kernel(
    int64_t M, int N, const float* __restrict__ dX, int64_t ld_dx, const float* __restrict__ X0,
    int64_t ld_x0, const float* __restrict__ U, int64_t ld_u, const float* __restrict__ prob,
    int64_t prob_stride, float* __restrict__ dU, int64_t ld_du, float* __restrict__ acc, int64_t ld_acc,
    int accumulate, float* __restrict__ dp, int64_t dp_stride) {
  const int lane = threadIdx.x % kWave;
  const int64_t wpb = kBlock / kWave;
  for (int64_t i = (int64_t)blockIdx.x * wpb + threadIdx.x / kWave; i < M; i += (int64_t)gridDim.x * wpb) {
    const float pe = prob[i * prob_stride];
    float dot = 0.f;
    for (int j = lane; j < N; j += kWave) {
      const float g = dX[i * ld_dx + j], x = X0[i * ld_x0 + j], u = U[i * ld_u + j];
      dU[i * ld_du + j] = g * x * pe;
      const float a = g * pe * u;
      acc[i * ld_acc + j] = accumulate ? acc[i * ld_acc + j] + a : a;
      dot += g * x * u;
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) dot += __shfl_xor(dot, o, kWave);
    if (lane == 0) dp[i * dp_stride] = dot;
  }
}

// softmax backward over a handful of columns: dz = p * (dp - sum_e p_e dp_e)
__global__ __launch_bounds__(kBlock) void softmax_rows_bwd_kernel(int64_t M, int E, const float* __restrict__ p,
                                                                  int64_t ldp, const float* __restrict__ dp,
                                                                  int64_t lddp, float* __restrict__ dz,
                                                                  int64_t lddz) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < M; i += (int64_t)gridDim.x * kBlock) {
    float s = 0.f;
    for (int e = 0; e < E; ++e) s += p[i * ldp + e] * dp[i * lddp + e];
    for (int e = 0; e < E; ++e) dz[i * lddz + e] = p[i * ldp + e] * (dp[i * lddp + e] - s);
  }
}

// row-wise softmax over a handful of columns (CrossNetMix gate, dcn_v2/net.py:315: E = num_experts)
__global__ __launch_bounds__(kBlock) void softmax_rows_kernel(int64_t M, int E,
                                                              const float* __restrict__ x, int64_t ldx,
                                                              float* __restrict__ y, int64_t ldy) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < M; i += (int64_t)gridDim.x * kBlock) {
    float mx = x[i * ldx];
    for (int e = 1; e < E; ++e) mx = fmaxf(mx, x[i * ldx + e]);
    float sum = 0.f;
    for (int e = 0; e < E; ++e) sum += expf(x[i * ldx + e] - mx);
    const float inv = 1.f / sum;
    for (int e = 0; e < E; ++e) y[i * ldy + e] = expf(x[i * ldx + e] - mx) * inv;
  }
}


// ------------------------------------------------------------------ train-mode Dropout + L2Decay (DCN-v2 DNN tower)
// /root/reference/models/rank/dcn_v2/net.py:158,181-183: `y_dnn = self.drop_out(y_dnn)` after EVERY element of
// _mlp_layers — after the Linear and again after its ReLU — paddle.nn.Dropout(p = 0.5), mode upscale_in_train [EXT]:
// y = x * keep / (1 - p).  Paddle draws its masks from the device generator (not reproducible from the reference);
// here keep(i) is a pure function of (seed, stream, element): bit 32.. of mix64(seed ^ mix64(stream << 40 | i))
// against p * 2^32, so the forward, the backward (same kernel on the gradient) and the oracle agree without a stored
// mask.  Up to two streams are applied at once (the two dropouts around a ReLU commute with it: relu(x * c) =
// c * relu(x) for c >= 0): keep = keepA & keepB, scale = 1 / (1 - p)^nmask.
__global__ __launch_bounds__(kBlock) void dropout_kernel(int64_t rows, int cols, int64_t ld_in, int64_t ld_out,
                                                         const float* __restrict__ in, float* __restrict__ out,
                                                         uint32_t thresh, float scale, uint64_t seed,
                                                         uint64_t stream_a, uint64_t stream_b, int nmask) {
  const int64_t n = rows * cols;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += (int64_t)gridDim.x * kBlock) {
    const int64_t r = e / cols;
    const int c = (int)(e - r * cols);
    // stream_a / stream_b arrive as the per-stream keys mix64(seed ^ mix64(stream + golden)) (rec_dropout): any 64-bit
    // stream id, any element index — no packing of the two into one word (ids above 2^24 failed)
    bool keep = (uint32_t)(mix64(stream_a + (uint64_t)e) >> 32) >= thresh;
    if (nmask > 1) keep = keep && (uint32_t)(mix64(stream_b + (uint64_t)e) >> 32) >= thresh;
    const float x = in[r * ld_in + c];
    out[r * ld_out + c] = keep ? x * scale : 0.f;
  }
}

// grad += (coeff / grad_scale) * w : paddle.regularizer.L2Decay(coeff) (dcn_v2/net.py:164-170) appended to the
// gradient AFTER gradient clipping [EXT Optimizer._apply_optimize: clip, then regularization] — the Adam kernels
// multiply the whole gradient by the clipping coefficient, so the term is pre-divided by it.
__global__ __launch_bounds__(kBlock) void l2_decay_grad_kernel(int64_t n, float* __restrict__ grad,
                                                               const float* __restrict__ w, float coeff,
                                                               const float* __restrict__ grad_scale) {
  const float c = grad_scale ? coeff / grad_scale[0] : coeff;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
    grad[i] += c * w[i];
}

}  // namespace rec

using namespace rec;

extern "C" int rec_softmax_rows(int64_t m, int32_t n, const float* x, int32_t ldx, float* y,
                                int32_t ldy, void* stream) {
  REC_REQUIRE(m >= 0 && n > 0 && n <= 64 && ldx >= n && ldy >= n, REC_EINVAL, "bad sizes (n <= 64)");
  if (m == 0) return REC_OK;
  REC_REQUIRE(x && y, REC_EINVAL, "null pointer argument");
  int64_t grid = (m + kBlock - 1) / kBlock;
  if (grid > kNumCU * 8) grid = kNumCU * 8;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream, m,
                     n, x, (int64_t)ldx, y, (int64_t)ldy);
  return check_launch("rec_softmax_rows");
}

extern "C" int rec_cross_bwd_prep(int64_t m, int32_t n, const float* dX, int32_t ld_dx,
                                  const float* X0, int32_t ld_x0, const float* U, int32_t ld_u,
                                  float* dU, int32_t ld_du, float* dX0_acc, int32_t ld_acc,
                                  int32_t accumulate, void* stream) {
  REC_REQUIRE(m >= 0 && n > 0 && ld_dx >= n && ld_x0 >= n && ld_u >= n && ld_du >= n && ld_acc >= n,
              REC_EINVAL, "bad sizes");
  if (m == 0) return REC_OK;
  REC_REQUIRE(dX && X0 && U && dU && dX0_acc, REC_EINVAL, "null pointer argument");
  const bool vec = n % 4 == 0 && ld_dx % 4 == 0 && ld_x0 % 4 == 0 && ld_u % 4 == 0 && ld_du % 4 == 0 &&
                   ld_acc % 4 == 0 &&
                   (((uintptr_t)dX | (uintptr_t)X0 | (uintptr_t)U | (uintptr_t)dU | (uintptr_t)dX0_acc) % 16) == 0;
  int64_t grid = (m * (int64_t)n / (vec ? 4 : 1) + kBlock - 1) / kBlock;
  if (grid > kNumCU * 8) grid = kNumCU * 8;
  hipLaunchKernelGGL(cross_bwd_prep_kernel, dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream,
                     m, n, dX, (int64_t)ld_dx, X0, (int64_t)ld_x0, U, (int64_t)ld_u, dU, (int64_t)ld_du,
                     dX0_acc, (int64_t)ld_acc, accumulate, vec);
  return check_launch("rec_cross_bwd_prep");
}

extern "C" int rec_moe_bwd_prep(int64_t m, int32_t n, const float* dX, int32_t ld_dx, const float* X0,
                                int32_t ld_x0, const float* U, int32_t ld_u, const float* prob,
                                int32_t prob_stride, float* dU, int32_t ld_du, float* dX0_acc,
                                int32_t ld_acc, int32_t accumulate, float* dp, int32_t dp_stride,
                                void* stream) {
  REC_REQUIRE(m >= 0 && n > 0 && ld_dx >= n && ld_x0 >= n && ld_u >= n && ld_du >= n && ld_acc >= n &&
                  prob_stride >= 1 && dp_stride >= 1, REC_EINVAL, "bad sizes");
  if (m == 0) return REC_OK;
  REC_REQUIRE(dX && X0 && U && prob && dU && dX0_acc && dp, REC_EINVAL, "null pointer argument");
  int64_t grid = (m + kBlock / kWave - 1) / (kBlock / kWave);
  if (grid > kNumCU * 8) grid = kNumCU * 8;
  hipLaunchKernelGGL(moe_bwd_prep_kernel, dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream, m, n,
                     dX, (int64_t)ld_dx, X0, (int64_t)ld_x0, U, (int64_t)ld_u, prob, (int64_t)prob_stride, dU,
                     (int64_t)ld_du, dX0_acc, (int64_t)ld_acc, accumulate, dp, (int64_t)dp_stride);
  return check_launch("rec_moe_bwd_prep");
}

extern "C" int rec_softmax_rows_bwd(int64_t m, int32_t n, const float* p, int32_t ldp, const float* dp,
                                    int32_t lddp, float* dz, int32_t lddz, void* stream) {
  REC_REQUIRE(m >= 0 && n > 0 && n <= 64 && ldp >= n && lddp >= n && lddz >= n, REC_EINVAL, "bad sizes");
  if (m == 0) return REC_OK;
  REC_REQUIRE(p && dp && dz, REC_EINVAL, "null pointer argument");
  int64_t grid = (m + kBlock - 1) / kBlock;
  if (grid > kNumCU * 8) grid = kNumCU * 8;
  hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream, m, n,
                     p, (int64_t)ldp, dp, (int64_t)lddp, dz, (int64_t)lddz);
  return check_launch("rec_softmax_rows_bwd");
}

// ---------------------------------------------------------------------------------------------
// Dense-field folding for the DeepFM top MLP (rec_deepfm_desc.compact_dense): tiny parameter-space kernels.
namespace rec {
__global__ __launch_bounds__(kBlock) void dense_fold_fwd_kernel(int S, int Dn, int D, int NO,
                                                                const float* __restrict__ dw,
                                                                const float* __restrict__ W0,
                                                                float* __restrict__ M) {
  for (int e = blockIdx.x * kBlock + threadIdx.x; e < Dn * NO; e += gridDim.x * kBlock) {
    const int j = e / NO, n = e % NO;
    float t = 0.f;
    for (int d = 0; d < D; ++d) t += dw[j * D + d] * W0[(int64_t)((S + j) * D + d) * NO + n];
    M[e] = t;
  }
}
__global__ __launch_bounds__(kBlock) void dense_fold_bwd_w_kernel(int S, int Dn, int D, int NO,
                                                                  const float* __restrict__ dw,
                                                                  const float* __restrict__ dM,
                                                                  float* __restrict__ dW0) {
  for (int e = blockIdx.x * kBlock + threadIdx.x; e < Dn * D * NO; e += gridDim.x * kBlock) {
    const int n = e % NO, jd = e / NO, j = jd / D;
    dW0[(int64_t)(S * D + jd) * NO + n] = dw[jd] * dM[j * NO + n];
  }
}
// one block per (j,d): d_dense_w[j,d] (+)= sum_n dM[j,n] * W0[(S+j)*D+d, n]   (fixed-order tree)
__global__ __launch_bounds__(kBlock) void dense_fold_bwd_dw_kernel(int S, int D, int NO,
                                                                   const float* __restrict__ W0,
                                                                   const float* __restrict__ dM,
                                                                   float* __restrict__ ddw, int accumulate) {
  __shared__ float red[kBlock];
  const int jd = blockIdx.x, j = jd / D;
  float t = 0.f;
  for (int n = threadIdx.x; n < NO; n += kBlock) t += dM[j * NO + n] * W0[(int64_t)(S * D + jd) * NO + n];
  red[threadIdx.x] = t;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) ddw[jd] = (accumulate ? ddw[jd] : 0.f) + red[0];
}
// The whole fold of a step in ONE launch each way (the copy of the sparse rows was a hipMemcpyAsync of its own and the
// backward two kernels behind a copy: five launches on the critical path of a step that is ~30 launches long at the
// reference's batch size).  Block roles by index range; every element is computed exactly as by the kernels above.
//   fwd:  W0f[r, :] = W0[r, :] for r < S*D;   W0f[S*D + j, :] = M[j, :]
__global__ __launch_bounds__(kBlock) void dense_fold_fwd_full_kernel(FoldFwd r) {
  dense_fold_fwd_role(blockIdx.x, threadIdx.x, r);      // tail_roles.h
}
//   bwd:  dW0f [(S+1)*D, NO] = feat'^T dZ0 (its rows S*D.. hold dM):
//         dW0[r, :] = dW0f[r, :] for r < S*D;   dW0[S*D + jd, n] = dw[jd] * dM[jd / D, n];
//         d_dense_w[jd] (+)= sum_n dM[jd / D, n] * W0[S*D + jd, n]
__global__ __launch_bounds__(kBlock) void dense_fold_bwd_full_kernel(int S, int Dn, int D, int NO, int copy_blocks,
                                                                     int w_blocks, const float* __restrict__ dw,
                                                                     const float* __restrict__ W0,
                                                                     const float* __restrict__ dW0f,
                                                                     float* __restrict__ dW0, float* __restrict__ ddw,
                                                                     int accumulate) {
  const float* dM = dW0f + (int64_t)S * D * NO;
  int b = (int)blockIdx.x;
  if (b < copy_blocks) {
    const int64_t total = (int64_t)S * D * NO;
    for (int64_t e = (int64_t)b * kBlock + threadIdx.x; e < total; e += (int64_t)copy_blocks * kBlock) dW0[e] = dW0f[e];
    return;
  }
  b -= copy_blocks;
  if (b < w_blocks) {
    for (int e = b * kBlock + threadIdx.x; e < Dn * D * NO; e += w_blocks * kBlock) {
      const int n = e % NO, jd = e / NO, j = jd / D;
      dW0[(int64_t)(S * D + jd) * NO + n] = dw[jd] * dM[j * NO + n];
    }
    return;
  }
  b -= w_blocks;                                  // one block per (j, d): fixed-order tree, as dense_fold_bwd_dw_kernel
  __shared__ float red[kBlock];
  const int jd = b, j = jd / D;
  float t = 0.f;
  for (int n = threadIdx.x; n < NO; n += kBlock) t += dM[j * NO + n] * W0[(int64_t)(S * D + jd) * NO + n];
  red[threadIdx.x] = t;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) ddw[jd] = (accumulate ? ddw[jd] : 0.f) + red[0];
}
}  // namespace rec

rec::FoldFwd rec::dense_fold_fwd_plan(int32_t num_slots, int32_t num_dense, int32_t emb_dim, int32_t n_out,
                                      const float* dense_w, const float* W0, float* W0_folded) {
  const int64_t copy_elems = (int64_t)num_slots * emb_dim * n_out;
  int copy_blocks = (int)((copy_elems + rec::kBlock * 4 - 1) / (rec::kBlock * 4));
  if (copy_blocks > 512) copy_blocks = 512;
  const int m_blocks = (num_dense * n_out + rec::kBlock - 1) / rec::kBlock;
  return rec::FoldFwd{copy_blocks + m_blocks, num_slots, num_dense, emb_dim, n_out, copy_blocks, dense_w, W0, W0_folded};
}

extern "C" int rec_dense_fold_fwd_full(int32_t num_slots, int32_t num_dense, int32_t emb_dim, int32_t n_out,
                                       const float* dense_w, const float* W0, float* W0_folded, void* stream) {
  REC_REQUIRE(num_slots > 0 && num_dense > 0 && emb_dim > 0 && n_out > 0 && dense_w && W0 && W0_folded, REC_EINVAL,
              "bad arguments");
  const rec::FoldFwd r = rec::dense_fold_fwd_plan(num_slots, num_dense, emb_dim, n_out, dense_w, W0, W0_folded);
  hipLaunchKernelGGL(rec::dense_fold_fwd_full_kernel, dim3(r.blocks), dim3(rec::kBlock), 0, (hipStream_t)stream, r);
  return rec::check_launch("rec_dense_fold_fwd_full");
}

extern "C" int rec_dense_fold_bwd_full(int32_t num_slots, int32_t num_dense, int32_t emb_dim, int32_t n_out,
                                       const float* dense_w, const float* W0, const float* dW0_folded, float* dW0,
                                       float* d_dense_w, int32_t accumulate_ddw, void* stream) {
  REC_REQUIRE(num_slots > 0 && num_dense > 0 && emb_dim > 0 && n_out > 0 && dense_w && W0 && dW0_folded && dW0 &&
                  d_dense_w && dW0_folded != dW0, REC_EINVAL, "bad arguments");
  const int64_t copy_elems = (int64_t)num_slots * emb_dim * n_out;
  int copy_blocks = (int)((copy_elems + rec::kBlock * 4 - 1) / (rec::kBlock * 4));
  if (copy_blocks > 512) copy_blocks = 512;
  int w_blocks = (num_dense * emb_dim * n_out + rec::kBlock * 2 - 1) / (rec::kBlock * 2);
  if (w_blocks > 512) w_blocks = 512;
  hipLaunchKernelGGL(rec::dense_fold_bwd_full_kernel, dim3(copy_blocks + w_blocks + num_dense * emb_dim),
                     dim3(rec::kBlock), 0, (hipStream_t)stream, num_slots, num_dense, emb_dim, n_out, copy_blocks,
                     w_blocks, dense_w, W0, dW0_folded, dW0, d_dense_w, accumulate_ddw);
  return rec::check_launch("rec_dense_fold_bwd_full");
}

extern "C" int rec_dense_fold_fwd(int32_t num_slots, int32_t num_dense, int32_t emb_dim, int32_t n_out,
                                  const float* dense_w, const float* W0, float* M, void* stream) {
  REC_REQUIRE(num_slots > 0 && num_dense > 0 && emb_dim > 0 && n_out > 0 && dense_w && W0 && M, REC_EINVAL,
              "bad arguments");
  const int total = num_dense * n_out;
  hipLaunchKernelGGL(rec::dense_fold_fwd_kernel, dim3((total + rec::kBlock - 1) / rec::kBlock), dim3(rec::kBlock),
                     0, (hipStream_t)stream, num_slots, num_dense, emb_dim, n_out, dense_w, W0, M);
  return rec::check_launch("rec_dense_fold_fwd");
}

extern "C" int rec_dense_fold_bwd(int32_t num_slots, int32_t num_dense, int32_t emb_dim, int32_t n_out,
                                  const float* dense_w, const float* W0, const float* dM, float* dW0,
                                  float* d_dense_w, int32_t accumulate_ddw, void* stream) {
  REC_REQUIRE(num_slots > 0 && num_dense > 0 && emb_dim > 0 && n_out > 0 && dense_w && W0 && dM && dW0 &&
                  d_dense_w, REC_EINVAL, "bad arguments");
  const int total = num_dense * emb_dim * n_out;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(rec::dense_fold_bwd_w_kernel, dim3((total + rec::kBlock - 1) / rec::kBlock),
                     dim3(rec::kBlock), 0, st, num_slots, num_dense, emb_dim, n_out, dense_w, dM, dW0);
  hipLaunchKernelGGL(rec::dense_fold_bwd_dw_kernel, dim3(num_dense * emb_dim), dim3(rec::kBlock), 0, st,
                     num_slots, emb_dim, n_out, W0, dM, d_dense_w, accumulate_ddw);
  return rec::check_launch("rec_dense_fold_bwd");
}

extern "C" int rec_dropout(int64_t rows, int32_t cols, int64_t ld_in, int64_t ld_out, const float* in, float* out,
                           float p, uint64_t seed, uint64_t stream_a, uint64_t stream_b, int32_t nmask,
                           void* stream) {
  REC_REQUIRE(rows >= 0 && cols > 0 && ld_in >= cols && ld_out >= cols && in && out, REC_EINVAL, "bad arguments");
  REC_REQUIRE(p >= 0.f && p < 1.f && (nmask == 1 || nmask == 2), REC_EINVAL, "p must be in [0,1), nmask 1 or 2");
  if (rows == 0) return REC_OK;
  const uint64_t key_a = mix64(seed ^ mix64(stream_a + 0x9E3779B97F4A7C15ull));
  const uint64_t key_b = mix64(seed ^ mix64(stream_b + 0x9E3779B97F4A7C15ull));
  const uint32_t thresh = (uint32_t)((double)p * 4294967296.0);
  float scale = 1.f / (1.f - p);
  if (nmask == 2) scale *= scale;
  int64_t grid = (rows * cols + kBlock - 1) / kBlock;
  if (grid > kNumCU * 16) grid = kNumCU * 16;
  hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream, rows, cols, ld_in,
                     ld_out, in, out, thresh, scale, seed, key_a, key_b, nmask);
  return check_launch("rec_dropout");
}

extern "C" int rec_l2_decay_grad(int64_t n, float* grad, const float* w, float coeff, const float* grad_scale,
                                 void* stream) {
  REC_REQUIRE(n >= 0 && grad && w && coeff >= 0.f, REC_EINVAL, "bad arguments");
  if (n == 0 || coeff == 0.f) return REC_OK;
  int64_t grid = (n + kBlock - 1) / kBlock;
  if (grid > kNumCU * 16) grid = kNumCU * 16;
  hipLaunchKernelGGL(l2_decay_grad_kernel, dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream, n, grad, w,
                     coeff, grad_scale);
  return check_launch("rec_l2_decay_grad");
}
