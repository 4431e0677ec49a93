// DLRM's pieces that are not a GEMM or a lookup (gfx950): BatchNorm1D over the batch, the pairwise dot interaction,
// top-1 accuracy counts.
//
// Reference: /root/reference/models/rank/dlrm/net.py
//   MLPLayer (:128-178): every layer is Linear -> ReLU -> BatchNorm1D (the `else` branch at :158 is never taken)
//   interaction (:96-121): T = [emb_1 .. emb_26, x] [B,27,D];  Z = bmm(T, T^T);  Zflat = the strictly upper triangle of Z
//                          in row-major order (triu + MIN_FLOAT mask + masked_select);  R = concat(x, Zflat)
//   dygraph_model.py:58-63 metrics: Auc + Accuracy of softmax(raw)
// BatchNorm1D [EXT Paddle batch_norm]: training normalises with the batch mean and the BIASED batch variance,
//   running = momentum * running + (1 - momentum) * batch (momentum 0.9, epsilon 1e-5, biased variance there too);
//   eval uses the running statistics.
// All of it is HBM-bound streaming: column statistics are two fixed-order passes (mean, then sum (x-mean)^2 — no
// E[x^2]-E[x]^2 cancellation), partials per row block combined in block order (deterministic).
#include "rec_common.h"

namespace rec {
namespace {

constexpr int kColTile = 64;        // columns per block (one wave-width of consecutive floats per row)
constexpr int kRowLanes = kBlock / kColTile;
constexpr int kMaxRowBlocks = 128;

// MODE 0: sum x              MODE 1: sum (x - mean)^2          MODE 2: sum dy, sum dy * (x - mean) * invstd
template <int MODE>
__global__ __launch_bounds__(kBlock) void bn_colreduce_kernel(int64_t M, int N, const float* __restrict__ X, int64_t ldx,
                                                              const float* __restrict__ dY, int64_t lddy,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, float* __restrict__ part0,
                                                              float* __restrict__ part1) {
  __shared__ float red[2][kRowLanes][kColTile];
  const int c = blockIdx.x * kColTile + threadIdx.x % kColTile;
  const int rl = threadIdx.x / kColTile;
  const bool on = c < N;
  const float mu = (MODE != 0 && on) ? mean[c] : 0.f;
  const float is = (MODE == 2 && on) ? invstd[c] : 0.f;
  // rows of this block: a contiguous range (fixed summation order whatever the grid)
  const int64_t per = (M + gridDim.y - 1) / gridDim.y;
  const int64_t r0 = (int64_t)blockIdx.y * per, r1 = r0 + per < M ? r0 + per : M;
  float a0 = 0.f, a1 = 0.f;
  if (on) {
    // four rows in flight per thread on four interleaved chains, folded in a fixed order (one dependent load-add chain
    // per thread made a [4096 x 512] reduce take 60 us: 23 + 19 + 9 % of a DLRM step at B 4096)
    float p0[4] = {0.f, 0.f, 0.f, 0.f}, p1[4] = {0.f, 0.f, 0.f, 0.f};
    int64_t r = r0 + rl;
    for (; r + 3 * kRowLanes < r1; r += 4 * kRowLanes) {
      float x[4], g[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        x[u] = X[(r + u * kRowLanes) * ldx + c];
        g[u] = MODE == 2 ? dY[(r + u * kRowLanes) * lddy + c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (MODE == 0) p0[u] += x[u];
        if (MODE == 1) { const float d = x[u] - mu; p0[u] += d * d; }
        if (MODE == 2) { p0[u] += g[u]; p1[u] += g[u] * (x[u] - mu) * is; }
      }
    }
    for (int u = 0; r < r1; r += kRowLanes, ++u) {
      const float x = X[r * ldx + c];
      if (MODE == 0) p0[u] += x;
      if (MODE == 1) { const float d = x - mu; p0[u] += d * d; }
      if (MODE == 2) { const float g = dY[r * lddy + c]; p0[u] += g; p1[u] += g * (x - mu) * is; }
    }
    a0 = (p0[0] + p0[1]) + (p0[2] + p0[3]);
    a1 = (p1[0] + p1[1]) + (p1[2] + p1[3]);
  }
  red[0][rl][threadIdx.x % kColTile] = a0;
  red[1][rl][threadIdx.x % kColTile] = a1;
  __syncthreads();
  if (rl == 0 && on) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < kRowLanes; ++i) { s0 += red[0][i][threadIdx.x]; s1 += red[1][i][threadIdx.x]; }
    part0[(int64_t)blockIdx.y * N + c] = s0;
    if (MODE == 2) part1[(int64_t)blockIdx.y * N + c] = s1;
  }
}

// STEP 0: mean = sum / M             STEP 1: var = sum / M -> invstd, running stats     STEP 2: dgamma, dbeta
template <int STEP>
__global__ __launch_bounds__(kBlock) void bn_finalize_kernel(int64_t M, int N, int GY, const float* __restrict__ part0,
                                                             const float* __restrict__ part1, float* __restrict__ out0,
                                                             float* __restrict__ out1, float* running_mean,
                                                             float* running_var, float momentum, float eps) {
  const int c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= N) return;
  float s0 = 0.f, s1 = 0.f;
  for (int g = 0; g < GY; ++g) {
    s0 += part0[(int64_t)g * N + c];
    if (STEP == 2) s1 += part1[(int64_t)g * N + c];
  }
  if (STEP == 0) out0[c] = s0 / (float)M;
  if (STEP == 1) {
    const float var = s0 / (float)M;                 // biased
    out1[c] = 1.f / sqrtf(var + eps);
    if (running_mean) running_mean[c] = momentum * running_mean[c] + (1.f - momentum) * out0[c];
    if (running_var) running_var[c] = momentum * running_var[c] + (1.f - momentum) * var;
  }
  if (STEP == 2) { out0[c] = s1; out1[c] = s0; }     // dgamma = sum dy * xhat, dbeta = sum dy
}

__global__ __launch_bounds__(kBlock) void bn_apply_kernel(int64_t M, int N, const float* __restrict__ X, int64_t ldx,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          float* __restrict__ Y, int64_t ldy) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= M * N) return;
  const int64_t r = i / N;
  const int c = (int)(i - r * N);
  Y[r * ldy + c] = (X[r * ldx + c] - mean[c]) * invstd[c] * gamma[c] + beta[c];
}

// dx = gamma * invstd * (dy - sum_dy / M - xhat * sum_dy_xhat / M);  relu_mask: the BN input was a ReLU output — its
// gradient passes only where that output was > 0 (the ReLU backward folded in, dlrm/net.py:146-152 order)
__global__ __launch_bounds__(kBlock) void bn_bwd_apply_kernel(int64_t M, int N, const float* __restrict__ X, int64_t ldx,
                                                              const float* __restrict__ dY, int64_t lddy,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ invstd,
                                                              const float* __restrict__ dgamma,
                                                              const float* __restrict__ dbeta, int relu_mask,
                                                              float* __restrict__ dX, int64_t lddx) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= M * N) return;
  const int64_t r = i / N;
  const int c = (int)(i - r * N);
  const float x = X[r * ldx + c];
  const float xhat = (x - mean[c]) * invstd[c];
  const float inv_m = 1.f / (float)M;
  float g = gamma[c] * invstd[c] * (dY[r * lddy + c] - dbeta[c] * inv_m - xhat * dgamma[c] * inv_m);
  if (relu_mask && !(x > 0.f)) g = 0.f;
  dX[r * lddx + c] = g;
}

// ---- pairwise dot interaction: one wave per sample, T staged in LDS
constexpr int kMaxFields = 64;

__global__ __launch_bounds__(kBlock) void dot_interact_fwd_kernel(int64_t B, int F, int D, const float* __restrict__ T,
                                                                  int64_t ldt, float* __restrict__ R, int64_t ldr) {
  extern __shared__ float smem[];
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  const int DP = D | 1;                                   // odd pitch: rows land on different banks
  float* ts = smem + (size_t)wave * F * DP;
  const int P = F * (F - 1) / 2;
  for (int64_t b = (int64_t)blockIdx.x * (kBlock / kWave) + wave; b < B; b += (int64_t)gridDim.x * (kBlock / kWave)) {
    const float* t = T + b * ldt;
    for (int i = lane; i < F * D; i += kWave) ts[(i / D) * DP + i % D] = t[i];
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    float* r = R + b * ldr;
    for (int d = lane; d < D; d += kWave) r[d] = ts[(F - 1) * DP + d];          // x = the last field (net.py:123)
    // pair p -> (i, j), i < j, row-major over the strict upper triangle
    int i = 0, row_start = 0;
    for (int p = lane; p < P; p += kWave) {
      while (p >= row_start + (F - 1 - i)) { row_start += F - 1 - i; ++i; }
      const int j = i + 1 + (p - row_start);
      float a = 0.f;
      for (int d = 0; d < D; ++d) a += ts[i * DP + d] * ts[j * DP + d];
      r[D + p] = a;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ __launch_bounds__(kBlock) void dot_interact_bwd_kernel(int64_t B, int F, int D, const float* __restrict__ T,
                                                                  int64_t ldt, const float* __restrict__ dR, int64_t ldr,
                                                                  float* __restrict__ dT, int64_t lddt) {
  extern __shared__ float smem[];
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  const int DP = D | 1, FP = F | 1;
  float* ts = smem + (size_t)wave * (F * DP + F * FP);
  float* zs = ts + F * DP;                                // dZ symmetric [F][FP], zero diagonal
  const int P = F * (F - 1) / 2;
  for (int64_t b = (int64_t)blockIdx.x * (kBlock / kWave) + wave; b < B; b += (int64_t)gridDim.x * (kBlock / kWave)) {
    const float* t = T + b * ldt;
    const float* dr = dR + b * ldr;
    for (int i = lane; i < F * D; i += kWave) ts[(i / D) * DP + i % D] = t[i];
    for (int i = lane; i < F; i += kWave) zs[i * FP + i] = 0.f;
    int i = 0, row_start = 0;
    for (int p = lane; p < P; p += kWave) {
      while (p >= row_start + (F - 1 - i)) { row_start += F - 1 - i; ++i; }
      const int j = i + 1 + (p - row_start);
      const float g = dr[D + p];
      zs[i * FP + j] = g;
      zs[j * FP + i] = g;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    float* o = dT + b * lddt;
    for (int e = lane; e < F * D; e += kWave) {
      const int f = e / D, d = e - f * D;
      float a = (f == F - 1) ? dr[d] : 0.f;               // the copy of x at the head of R
      for (int j = 0; j < F; ++j) a += zs[f * FP + j] * ts[j * DP + d];         // fixed order over j
      o[e] = a;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ __launch_bounds__(kBlock) void accuracy_kernel(int64_t n, const float* __restrict__ pred,
                                                          const int64_t* __restrict__ label,
                                                          unsigned long long* __restrict__ counts) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  int ok = 0;
  if (i < n) ok = ((pred[i] > 0.5f) ? 1 : 0) == (label[i] != 0 ? 1 : 0);
  const unsigned long long m = __ballot(ok);
  if (threadIdx.x % kWave == 0 && m) atomicAdd(counts, (unsigned long long)__popcll(m));     // integer: exact
  if (i == 0) atomicAdd(counts + 1, (unsigned long long)n);
}

int row_blocks(int64_t M) {     // 128 rows per block at small batches (enough blocks to fill the chip), up to 128 row blocks
  int64_t g = (M + 127) / 128;
  return (int)(g < 1 ? 1 : g > kMaxRowBlocks ? kMaxRowBlocks : g);
}

}  // namespace
}  // namespace rec

using namespace rec;

extern "C" int rec_batchnorm_workspace_bytes(int64_t m, int32_t n, size_t* bytes) {
  REC_REQUIRE(bytes && m >= 0 && n > 0, REC_EINVAL, "bad arguments");
  *bytes = sizeof(float) * (size_t)2 * kMaxRowBlocks * n;
  return REC_OK;
}

extern "C" int rec_batchnorm_fwd(int64_t m, int32_t n, const float* X, int64_t ldx, const float* gamma,
                                 const float* beta, float* running_mean, float* running_var, float momentum,
                                 float eps, int32_t training, float* Y, int64_t ldy, float* save_mean,
                                 float* save_invstd, void* workspace, size_t workspace_bytes, void* stream) {
  REC_REQUIRE(m > 0 && n > 0 && ldx >= n && ldy >= n, REC_EINVAL, "bad sizes");
  REC_REQUIRE(X && gamma && beta && Y && save_mean && save_invstd, REC_EINVAL, "null pointer argument");
  hipStream_t s = (hipStream_t)stream;
  const int64_t total = m * n;
  const unsigned eb = (unsigned)((total + kBlock - 1) / kBlock);
  if (training) {
    size_t need = 0;
    rec_batchnorm_workspace_bytes(m, n, &need);
    REC_REQUIRE(workspace && workspace_bytes >= need, REC_EWORKSPACE, "workspace too small (%zu < %zu)",
                workspace_bytes, need);
    float* part = (float*)workspace;
    const int gy = row_blocks(m);
    const dim3 grid((n + kColTile - 1) / kColTile, gy);
    const unsigned fb = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(bn_colreduce_kernel<0>, grid, dim3(kBlock), 0, s, m, n, X, ldx, nullptr, 0, nullptr, nullptr,
                       part, nullptr);
    hipLaunchKernelGGL(bn_finalize_kernel<0>, dim3(fb), dim3(kBlock), 0, s, m, n, gy, part, nullptr, save_mean,
                       nullptr, nullptr, nullptr, momentum, eps);
    hipLaunchKernelGGL(bn_colreduce_kernel<1>, grid, dim3(kBlock), 0, s, m, n, X, ldx, nullptr, 0, save_mean, nullptr,
                       part, nullptr);
    hipLaunchKernelGGL(bn_finalize_kernel<1>, dim3(fb), dim3(kBlock), 0, s, m, n, gy, part, nullptr, save_mean,
                       save_invstd, running_mean, running_var, momentum, eps);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(eb), dim3(kBlock), 0, s, m, n, X, ldx, gamma, beta, save_mean,
                       save_invstd, Y, ldy);
  } else {
    REC_REQUIRE(running_mean && running_var, REC_EINVAL, "eval mode needs the running statistics");
    // invstd of the running variance into save_invstd, running mean into save_mean: one tiny launch each way
    hipLaunchKernelGGL(bn_finalize_kernel<1>, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, (int64_t)1, n, 1,
                       running_var, nullptr, save_mean, save_invstd, nullptr, nullptr, momentum, eps);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(eb), dim3(kBlock), 0, s, m, n, X, ldx, gamma, beta, running_mean,
                       save_invstd, Y, ldy);
    if (hipMemcpyAsync(save_mean, running_mean, sizeof(float) * n, hipMemcpyDeviceToDevice, s) != hipSuccess) {
      set_error("rec_batchnorm_fwd: copy of the running mean failed");
      return REC_EHIP;
    }
  }
  return check_launch("rec_batchnorm_fwd");
}

extern "C" int rec_batchnorm_bwd(int64_t m, int32_t n, const float* X, int64_t ldx, const float* dY, int64_t lddy,
                                 const float* gamma, const float* save_mean, const float* save_invstd,
                                 int32_t relu_mask, float* dX, int64_t lddx, float* dgamma, float* dbeta,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  REC_REQUIRE(m > 0 && n > 0 && ldx >= n && lddy >= n && lddx >= n, REC_EINVAL, "bad sizes");
  REC_REQUIRE(X && dY && gamma && save_mean && save_invstd && dX && dgamma && dbeta, REC_EINVAL,
              "null pointer argument");
  size_t need = 0;
  rec_batchnorm_workspace_bytes(m, n, &need);
  REC_REQUIRE(workspace && workspace_bytes >= need, REC_EWORKSPACE, "workspace too small (%zu < %zu)", workspace_bytes,
              need);
  hipStream_t s = (hipStream_t)stream;
  float* part0 = (float*)workspace;
  float* part1 = part0 + (size_t)kMaxRowBlocks * n;
  const int gy = row_blocks(m);
  hipLaunchKernelGGL(bn_colreduce_kernel<2>, dim3((n + kColTile - 1) / kColTile, gy), dim3(kBlock), 0, s, m, n, X, ldx,
                     dY, lddy, save_mean, save_invstd, part0, part1);
  hipLaunchKernelGGL(bn_finalize_kernel<2>, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, m, n, gy, part0, part1,
                     dgamma, dbeta, nullptr, nullptr, 0.f, 0.f);
  const int64_t total = m * n;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, m, n, X,
                     ldx, dY, lddy, gamma, save_mean, save_invstd, dgamma, dbeta, relu_mask, dX, lddx);
  return check_launch("rec_batchnorm_bwd");
}

extern "C" int rec_dot_interact_fwd(int64_t batch, int32_t F, int32_t D, const float* T, int64_t ldt, float* R,
                                    int64_t ldr, void* stream) {
  REC_REQUIRE(batch >= 0 && F >= 2 && F <= kMaxFields && D > 0 && ldt >= (int64_t)F * D &&
                  ldr >= D + (int64_t)F * (F - 1) / 2, REC_EINVAL, "bad sizes");
  if (batch == 0) return REC_OK;
  REC_REQUIRE(T && R, REC_EINVAL, "null pointer argument");
  const size_t shmem = sizeof(float) * (kBlock / kWave) * (size_t)F * (D | 1);
  REC_REQUIRE(shmem <= 64 * 1024, REC_ESHAPE, "%d fields of dim %d do not fit the LDS stage", F, D);
  const int per = kBlock / kWave;
  int64_t grid = (batch + per - 1) / per;
  if (grid > 16 * kNumCU) grid = 16 * kNumCU;
  hipLaunchKernelGGL(dot_interact_fwd_kernel, dim3((unsigned)grid), dim3(kBlock), shmem, (hipStream_t)stream, batch, F,
                     D, T, ldt, R, ldr);
  return check_launch("rec_dot_interact_fwd");
}

extern "C" int rec_dot_interact_bwd(int64_t batch, int32_t F, int32_t D, const float* T, int64_t ldt, const float* dR,
                                    int64_t ldr, float* dT, int64_t lddt, void* stream) {
  REC_REQUIRE(batch >= 0 && F >= 2 && F <= kMaxFields && D > 0 && ldt >= (int64_t)F * D && lddt >= (int64_t)F * D &&
                  ldr >= D + (int64_t)F * (F - 1) / 2, REC_EINVAL, "bad sizes");
  if (batch == 0) return REC_OK;
  REC_REQUIRE(T && dR && dT, REC_EINVAL, "null pointer argument");
  const size_t shmem = sizeof(float) * (kBlock / kWave) * ((size_t)F * (D | 1) + (size_t)F * (F | 1));
  REC_REQUIRE(shmem <= 64 * 1024, REC_ESHAPE, "%d fields of dim %d do not fit the LDS stage", F, D);
  const int per = kBlock / kWave;
  int64_t grid = (batch + per - 1) / per;
  if (grid > 16 * kNumCU) grid = 16 * kNumCU;
  hipLaunchKernelGGL(dot_interact_bwd_kernel, dim3((unsigned)grid), dim3(kBlock), shmem, (hipStream_t)stream, batch, F,
                     D, T, ldt, dR, ldr, dT, lddt);
  return check_launch("rec_dot_interact_bwd");
}

extern "C" int rec_accuracy_count(int64_t n, const float* pred, const int64_t* label, int64_t* counts, void* stream) {
  REC_REQUIRE(n >= 0, REC_EINVAL, "bad sizes");
  if (n == 0) return REC_OK;
  REC_REQUIRE(pred && label && counts, REC_EINVAL, "null pointer argument");
  hipLaunchKernelGGL(accuracy_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     (hipStream_t)stream, n, pred, label, (unsigned long long*)counts);
  return check_launch("rec_accuracy_count");
}
