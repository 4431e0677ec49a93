// Loss head, AUC histogram, table fill, host-side feature hash and error plumbing (gfx950).
//
//   rec_sigmoid_logloss <- F.sigmoid (/root/reference/models/rank/deepfm/net.py:47) +
//                          log_loss/mean (deepfm/dygraph_model.py:53-58), forward value and dz
//   rec_auc_histogram   <- paddle.metric.Auc.update [EXT] (deepfm/dygraph_model.py:69-73,83-84),
//                          bucket arithmetic as tools/utils/utils_single.py:160-206 consumes it
//   rec_xxh32*          <- xxhash.xxh32(str(idx)+feat).intdigest() % hash_dim
//                          (models/rank/dnn/benchmark_reader.py:52) — XXH32 per its published spec
#include <stdarg.h>
#include <string.h>

#include <string>

#include "rec_common.h"
#include "tail_roles.h"

namespace rec {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

constexpr int kLossBlocks = 1024;

__global__ __launch_bounds__(kBlock) void sigmoid_logloss_kernel(
    int64_t B, float invB, const float* __restrict__ y1, const float* __restrict__ y2,
    const float* __restrict__ y3, const int64_t* __restrict__ label, float eps, float clip_lo,
    float clip_hi, float* __restrict__ pred, float* __restrict__ dz, float* __restrict__ partial) {
  __shared__ float red[kBlock / kWave];
  float local = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < B;
       i += (int64_t)gridDim.x * kBlock) {
    float z = y1[i];
    if (y2) z += y2[i];
    if (y3) z += y3[i];
    // paddle.clip(y, min, max) in front of the sigmoid (slot_dnn/net.py:84); its gradient is 1 strictly
    // inside the interval, 0 elsewhere [EXT ClipGradFunctor]
    const bool clipped = clip_lo < clip_hi;
    const float open = (!clipped || (z > clip_lo && z < clip_hi)) ? 1.f : 0.f;
    if (clipped) z = fminf(fmaxf(z, clip_lo), clip_hi);
    const float p = 1.f / (1.f + expf(-z));
    const float t = (float)label[i];
    const float cost = -t * logf(p + eps) - (1.f - t) * logf(1.f - p + eps);
    local += cost;
    if (pred) pred[i] = p;
    if (dz) dz[i] = (-t / (p + eps) + (1.f - t) / (1.f - p + eps)) * invB * (p * (1.f - p)) * open;
  }
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) local += __shfl_xor(local, o, kWave);
  if (threadIdx.x % kWave == 0) red[threadIdx.x / kWave] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < kBlock / kWave; ++w) t += red[w];
    partial[blockIdx.x] = t;
  }
}

__global__ void fold_loss_kernel(const float* __restrict__ partial, int n, float invB,
                                 float* __restrict__ out) {
  __shared__ float red[kBlock];
  float t = 0.f;
  for (int i = threadIdx.x; i < n; i += kBlock) t += partial[i];
  red[threadIdx.x] = t;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0] * invB;
}

// binary_cross_entropy_with_logits(reduction='mean') [EXT] (din/dygraph_model.py:58-61) and its gradient:
//   cost = max(z,0) - z*t + log1p(exp(-|z|));   d mean / dz = (sigmoid(z) - t) / B
__global__ __launch_bounds__(kBlock) void bce_logits_kernel(int64_t B, float invB,
                                                            const float* __restrict__ z,
                                                            const float* __restrict__ label,
                                                            float* __restrict__ pred, float* __restrict__ dz,
                                                            float* __restrict__ partial, float* __restrict__ loss_one) {
  __shared__ float red[kBlock / kWave];
  float local = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < B; i += (int64_t)gridDim.x * kBlock) {
    const float zi = z[i], t = label[i];
    local += fmaxf(zi, 0.f) - zi * t + log1pf(expf(-fabsf(zi)));
    const float p = 1.f / (1.f + expf(-zi));
    if (pred) pred[i] = p;
    if (dz) dz[i] = (p - t) * invB;
  }
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) local += __shfl_xor(local, o, kWave);
  if (threadIdx.x % kWave == 0) red[threadIdx.x / kWave] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < kBlock / kWave; ++w) t += red[w];
    partial[blockIdx.x] = t;
    // a launch of ONE block (batch <= 256: din/config.yaml's 32) is its own fold: fold_loss_kernel over one partial is
    // (0 + t + 0 + ...) * invB — the same float, one launch less on a launch-bound step
    if (loss_one) loss_one[0] = (0.f + t) * invB;
  }
}

// per-block LDS histograms (int32), flushed with 64-bit integer atomics: exact for any order
__global__ __launch_bounds__(kBlock) void auc_hist_kernel(
    int64_t B, const float* __restrict__ pred, const int64_t* __restrict__ label, int T,
    unsigned long long* __restrict__ pos, unsigned long long* __restrict__ neg) {
  extern __shared__ int sh[];  // [2][T+1]
  const int nb = T + 1;
  for (int i = threadIdx.x; i < 2 * nb; i += kBlock) sh[i] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < B;
       i += (int64_t)gridDim.x * kBlock) {
    int bucket = (int)(pred[i] * (float)T);
    bucket = bucket < 0 ? 0 : (bucket > T ? T : bucket);
    atomicAdd(&sh[(label[i] != 0 ? 0 : nb) + bucket], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nb; i += kBlock) {
    if (sh[i]) atomicAdd(&pos[i], (unsigned long long)sh[i]);
    if (sh[nb + i]) atomicAdd(&neg[i], (unsigned long long)sh[nb + i]);
  }
}

// ---------------------------------------------------------------------------------- the CTR head in one pass
// The tail of every CTR tower here is  h [B,n] (behind a ReLU) -> Linear(n, 1) -> logit = y1 + y2 + y_dnn -> sigmoid ->
// log_loss -> mean  (deepfm/net.py:169-174 + dygraph_model.py:76-85), and its backward needs h again: dx = dz w (under
// the ReLU mask of h), dW = h^T dz, db = sum dz.  As separate calls that is a GEMV pass over h (26 us at B 65536), two
// loss launches, and a second pass over h for the backward (35 us) — five launches on the step's critical path.  Here a
// wave keeps a row of h in registers: dot product, loss and dz, dx written, dW accumulated per lane — ONE read of h.
// Lane l owns columns 4l..4l+3 and 256+4l..+3 (n <= 512, n % 4 == 0); a wave walks rows wave, wave + W, ... in ascending
// order, two rows in flight; per-wave partials fold in LDS in wave order, one partial row [n + 2] per block (dW | db |
// sum of costs), folded across blocks by ctr_head_fold_kernel in ascending block order: deterministic.
constexpr int kCtrHeadMaxBlocks = kNumCU * 4;
__global__ __launch_bounds__(kBlock) void ctr_head_kernel(
    int64_t B, int N, float invB, const float* __restrict__ act, int64_t ld_act, const float* __restrict__ w,
    const float* __restrict__ bias, const float* __restrict__ y1, const float* __restrict__ y2,
    const int64_t* __restrict__ label, float eps, float clip_lo, float clip_hi, int relu, float* __restrict__ y_out,
    float* __restrict__ pred, float* __restrict__ dz_out, float* __restrict__ dx, int64_t ld_dx,
    float* __restrict__ partial) {
  extern __shared__ float ctr_red[];                       // [waves][N + 2]
  constexpr int W = kBlock / kWave;
  const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
  const int c0 = lane * 4, c1 = 256 + lane * 4;
  const bool ok0 = c0 < N, ok1 = c1 < N;
  float w0[4] = {0.f, 0.f, 0.f, 0.f}, w1[4] = {0.f, 0.f, 0.f, 0.f};
  if (ok0) vload<4>(w0, w + c0);
  if (ok1) vload<4>(w1, w + c1);
  const float b0 = bias ? bias[0] : 0.f;
  const bool clipped = clip_lo < clip_hi;
  float g0[4] = {0.f, 0.f, 0.f, 0.f}, g1[4] = {0.f, 0.f, 0.f, 0.f}, gb = 0.f, cost_sum = 0.f;
  const int64_t nw = (int64_t)gridDim.x * W;
  for (int64_t r = (int64_t)blockIdx.x * W + wave; r < B; r += 2 * nw) {
    const int64_t rr[2] = {r, r + nw};
    float a0[2][4], a1[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int v = 0; v < 4; ++v) a0[u][v] = a1[u][v] = 0.f;
      if (rr[u] < B) {
        if (ok0) vload_nt<4>(a0[u], act + rr[u] * ld_act + c0);
        if (ok1) vload_nt<4>(a1[u], act + rr[u] * ld_act + c1);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (rr[u] >= B) continue;                             // wave-uniform
      const int64_t i = rr[u];
      float acc = 0.f;
      acc += a0[u][0] * w0[0] + a0[u][1] * w0[1] + a0[u][2] * w0[2] + a0[u][3] * w0[3];
      acc += a1[u][0] * w1[0] + a1[u][1] * w1[1] + a1[u][2] * w1[2] + a1[u][3] * w1[3];
#pragma unroll
      for (int o = kWave / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, kWave);
      const float y = acc + b0;
      float z = y1 ? y1[i] : 0.f;
      if (y2) z += y2[i];
      z = y1 ? z + y : y;
      const float open = (!clipped || (z > clip_lo && z < clip_hi)) ? 1.f : 0.f;
      if (clipped) z = fminf(fmaxf(z, clip_lo), clip_hi);
      const float p = 1.f / (1.f + expf(-z));
      const float t = (float)label[i];
      cost_sum += -t * logf(p + eps) - (1.f - t) * logf(1.f - p + eps);
      const float dz = (-t / (p + eps) + (1.f - t) / (1.f - p + eps)) * invB * (p * (1.f - p)) * open;
      gb += dz;
      if (lane == 0) {
        if (y_out) y_out[i] = y;
        pred[i] = p;
        dz_out[i] = dz;
      }
      float o0[4], o1[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        g0[v] += a0[u][v] * dz;
        g1[v] += a1[u][v] * dz;
        o0[v] = (!relu || a0[u][v] > 0.f) ? dz * w0[v] : 0.f;
        o1[v] = (!relu || a1[u][v] > 0.f) ? dz * w1[v] : 0.f;
      }
      if (ok0) vstore<4>(dx + i * ld_dx + c0, o0);
      if (ok1) vstore<4>(dx + i * ld_dx + c1, o1);
    }
  }
  float* mine = ctr_red + wave * (N + 2);
  if (ok0) { mine[c0] = g0[0]; mine[c0 + 1] = g0[1]; mine[c0 + 2] = g0[2]; mine[c0 + 3] = g0[3]; }
  if (ok1) { mine[c1] = g1[0]; mine[c1 + 1] = g1[1]; mine[c1 + 2] = g1[2]; mine[c1 + 3] = g1[3]; }
  if (lane == 0) { mine[N] = gb; mine[N + 1] = cost_sum; }
  __syncthreads();
  float* prow = partial + (int64_t)blockIdx.x * (N + 2);
  for (int j = threadIdx.x; j < N + 2; j += kBlock) {
    float s = ctr_red[j];
#pragma unroll
    for (int q = 1; q < W; ++q) s += ctr_red[q * (N + 2) + j];
    prow[j] = s;
  }
}

// column j of the [nblk][n2] partial rows, blocks in ascending order.  A block owns 16 columns, thread (c, q) the rows
// q, q + 16, ... of column c on four interleaved chains (loads in flight instead of one dependent chain per column: the
// first version of this kernel summed 512 rows per thread one after the other and cost the step 130 us), then the 16
// row groups fold in ascending q.  Columns < n2 - 2 -> dw, n2 - 2 -> db, n2 - 1 -> loss = sum of costs * invB.
__global__ __launch_bounds__(kBlock) void ctr_head_fold_kernel(int nblk, int n2, const float* __restrict__ partial,
                                                               float invB, float* __restrict__ dw,
                                                               float* __restrict__ db, float* __restrict__ loss) {
  ctr_head_fold_role<false>(blockIdx.x, threadIdx.x, nblk, n2, partial, invB, dw, db, loss, DenseAdam{}, 0, 0);   // tail_roles.h
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__global__ void fill_uniform_kernel(int64_t n, float* __restrict__ buf, float lo, float hi,
                                    uint64_t seed) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t r = splitmix64(seed ^ splitmix64((uint64_t)i));
    const float u = (float)(r >> 40) * (1.0f / 16777216.0f);  // 24 random bits -> [0,1)
    buf[i] = lo + (hi - lo) * u;
  }
}

}  // namespace rec

using namespace rec;

extern "C" const char* rec_last_error(void) { return rec::g_err; }
extern "C" int rec_version(void) { return 100; }

extern "C" int rec_logloss_workspace_bytes(int64_t batch, size_t* bytes) {
  REC_REQUIRE(bytes && batch >= 0, REC_EINVAL, "bad arguments");
  *bytes = kLossBlocks * sizeof(float);
  return REC_OK;
}

extern "C" int rec_sigmoid_logloss(int64_t batch, int64_t mean_over, const float* y1,
                                   const float* y2, const float* y_dnn, const int64_t* label,
                                   float eps, float clip_lo, float clip_hi,
                                   float* pred, float* dz, float* loss_out, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  REC_REQUIRE(batch > 0 && mean_over >= 0 && y1 && label && loss_out, REC_EINVAL,
              "bad arguments");
  const float inv = 1.f / (float)(mean_over > 0 ? mean_over : batch);
  REC_REQUIRE(workspace && workspace_bytes >= kLossBlocks * sizeof(float), REC_EWORKSPACE,
              "workspace too small");
  int64_t grid = (batch + kBlock - 1) / kBlock;
  if (grid > kLossBlocks) grid = kLossBlocks;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sigmoid_logloss_kernel, dim3((unsigned)grid), dim3(kBlock), 0, st, batch, inv,
                     y1, y2, y_dnn, label, eps, clip_lo, clip_hi, pred, dz, (float*)workspace);
  hipLaunchKernelGGL(fold_loss_kernel, dim3(1), dim3(kBlock), 0, st, (const float*)workspace,
                     (int)grid, inv, loss_out);
  return check_launch("rec_sigmoid_logloss");
}

static int64_t ctr_head_grid(int64_t batch) {
  // ~4 rows per wave at least; the launch-bound batches (the reference's 512) one pass of two rows per wave
  const int per = (kBlock / kWave) * (batch <= 4096 ? 2 : 4);
  int64_t grid = (batch + per - 1) / per;
  if (grid > kCtrHeadMaxBlocks) grid = kCtrHeadMaxBlocks;
  return grid < 1 ? 1 : grid;
}

extern "C" int rec_ctr_head_workspace_bytes(int64_t batch, int32_t n, size_t* bytes) {
  REC_REQUIRE(bytes && batch >= 0 && n > 0, REC_EINVAL, "bad arguments");
  *bytes = align_up((size_t)ctr_head_grid(batch) * (n + 2) * sizeof(float), 256);
  return REC_OK;
}

static int ctr_head_impl(int64_t batch, int32_t n, int64_t mean_over, const float* act, int64_t ld_act, const float* w,
                         const float* bias, const float* y1, const float* y2, const int64_t* label, float eps,
                         float clip_lo, float clip_hi, int32_t relu, float* y_dnn, float* pred, float* dz,
                         float* loss_out, float* dx, int64_t ld_dx, float* dw, float* db, void* workspace,
                         size_t workspace_bytes, void* stream, bool fold, int* nblk, float* invB) {
  REC_REQUIRE(batch > 0 && mean_over >= 0, REC_EINVAL, "bad batch");
  REC_REQUIRE(n > 0 && n % 4 == 0 && n <= 512, REC_ESHAPE, "rec_ctr_head_fwd_bwd: n must be a multiple of 4 and <= 512");
  REC_REQUIRE(act && w && label && pred && dz && dx && (!fold || (loss_out && dw && db)), REC_EINVAL,
              "null pointer argument");
  REC_REQUIRE(!y2 || y1, REC_EINVAL, "y2 without y1");
  REC_REQUIRE(ld_act >= n && ld_dx >= n && ld_act % 4 == 0 && ld_dx % 4 == 0 && ((uintptr_t)act) % 16 == 0 &&
                  ((uintptr_t)dx) % 16 == 0 && ((uintptr_t)w) % 16 == 0, REC_ESHAPE,
              "rec_ctr_head_fwd_bwd: act / dx / w must be 16-byte aligned with row strides that are multiples of 4");
  size_t need = 0;
  rec_ctr_head_workspace_bytes(batch, n, &need);
  REC_REQUIRE(workspace && workspace_bytes >= need, REC_EWORKSPACE, "workspace %zu < %zu", workspace_bytes, need);
  const float inv = 1.f / (float)(mean_over > 0 ? mean_over : batch);
  const int64_t grid = ctr_head_grid(batch);
  hipStream_t st = (hipStream_t)stream;
  const size_t shmem = (size_t)(kBlock / kWave) * (n + 2) * sizeof(float);
  hipLaunchKernelGGL(ctr_head_kernel, dim3((unsigned)grid), dim3(kBlock), shmem, st, batch, n, inv, act, ld_act, w, bias, y1,
                     y2, label, eps, clip_lo, clip_hi, relu, y_dnn, pred, dz, dx, ld_dx, (float*)workspace);
  if (fold)
    hipLaunchKernelGGL(ctr_head_fold_kernel, dim3((unsigned)((n + 2 + 15) / 16)), dim3(kBlock), 0, st, (int)grid,
                       n + 2, (const float*)workspace, inv, dw, db, loss_out);
  if (nblk) *nblk = (int)grid;
  if (invB) *invB = inv;
  return check_launch("rec_ctr_head_fwd_bwd");
}

extern "C" int rec_ctr_head_fwd_bwd(int64_t batch, int32_t n, int64_t mean_over, const float* act, int64_t ld_act,
                                    const float* w, const float* bias, const float* y1, const float* y2,
                                    const int64_t* label, float eps, float clip_lo, float clip_hi, int32_t relu,
                                    float* y_dnn, float* pred, float* dz, float* loss_out, float* dx, int64_t ld_dx,
                                    float* dw, float* db, void* workspace, size_t workspace_bytes, void* stream) {
  return ctr_head_impl(batch, n, mean_over, act, ld_act, w, bias, y1, y2, label, eps, clip_lo, clip_hi, relu, y_dnn, pred,
                       dz, loss_out, dx, ld_dx, dw, db, workspace, workspace_bytes, stream, true, nullptr, nullptr);
}

// the head without its fold launch (tail_roles.h: the fold rides in the step's tail launch)
int rec::ctr_head_fwd_bwd_partial(int64_t batch, int32_t n, int64_t mean_over, const float* act, int64_t ld_act,
                                  const float* w, const float* bias, const float* y1, const float* y2,
                                  const int64_t* label, float eps, float clip_lo, float clip_hi, int32_t relu,
                                  float* y_dnn, float* pred, float* dz, float* dx, int64_t ld_dx, void* workspace,
                                  size_t workspace_bytes, void* stream, int* nblk, float* invB) {
  return ctr_head_impl(batch, n, mean_over, act, ld_act, w, bias, y1, y2, label, eps, clip_lo, clip_hi, relu, y_dnn, pred,
                       dz, nullptr, dx, ld_dx, nullptr, nullptr, workspace, workspace_bytes, stream, false, nblk, invB);
}

extern "C" int rec_bce_with_logits(int64_t batch, int64_t mean_over, const float* logit,
                                   const float* label, float* pred, float* dz, float* loss_out,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  REC_REQUIRE(batch > 0 && mean_over >= 0 && logit && label && loss_out, REC_EINVAL, "bad arguments");
  REC_REQUIRE(workspace && workspace_bytes >= kLossBlocks * sizeof(float), REC_EWORKSPACE,
              "workspace too small");
  const float inv = 1.f / (float)(mean_over > 0 ? mean_over : batch);
  int64_t grid = (batch + kBlock - 1) / kBlock;
  if (grid > kLossBlocks) grid = kLossBlocks;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(bce_logits_kernel, dim3((unsigned)grid), dim3(kBlock), 0, st, batch, inv, logit, label,
                     pred, dz, (float*)workspace, grid == 1 ? loss_out : (float*)nullptr);
  if (grid > 1)
    hipLaunchKernelGGL(fold_loss_kernel, dim3(1), dim3(kBlock), 0, st, (const float*)workspace, (int)grid, inv,
                       loss_out);
  return check_launch("rec_bce_with_logits");
}

extern "C" int rec_auc_histogram(int64_t batch, const float* pred, const int64_t* label,
                                 int32_t num_thresholds, int64_t* stat_pos, int64_t* stat_neg,
                                 void* stream) {
  REC_REQUIRE(batch >= 0 && num_thresholds > 0 && num_thresholds < 16000, REC_EINVAL,
              "bad arguments");
  if (batch == 0) return REC_OK;
  REC_REQUIRE(pred && label && stat_pos && stat_neg, REC_EINVAL, "null pointer argument");
  int64_t grid = (batch + kBlock * 16 - 1) / (kBlock * 16);
  if (grid > kNumCU * 2) grid = kNumCU * 2;
  const size_t shmem = 2 * (size_t)(num_thresholds + 1) * sizeof(int);
  hipLaunchKernelGGL(auc_hist_kernel, dim3((unsigned)grid), dim3(kBlock), shmem,
                     (hipStream_t)stream, batch, pred, label, num_thresholds,
                     (unsigned long long*)stat_pos, (unsigned long long*)stat_neg);
  return check_launch("rec_auc_histogram");
}

__global__ void spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

extern "C" int rec_stream_spin(int32_t micros, void* stream) {
  REC_REQUIRE(micros >= 0 && micros <= 1000000, REC_EINVAL, "micros out of range");
  int dev = 0, khz = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0)
    khz = 100000;   // gfx9 constant 100 MHz counter
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream,
                     (long long)micros * khz / 1000);
  return check_launch("rec_stream_spin");
}

extern "C" int rec_copy_async(void* dst, const void* src, size_t bytes, void* stream) {
  REC_REQUIRE(bytes == 0 || (dst && src), REC_EINVAL, "null pointer argument");
  if (bytes == 0) return REC_OK;
  REC_REQUIRE(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess, REC_EHIP,
              "hipMemcpyAsync failed");
  return REC_OK;
}

extern "C" int rec_copy_2d_async(void* dst, size_t dst_pitch_bytes, const void* src, size_t src_pitch_bytes,
                                 size_t width_bytes, size_t rows, void* stream) {
  if (width_bytes == 0 || rows == 0) return REC_OK;
  REC_REQUIRE(dst && src, REC_EINVAL, "null pointer argument");
  REC_REQUIRE(dst_pitch_bytes >= width_bytes && src_pitch_bytes >= width_bytes, REC_ESHAPE, "pitch < width");
  REC_REQUIRE(hipMemcpy2DAsync(dst, dst_pitch_bytes, src, src_pitch_bytes, width_bytes, rows, hipMemcpyDeviceToDevice,
                               (hipStream_t)stream) == hipSuccess, REC_EHIP, "hipMemcpy2DAsync failed");
  return REC_OK;
}

namespace rec {
// 32 x 32 tile through LDS (+1 padding column: conflict-free column reads); 256 threads, 4 rows each.
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            int64_t rows, int64_t cols) {
  __shared__ float tile[32][33];
  const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) {
    int64_t r = r0 + k, c = c0 + tx;
    if (r < rows && c < cols) tile[k][tx] = in[r * cols + c];
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    int64_t c = c0 + k, r = r0 + tx;
    if (r < rows && c < cols) out[c * rows + r] = tile[tx][k];
  }
}
}  // namespace rec

namespace rec {
__global__ __launch_bounds__(256) void cast_f32_i64_kernel(int64_t n, const float* __restrict__ src, int64_t stride,
                                                           int64_t* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = (int64_t)llrintf(src[i * stride]);
}
}  // namespace rec

extern "C" int rec_cast_f32_i64(int64_t n, const float* src, int64_t src_stride, int64_t* dst, void* stream) {
  REC_REQUIRE(n >= 0 && src_stride >= 1, REC_EINVAL, "bad sizes");
  if (n == 0) return REC_OK;
  REC_REQUIRE(src && dst, REC_EINVAL, "null pointer argument");
  REC_REQUIRE((n + 255) / 256 < (1ll << 31), REC_ESHAPE, "n too large");
  rec::cast_f32_i64_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, (hipStream_t)stream>>>(n, src, src_stride, dst);
  return rec::check_launch("rec_cast_f32_i64");
}

extern "C" int rec_transpose_f32(int64_t rows, int64_t cols, const float* in, float* out, void* stream) {
  REC_REQUIRE(rows >= 0 && cols >= 0, REC_ESHAPE, "negative shape");
  if (rows == 0 || cols == 0) return REC_OK;
  REC_REQUIRE(in && out, REC_EINVAL, "null pointer argument");
  REC_REQUIRE((rows + 31) / 32 <= 65535, REC_ESHAPE, "rows too large for one launch");
  dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
  rec::transpose_f32_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(in, out, rows, cols);
  return rec::check_launch("rec_transpose_f32");
}

extern "C" int rec_stream_create_cu_range(int32_t cu_begin, int32_t cu_end, void** stream) {
  REC_REQUIRE(stream && cu_begin >= 0 && cu_end > cu_begin && cu_end <= 1024, REC_EINVAL, "bad CU range");
  uint32_t mask[32] = {0};
  for (int i = cu_begin; i < cu_end; ++i) mask[i / 32] |= 1u << (i % 32);
  hipStream_t s = nullptr;
  const hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)((cu_end + 31) / 32), mask);
  if (e != hipSuccess) {
    set_error("hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e));
    return REC_EHIP;
  }
  *stream = (void*)s;
  return REC_OK;
}

extern "C" int rec_stream_create_cu_stride(int32_t first, int32_t stride, int32_t cu_total, void** stream) {
  REC_REQUIRE(stream && first >= 0 && stride >= 1 && cu_total > first && cu_total <= 1024, REC_EINVAL, "bad CU stride");
  uint32_t mask[32] = {0};
  for (int i = first; i < cu_total; i += stride) mask[i / 32] |= 1u << (i % 32);
  hipStream_t s = nullptr;
  const hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)((cu_total + 31) / 32), mask);
  if (e != hipSuccess) {
    set_error("hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e));
    return REC_EHIP;
  }
  *stream = (void*)s;
  return REC_OK;
}

extern "C" int rec_stream_destroy(void* stream) {
  if (!stream) return REC_OK;
  const hipError_t e = hipStreamDestroy((hipStream_t)stream);
  if (e != hipSuccess) {
    set_error("hipStreamDestroy: %s", hipGetErrorString(e));
    return REC_EHIP;
  }
  return REC_OK;
}

extern "C" int rec_fill_uniform(int64_t n, float* buf, float lo, float hi, uint64_t seed,
                                void* stream) {
  REC_REQUIRE(n >= 0 && (n == 0 || buf), REC_EINVAL, "bad arguments");
  if (n == 0) return REC_OK;
  hipLaunchKernelGGL(fill_uniform_kernel, dim3(kNumCU * 8), dim3(kBlock), 0, (hipStream_t)stream,
                     n, buf, lo, hi, seed);
  return check_launch("rec_fill_uniform");
}

// ------------------------------------------------------------------ XXH32 (host), published spec
static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

extern "C" uint32_t rec_xxh32(const void* bytes, size_t len, uint32_t seed) {
  const uint32_t P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u,
                 P5 = 374761393u;
  const uint8_t* p = (const uint8_t*)bytes;
  const uint8_t* const end = p + len;
  uint32_t h;
  if (len >= 16) {
    uint32_t v[4] = {seed + P1 + P2, seed + P2, seed, seed - P1};
    for (; p + 16 <= end; p += 16) {
      uint32_t w[4];
      memcpy(w, p, 16);
      for (int i = 0; i < 4; ++i) v[i] = rotl32(v[i] + w[i] * P2, 13) * P1;
    }
    h = rotl32(v[0], 1) + rotl32(v[1], 7) + rotl32(v[2], 12) + rotl32(v[3], 18);
  } else {
    h = seed + P5;
  }
  h += (uint32_t)len;
  for (; p + 4 <= end; p += 4) {
    uint32_t w;
    memcpy(&w, p, 4);
    h = rotl32(h + w * P3, 17) * P4;
  }
  for (; p < end; ++p) h = rotl32(h + (*p) * P5, 11) * P1;
  h ^= h >> 15;
  h *= P2;
  h ^= h >> 13;
  h *= P3;
  h ^= h >> 16;
  return h;
}

extern "C" int rec_xxh32_hash_mod(const char* const* strings, const int32_t* field_idx, int64_t n,
                                  uint32_t hash_dim, int64_t* out) {
  REC_REQUIRE(n >= 0 && hash_dim > 0 && (n == 0 || (strings && field_idx && out)), REC_EINVAL,
              "bad arguments");
  std::string buf;
  for (int64_t i = 0; i < n; ++i) {
    buf = std::to_string(field_idx[i]);  // str(idx) + features[idx]
    buf += strings[i];
    out[i] = (int64_t)(rec_xxh32(buf.data(), buf.size(), 0) % hash_dim);
  }
  return REC_OK;
}
