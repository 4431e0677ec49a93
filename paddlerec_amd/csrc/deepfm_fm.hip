// DeepFM: embedding lookup + FM first/second order, forward and backward, fused (gfx950).
//
// Replaces the Paddle op chain of /root/reference/models/rank/deepfm/net.py:105-139
//   concat -> Embedding x2 -> multiply/unsqueeze/sum -> concat -> sum/square/sum/sub/sum/scale
// (6+ kernels each re-streaming feat_embeddings [B,39,D]) with ONE pass:
//   ids -> row gathers (coalesced 16-B lanes, all S gathers of a sample in flight at once)
//       -> running sum / sum-of-squares in registers -> feat written once -> y1,y2.
//
// Work decomposition (HBM-bound, no MFMA): a "row group" of LANES lanes owns one sample; each lane
// owns VEC consecutive floats of the embedding dimension (D=16 -> 4 lanes x float4 = one 64-B row
// per group load; a wave gathers 16 rows per instruction).  The FM reductions over fields are
// lane-local; only y1/y2 need a log2(LANES)-step cross-lane sum.
#include "rec_common.h"

namespace rec {

// ------------------------------------------------------------------------------------------ fwd
template <int VEC, int LANES, int CH>
__global__ __launch_bounds__(kBlock) void fm_fwd_kernel(
    int64_t B, int S, int Dn, int D, int stride, int64_t N, int64_t pad,
    const int64_t* __restrict__ ids, const float* __restrict__ dense, const float* __restrict__ W,
    const float* __restrict__ W1, const float* __restrict__ dense_w,
    const float* __restrict__ dense_w_one, const int64_t* __restrict__ slot_off,
    float* __restrict__ y1, float* __restrict__ y2, float* __restrict__ feat,
    float* __restrict__ sum_emb, int32_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_dw = smem;            // [Dn*D]
  float* s_dw1 = smem + Dn * D;  // [Dn]
  for (int i = threadIdx.x; i < Dn * D; i += kBlock) s_dw[i] = dense_w[i];
  for (int i = threadIdx.x; i < Dn; i += kBlock) s_dw1[i] = dense_w_one[i];
  __syncthreads();

  const int lg = threadIdx.x % LANES;
  const int64_t b = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / LANES;
  const bool active = b < B;
  const int d0 = lg * VEC;
  const bool dvalid = active && d0 < D;
  const int F = S + Dn;

  float s[VEC], q[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) s[v] = q[v] = 0.f;
  float first = 0.f;
  const int64_t* idp = ids + b * S;
  float* fb = feat + (b * F) * (int64_t)D + d0;

  for (int s0 = 0; s0 < S; s0 += CH) {
    int64_t row[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int si = s0 + c;
      row[c] = -1;
      if (active && si < S) {
        const int64_t id = idp[si];
        if (id != pad || pad < 0) {
          const int64_t r = slot_off ? id + slot_off[si] : id;
          if (r >= 0 && r < N) {
            row[c] = r;
          } else if (lg == 0) {
            atomicOr(status, REC_FLAG_INDEX_OOB);
          }
        }
      }
    }
    float e[CH][VEC];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (dvalid && row[c] >= 0) {
        vload<VEC>(e[c], W + row[c] * stride + d0);
      } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) e[c][v] = 0.f;
      }
    }
    // first-order weights: the S scalar gathers of a sample are spread over its LANES lanes
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (row[c] >= 0 && ((s0 + c) % LANES) == lg) first += W1[row[c]];
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int si = s0 + c;
      if (dvalid && si < S) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          s[v] += e[c][v];
          q[v] += e[c][v] * e[c][v];
        }
        vstore<VEC>(fb + (int64_t)si * D, e[c]);
      }
    }
  }
  // dense fields: feat[b,S+j,:] = dense[b,j] * dense_w[j,:]   (net.py:118-121)
  if (active) {
    for (int j = 0; j < Dn; ++j) {
      const float x = dense[b * Dn + j];
      if ((j % LANES) == lg) first += x * s_dw1[j];
      if (dvalid) {
        float e[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          e[v] = x * s_dw[j * D + d0 + v];
          s[v] += e[v];
          q[v] += e[v] * e[v];
        }
        vstore<VEC>(fb + (int64_t)(S + j) * D, e);
      }
    }
  }
  float part = 0.f;
#pragma unroll
  for (int v = 0; v < VEC; ++v) part += s[v] * s[v] - q[v];
  if (!dvalid) part = 0.f;
  if (dvalid && sum_emb) vstore<VEC>(sum_emb + b * D + d0, s);
  const float tot2 = group_sum<LANES>(part);
  const float tot1 = group_sum<LANES>(active ? first : 0.f);
  if (active && lg == 0) {
    y1[b] = tot1;
    y2[b] = 0.5f * tot2;
  }
}

// ------------------------------------------------------------------------------------------ bwd
constexpr int kDnMax = 16;
constexpr int kBwdCH = 8;

template <int VEC, int LANES>
__global__ __launch_bounds__(kBlock) void fm_bwd_kernel(
    int64_t B, int S, int Dn, int D, const float* __restrict__ dense,
    const float* __restrict__ feat, const float* __restrict__ sum_emb,
    const float* __restrict__ dfeat, const float* __restrict__ dy1, const float* __restrict__ dy2,
    float* __restrict__ row_grad, float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [waves][Dn*D + Dn]
  const int lg = threadIdx.x % LANES;
  const int d0 = lg * VEC;
  const bool dvalid = d0 < D;
  const int F = S + Dn;
  constexpr int GPB = kBlock / LANES;  // sample groups per block
  const int64_t gstride = (int64_t)gridDim.x * GPB;

  float acc[kDnMax][VEC];
  float acc1[kDnMax];
#pragma unroll
  for (int j = 0; j < kDnMax; ++j) {
    acc1[j] = 0.f;
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[j][v] = 0.f;
  }

  for (int64_t b = (int64_t)blockIdx.x * GPB + threadIdx.x / LANES; b < B; b += gstride) {
    if (!dvalid) continue;
    float sb[VEC];
    vload<VEC>(sb, sum_emb + b * D + d0);
    const float g1 = dy1[b], g2 = dy2[b];
    const float* fb = feat + (b * F) * (int64_t)D + d0;
    const float* gb = dfeat + (b * F) * (int64_t)D + d0;
    float* rg = row_grad + (b * S) * (int64_t)D + d0;
    for (int f0 = 0; f0 < S; f0 += kBwdCH) {
      float e[kBwdCH][VEC], g[kBwdCH][VEC];
#pragma unroll
      for (int c = 0; c < kBwdCH; ++c) {
        if (f0 + c < S) {
          vload<VEC>(e[c], fb + (int64_t)(f0 + c) * D);
          vload<VEC>(g[c], gb + (int64_t)(f0 + c) * D);
        }
      }
#pragma unroll
      for (int c = 0; c < kBwdCH; ++c) {
        if (f0 + c < S) {
          float de[VEC];
#pragma unroll
          for (int v = 0; v < VEC; ++v) de[v] = g[c][v] + g2 * (sb[v] - e[c][v]);
          vstore<VEC>(rg + (int64_t)(f0 + c) * D, de);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < kDnMax; ++j) {
      if (j < Dn) {
        const float x = dense[b * Dn + j];
        float e[VEC], g[VEC];
        vload<VEC>(e, fb + (int64_t)(S + j) * D);
        vload<VEC>(g, gb + (int64_t)(S + j) * D);
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[j][v] += x * (g[v] + g2 * (sb[v] - e[v]));
        if (lg == 0) acc1[j] += g1 * x;
      }
    }
  }

  // fold the 64/LANES sample groups of a wave, then the waves of the block, in a fixed order
#pragma unroll
  for (int j = 0; j < kDnMax; ++j) {
#pragma unroll
    for (int o = LANES; o < kWave; o <<= 1) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[j][v] += __shfl_xor(acc[j][v], o, kWave);
      acc1[j] += __shfl_xor(acc1[j], o, kWave);
    }
  }
  const int K = Dn * D + Dn;
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  float* sw = smem + wave * K;
  if (lane < LANES && dvalid) {
#pragma unroll
    for (int j = 0; j < kDnMax; ++j) {
      if (j < Dn) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) sw[j * D + d0 + v] = acc[j][v];
        if (lane == 0) sw[Dn * D + j] = acc1[j];
      }
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += kBlock) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) t += smem[w * K + k];
    partial[(int64_t)blockIdx.x * K + k] = t;
  }
}

// out[k] = sum_blocks partial[blk][k], blocks folded in index order (deterministic)
__global__ void fold_partials_kernel(const float* __restrict__ partial, int nblk, int K, int split,
                                     float* __restrict__ out0, float* __restrict__ out1) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  float t = 0.f;
  for (int i = 0; i < nblk; ++i) t += partial[(int64_t)i * K + k];
  if (k < split) out0[k] = t; else out1[k - split] = t;
}

static int bwd_grid(int64_t B, int lanes) {
  const int gpb = kBlock / lanes;
  int64_t need = (B + gpb - 1) / gpb;
  const int64_t cap = kNumCU * 4;
  return (int)(need < cap ? (need > 0 ? need : 1) : cap);
}

static int check_desc(const rec_deepfm_desc* d) {
  REC_REQUIRE(d, REC_EINVAL, "desc is NULL");
  REC_REQUIRE(d->batch >= 0 && d->num_slots > 0 && d->num_dense >= 0 && d->emb_dim > 0,
              REC_EINVAL, "bad sizes B=%lld S=%d Dn=%d D=%d", (long long)d->batch, d->num_slots,
              d->num_dense, d->emb_dim);
  REC_REQUIRE(d->num_dense <= kDnMax, REC_ESHAPE, "num_dense %d > %d", d->num_dense, kDnMax);
  REC_REQUIRE(d->row_stride >= d->emb_dim, REC_EINVAL, "row_stride %d < emb_dim %d",
              d->row_stride, d->emb_dim);
  REC_REQUIRE(d->num_rows > 0, REC_EINVAL, "num_rows must be > 0");
  return REC_OK;
}

}  // namespace rec

using namespace rec;

extern "C" int rec_deepfm_fm_fwd(const rec_deepfm_desc* desc, const int64_t* ids,
                                 const float* dense, const float* W, const float* W1,
                                 const float* dense_w, const float* dense_w_one,
                                 const int64_t* slot_offset, float* y1, float* y2, float* feat,
                                 float* sum_emb, int32_t* status, void* stream) {
  if (int rc = check_desc(desc)) return rc;
  REC_REQUIRE(ids && W && W1 && y1 && y2 && feat && status, REC_EINVAL, "null pointer argument");
  REC_REQUIRE(desc->num_dense == 0 || (dense && dense_w && dense_w_one), REC_EINVAL,
              "dense inputs missing");
  if (desc->batch == 0) return REC_OK;
  const int S = desc->num_slots, Dn = desc->num_dense, D = desc->emb_dim;
  const size_t shmem = (size_t)(Dn * D + Dn) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  return dispatch_row_shape(D, desc->row_stride, [&](auto vec, auto lanes) -> int {
    constexpr int VEC = decltype(vec)::value, LANES = decltype(lanes)::value;
    const int gpb = kBlock / LANES;
    const int64_t grid = (desc->batch + gpb - 1) / gpb;
    REC_REQUIRE(grid < (1ll << 31), REC_ESHAPE, "batch too large");
#define REC_FWD(CH)                                                                              \
  hipLaunchKernelGGL((fm_fwd_kernel<VEC, LANES, CH>), dim3((unsigned)grid), dim3(kBlock), shmem, \
                     st, desc->batch, S, Dn, D, desc->row_stride, desc->num_rows,                \
                     desc->padding_idx, ids, dense, W, W1, dense_w, dense_w_one, slot_offset,    \
                     y1, y2, feat, sum_emb, status)
    if (S % 13 == 0) { REC_FWD(13); } else { REC_FWD(8); }
#undef REC_FWD
    return check_launch("rec_deepfm_fm_fwd");
  });
}

extern "C" int rec_deepfm_fm_bwd_workspace_bytes(const rec_deepfm_desc* desc, size_t* bytes) {
  if (int rc = check_desc(desc)) return rc;
  REC_REQUIRE(bytes, REC_EINVAL, "bytes is NULL");
  const int K = desc->num_dense * desc->emb_dim + desc->num_dense;
  *bytes = align_up((size_t)kNumCU * 4 * (K > 0 ? K : 1) * sizeof(float), 256);
  return REC_OK;
}

extern "C" int rec_deepfm_fm_bwd(const rec_deepfm_desc* desc, const float* dense,
                                 const float* feat, const float* sum_emb, const float* d_feat_dnn,
                                 const float* dy1, const float* dy2, float* row_grad,
                                 float* d_dense_w, float* d_dense_w_one, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  if (int rc = check_desc(desc)) return rc;
  REC_REQUIRE(feat && sum_emb && d_feat_dnn && dy1 && dy2 && row_grad, REC_EINVAL,
              "null pointer argument");
  const int S = desc->num_slots, Dn = desc->num_dense, D = desc->emb_dim;
  REC_REQUIRE(Dn == 0 || (dense && d_dense_w && d_dense_w_one), REC_EINVAL, "dense args missing");
  size_t need = 0;
  rec_deepfm_fm_bwd_workspace_bytes(desc, &need);
  REC_REQUIRE(workspace && workspace_bytes >= need, REC_EWORKSPACE, "workspace %zu < %zu",
              workspace_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  const int K = Dn * D + Dn;
  if (desc->batch == 0) {
    if (K) {
      (void)hipMemsetAsync(d_dense_w, 0, (size_t)Dn * D * sizeof(float), st);
      (void)hipMemsetAsync(d_dense_w_one, 0, (size_t)Dn * sizeof(float), st);
    }
    return REC_OK;
  }
  return dispatch_row_shape(D, D, [&](auto vec, auto lanes) -> int {
    constexpr int VEC = decltype(vec)::value, LANES = decltype(lanes)::value;
    const int grid = bwd_grid(desc->batch, LANES);
    const size_t shmem = (size_t)(kBlock / kWave) * (K > 0 ? K : 1) * sizeof(float);
    float* partial = (float*)workspace;
    hipLaunchKernelGGL((fm_bwd_kernel<VEC, LANES>), dim3(grid), dim3(kBlock), shmem, st,
                       desc->batch, S, Dn, D, dense, feat, sum_emb, d_feat_dnn, dy1, dy2,
                       row_grad, partial);
    if (K > 0) {
      hipLaunchKernelGGL(fold_partials_kernel, dim3((K + 255) / 256), dim3(256), 0, st, partial,
                         grid, K, Dn * D, d_dense_w, d_dense_w_one);
    }
    return check_launch("rec_deepfm_fm_bwd");
  });
}
