// DeepFM: embedding lookup + FM first/second order, forward and backward, fused (gfx950).
//
// Replaces the Paddle op chain of /root/reference/models/rank/deepfm/net.py:105-139
//   concat -> Embedding x2 -> multiply/unsqueeze/sum -> concat -> sum/square/sum/sub/sum/scale
// (6+ kernels each re-streaming feat_embeddings [B,39,D]) with ONE pass:
//   ids -> row gathers -> running sum / sum-of-squares in registers -> feat written once -> y1,y2
// and its backward (tools/trainer.py:151) with one streaming pass over feat / d_feat.
//
// Work decomposition (HBM-bound, no MFMA).  The unit of memory traffic is one field of one sample:
// D floats = a "row group" of LANES lanes x VEC floats (D=16: 4 lanes x float4 = one 64-B row).
// A wave holds 64/LANES field slots; they are laid out as SPW samples x FS consecutive fields, so
// that every wave-wide load/store of feat covers FS*D*4 contiguous bytes per sample
// (D=16: 2 samples x 8 fields = 2 x 512 B) — full cache lines, no half-used 128-B lines.
// A lane walks fields f = it*FS + fs; the FM sums over fields are lane-local partials folded with
// log2(FS) xor-shuffles at the end.  In the backward the same walk makes the dense-field index of a
// lane fixed across samples, so the batch reductions d_dense_w / d_dense_w_one accumulate in
// registers over a persistent grid-stride loop and are folded in a fixed order (deterministic).
#include "rec_common.h"

namespace rec {

constexpr int kWavesPerBlock = kBlock / kWave;

template <int LANES>
constexpr int fs_for() {  // fields per sample per wave instruction
  return (kWave / LANES) < 8 ? (kWave / LANES) : 8;
}

// ------------------------------------------------------------------------------------------ fwd
constexpr int kFwdUnroll = 4;

template <int VEC, int LANES>
__global__ __launch_bounds__(kBlock) void fm_fwd_kernel(
    int64_t B, int S, int Dn, int D, int stride, int64_t N, int64_t pad,
    const int64_t* __restrict__ ids, const float* __restrict__ dense, const float* __restrict__ W,
    const float* __restrict__ W1, const float* __restrict__ dense_w,
    const float* __restrict__ dense_w_one, const int64_t* __restrict__ slot_off,
    float* __restrict__ y1, float* __restrict__ y2, float* __restrict__ feat,
    float* __restrict__ sum_emb, int32_t* __restrict__ status) {
  constexpr int FS = fs_for<LANES>();
  constexpr int SPW = kWave / (LANES * FS);  // samples per wave
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_dw = smem;            // [Dn*D]
  float* s_dw1 = smem + Dn * D;  // [Dn]
  for (int i = threadIdx.x; i < Dn * D; i += kBlock) s_dw[i] = dense_w[i];
  for (int i = threadIdx.x; i < Dn; i += kBlock) s_dw1[i] = dense_w_one[i];
  __syncthreads();

  const int lane = threadIdx.x % kWave;
  const int lg = lane % LANES;
  const int fs = (lane / LANES) % FS;
  const int sp = lane / (LANES * FS);
  const int64_t wv = (int64_t)blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
  const int64_t b = wv * SPW + sp;
  const bool active = b < B;
  const int d0 = lg * VEC;
  const bool dvalid = active && d0 < D;
  const int F = S + Dn;
  const int NIT = (F + FS - 1) / FS;

  float s[VEC], q[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) s[v] = q[v] = 0.f;
  float first = 0.f;
  const int64_t* idp = ids + b * S;
  float* fb = feat + (b * F) * (int64_t)D + d0;

  for (int it0 = 0; it0 < NIT; it0 += kFwdUnroll) {
    int64_t row[kFwdUnroll];
    int f[kFwdUnroll];
#pragma unroll
    for (int u = 0; u < kFwdUnroll; ++u) {
      f[u] = (it0 + u) * FS + fs;
      row[u] = -1;
      if (active && f[u] < S) {
        const int64_t id = idp[f[u]];
        if (id != pad || pad < 0) {
          const int64_t r = slot_off ? id + slot_off[f[u]] : id;
          if (r >= 0 && r < N) {
            row[u] = r;
          } else if (lg == 0) {
            atomicOr(status, REC_FLAG_INDEX_OOB);
          }
        }
      }
    }
    float e[kFwdUnroll][VEC];
    float one[kFwdUnroll];
#pragma unroll
    for (int u = 0; u < kFwdUnroll; ++u) {
      one[u] = 0.f;
#pragma unroll
      for (int v = 0; v < VEC; ++v) e[u][v] = 0.f;
      if (row[u] >= 0) {                               // sparse field: gather (net.py:108,117)
        if (dvalid) vload<VEC>(e[u], W + row[u] * stride + d0);
        if (lg == 0) one[u] = W1[row[u]];
      } else if (active && f[u] >= S && f[u] < F) {    // dense field: x * dense_w (net.py:110-119)
        const int j = f[u] - S;
        const float x = dense[b * Dn + j];
        if (lg == 0) one[u] = x * s_dw1[j];
        if (dvalid) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) e[u][v] = x * s_dw[j * D + d0 + v];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kFwdUnroll; ++u) {
      first += one[u];
      if (dvalid && f[u] < F) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          s[v] += e[u][v];
          q[v] += e[u][v] * e[u][v];
        }
        vstore<VEC>(fb + (int64_t)f[u] * D, e[u]);
      }
    }
  }
  // fold the FS field slots of a sample (lanes that differ only in fs)
#pragma unroll
  for (int o = LANES; o < LANES * FS; o <<= 1) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      s[v] += __shfl_xor(s[v], o, kWave);
      q[v] += __shfl_xor(q[v], o, kWave);
    }
  }
  float part = 0.f;
#pragma unroll
  for (int v = 0; v < VEC; ++v) part += s[v] * s[v] - q[v];   // net.py:135-136
  if (!dvalid) part = 0.f;
  const float tot2 = group_sum<LANES>(part);
  const float tot1 = group_sum<LANES * FS>(active ? first : 0.f);
  if (dvalid && fs == 0 && sum_emb) vstore<VEC>(sum_emb + b * D + d0, s);
  if (active && fs == 0 && lg == 0) {
    y1[b] = tot1;          // net.py:113-114
    y2[b] = 0.5f * tot2;   // net.py:135
  }
}

// ------------------------------------------------------------------------------------------ bwd
constexpr int kDnMax = 16;
constexpr int kBwdUnroll = 4;
constexpr int kMaxDenseIters = 8;   // wave iterations that may contain dense fields
constexpr int kBwdMaxBlocks = kNumCU * 8;

template <int VEC, int LANES>
__global__ __launch_bounds__(kBlock) void fm_bwd_kernel(
    int64_t B, int S, int Dn, int D, const float* __restrict__ dense,
    const float* __restrict__ feat, const float* __restrict__ sum_emb,
    const float* __restrict__ dfeat, const float* __restrict__ dy1, const float* __restrict__ dy2,
    float* __restrict__ row_grad, float* __restrict__ partial) {
  constexpr int FS = fs_for<LANES>();
  constexpr int SPW = kWave / (LANES * FS);
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [waves][Dn*D + Dn]
  const int lane = threadIdx.x % kWave;
  const int wave = threadIdx.x / kWave;
  const int lg = lane % LANES;
  const int fs = (lane / LANES) % FS;
  const int sp = lane / (LANES * FS);
  const int d0 = lg * VEC;
  const int F = S + Dn;
  const int NIT = (F + FS - 1) / FS;
  const int it_d0 = S / FS;        // first wave iteration that can contain a dense field
  const int nd = NIT - it_d0;      // <= kMaxDenseIters (checked on the host)
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;

  float acc[kMaxDenseIters][VEC];
  float acc1[kMaxDenseIters];
#pragma unroll
  for (int k = 0; k < kMaxDenseIters; ++k) {
    acc1[k] = 0.f;
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[k][v] = 0.f;
  }

  for (int64_t wv = (int64_t)blockIdx.x * kWavesPerBlock + wave; wv * SPW < B; wv += nwaves) {
    const int64_t b = wv * SPW + sp;
    const bool dvalid = b < B && d0 < D;
    float sb[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) sb[v] = 0.f;
    float g1 = 0.f, g2 = 0.f;
    if (dvalid) {
      vload<VEC>(sb, sum_emb + b * D + d0);
      g1 = dy1[b];
      g2 = dy2[b];
    }
    const float* fb = feat + (b * F) * (int64_t)D + d0;
    const float* gb = dfeat + (b * F) * (int64_t)D + d0;
    float* rg = row_grad + (b * S) * (int64_t)D + d0;
    // wave iterations holding sparse fields only: d row = d_dnn + dy2 * (sum_emb - feat)
    for (int it0 = 0; it0 < it_d0; it0 += kBwdUnroll) {
      float e[kBwdUnroll][VEC], g[kBwdUnroll][VEC];
#pragma unroll
      for (int u = 0; u < kBwdUnroll; ++u) {
        if (dvalid && it0 + u < it_d0) {
          const int f = (it0 + u) * FS + fs;
          vload<VEC>(e[u], fb + (int64_t)f * D);
          vload<VEC>(g[u], gb + (int64_t)f * D);
        }
      }
#pragma unroll
      for (int u = 0; u < kBwdUnroll; ++u) {
        if (dvalid && it0 + u < it_d0) {
          const int f = (it0 + u) * FS + fs;
          float de[VEC];
#pragma unroll
          for (int v = 0; v < VEC; ++v) de[v] = g[u][v] + g2 * (sb[v] - e[u][v]);
          vstore<VEC>(rg + (int64_t)f * D, de);
        }
      }
    }
    // wave iterations that (may) hold dense fields: the (field, dims) of a lane is loop invariant
#pragma unroll
    for (int k = 0; k < kMaxDenseIters; ++k) {
      if (k < nd) {
        const int f = (it_d0 + k) * FS + fs;
        if (dvalid && f < F) {
          float e[VEC], g[VEC], de[VEC];
          vload<VEC>(e, fb + (int64_t)f * D);
          vload<VEC>(g, gb + (int64_t)f * D);
#pragma unroll
          for (int v = 0; v < VEC; ++v) de[v] = g[v] + g2 * (sb[v] - e[v]);
          if (f < S) {
            vstore<VEC>(rg + (int64_t)f * D, de);
          } else {
            const float x = dense[b * Dn + (f - S)];
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[k][v] += x * de[v];
            if (lg == 0) acc1[k] += g1 * x;
          }
        }
      }
    }
  }

  // fold the SPW samples of the wave, then the waves of the block (fixed order)
#pragma unroll
  for (int k = 0; k < kMaxDenseIters; ++k) {
#pragma unroll
    for (int o = LANES * FS; o < kWave; o <<= 1) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[k][v] += __shfl_xor(acc[k][v], o, kWave);
      acc1[k] += __shfl_xor(acc1[k], o, kWave);
    }
  }
  const int K = Dn * D + Dn;
  float* sw = smem + wave * K;
  if (sp == 0 && d0 < D) {
#pragma unroll
    for (int k = 0; k < kMaxDenseIters; ++k) {
      const int f = (it_d0 + k) * FS + fs;
      if (k < nd && f >= S && f < F) {
        const int j = f - S;
#pragma unroll
        for (int v = 0; v < VEC; ++v) sw[j * D + d0 + v] = acc[k][v];
        if (lg == 0) sw[Dn * D + j] = acc1[k];
      }
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += kBlock) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) t += smem[w * K + k];
    partial[(int64_t)k * gridDim.x + blockIdx.x] = t;   // [K][nblk]: contiguous for the fold
  }
}

// out[k] = sum_blk partial[k][blk]; one block per k, strided partial sums + fixed-order tree
__global__ __launch_bounds__(kBlock) void fold_partials_kernel(const float* __restrict__ partial,
                                                               int nblk, int split,
                                                               float* __restrict__ out0,
                                                               float* __restrict__ out1) {
  __shared__ float red[kBlock];
  const int k = blockIdx.x;
  float t = 0.f;
  for (int i = threadIdx.x; i < nblk; i += kBlock) t += partial[(int64_t)k * nblk + i];
  red[threadIdx.x] = t;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (k < split) out0[k] = red[0]; else out1[k - split] = red[0];
  }
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

template <int LANES>
static int bwd_shape_ok(int S, int Dn) {
  constexpr int FS = fs_for<LANES>();
  const int nit = (S + Dn + FS - 1) / FS;
  REC_REQUIRE(nit - S / FS <= kMaxDenseIters, REC_ESHAPE,
              "num_dense %d spans more than %d wave iterations at this emb_dim", Dn,
              kMaxDenseIters);
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
  if (desc->batch == 0) return REC_OK;
  REC_REQUIRE(ids && W && W1 && y1 && y2 && feat && status, REC_EINVAL, "null pointer argument");
  REC_REQUIRE(desc->num_dense == 0 || (dense && dense_w && dense_w_one), REC_EINVAL,
              "dense inputs missing");
  const int S = desc->num_slots, Dn = desc->num_dense, D = desc->emb_dim;
  const size_t shmem = (size_t)(Dn * D + Dn + 4) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  return dispatch_row_shape(D, desc->row_stride, [&](auto vec, auto lanes) -> int {
    constexpr int VEC = decltype(vec)::value, LANES = decltype(lanes)::value;
    constexpr int SPW = kWave / (LANES * fs_for<LANES>());
    const int64_t spb = (int64_t)SPW * kWavesPerBlock;  // samples per block
    const int64_t grid = (desc->batch + spb - 1) / spb;
    REC_REQUIRE(grid < (1ll << 31), REC_ESHAPE, "batch too large");
    hipLaunchKernelGGL((fm_fwd_kernel<VEC, LANES>), dim3((unsigned)grid), dim3(kBlock), shmem, st,
                       desc->batch, S, Dn, D, desc->row_stride, desc->num_rows, desc->padding_idx,
                       ids, dense, W, W1, dense_w, dense_w_one, slot_offset, y1, y2, feat, sum_emb,
                       status);
    return check_launch("rec_deepfm_fm_fwd");
  });
}

extern "C" int rec_deepfm_fm_bwd_workspace_bytes(const rec_deepfm_desc* desc, size_t* bytes) {
  if (int rc = check_desc(desc)) return rc;
  REC_REQUIRE(bytes, REC_EINVAL, "bytes is NULL");
  const int K = desc->num_dense * desc->emb_dim + desc->num_dense;
  *bytes = align_up((size_t)kBwdMaxBlocks * (K > 0 ? K : 1) * sizeof(float), 256);
  return REC_OK;
}

extern "C" int rec_deepfm_fm_bwd(const rec_deepfm_desc* desc, const float* dense,
                                 const float* feat, const float* sum_emb, const float* d_feat_dnn,
                                 const float* dy1, const float* dy2, float* row_grad,
                                 float* d_dense_w, float* d_dense_w_one, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  if (int rc = check_desc(desc)) return rc;
  const int S = desc->num_slots, Dn = desc->num_dense, D = desc->emb_dim;
  REC_REQUIRE(Dn == 0 || (d_dense_w && d_dense_w_one), REC_EINVAL, "dense args missing");
  REC_REQUIRE(desc->batch == 0 || (feat && sum_emb && d_feat_dnn && dy1 && dy2 && row_grad &&
                                   (Dn == 0 || dense)),
              REC_EINVAL, "null pointer argument");
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
    constexpr int SPW = kWave / (LANES * fs_for<LANES>());
    if (int rc = bwd_shape_ok<LANES>(S, Dn)) return rc;
    const int64_t spb = (int64_t)SPW * kWavesPerBlock;
    int64_t need_blocks = (desc->batch + spb - 1) / spb;
    const int grid = (int)(need_blocks < kBwdMaxBlocks ? need_blocks : kBwdMaxBlocks);
    const size_t shmem = (size_t)kWavesPerBlock * (K > 0 ? K : 1) * sizeof(float);
    float* partial = (float*)workspace;
    hipLaunchKernelGGL((fm_bwd_kernel<VEC, LANES>), dim3(grid), dim3(kBlock), shmem, st,
                       desc->batch, S, Dn, D, dense, feat, sum_emb, d_feat_dnn, dy1, dy2,
                       row_grad, partial);
    if (K > 0) {
      hipLaunchKernelGGL(fold_partials_kernel, dim3(K), dim3(kBlock), 0, st, partial, grid, Dn * D,
                         d_dense_w, d_dense_w_one);
    }
    return check_launch("rec_deepfm_fm_bwd");
  });
}
