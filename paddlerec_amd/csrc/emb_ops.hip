// Plain embedding gather and LoD multi-slot gather + sum-pool (gfx950).
//
//   rec_emb_gather         <- paddle.nn.Embedding forward
//                             (/root/reference/models/rank/deepfm/net.py:108,117;
//                              dcn_v2/net.py:93-96; din/net.py:141-147)
//   rec_emb_gather_sumpool <- paddle.static.nn.sparse_embedding + sequence_pool('sum')
//                             (/root/reference/models/rank/slot_dnn/net.py:63-75;
//                              dnn/static_model_lod.py:70-97)
// Row groups of LANES lanes, VEC floats per lane (see deepfm_fm.hip).  The sum-pool walks a
// sample's LoD segment CH ids at a time so CH row gathers are in flight per group; the pooled
// count of non-padding ids is an integer output (bit-exact target).
#include "rec_common.h"
#include "tail_roles.h"

namespace rec {

template <int VEC, int LANES>
__global__ __launch_bounds__(kBlock) void emb_gather_kernel(
    int64_t n, int D, int stride, int64_t N, int64_t pad, const int64_t* __restrict__ ids,
    const float* __restrict__ W, float* __restrict__ out, int group, int64_t group_stride,
    int32_t* __restrict__ status) {
  const int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / LANES;
  const int lg = threadIdx.x % LANES;
  const int d0 = lg * VEC;
  if (i >= n || d0 >= D) return;
  const int64_t id = ids[i];
  float e[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) e[v] = 0.f;
  if (id != pad || pad < 0) {
    if (id >= 0 && id < N) vload<VEC>(e, W + id * stride + d0);
    else if (lg == 0) atomicOr(status, REC_FLAG_INDEX_OOB);
  }
  const int64_t o = group > 0 ? (i / group) * group_stride + (i % group) * D : i * D;
  vstore<VEC>(out + o + d0, e);
}

// Several gathers in ONE launch (tail_roles.h, GatherJobs): a launch-bound step pays per launch, not per byte — DIN at batch 32
// looks up 32 target items, 32 target categories and 32 item biases as three launches of one block each.
__global__ __launch_bounds__(kBlock) void emb_gather_multi_kernel(GatherJobs js, int32_t* __restrict__ status) {
  gather_role((int)blockIdx.x, threadIdx.x, js, status);
}

// Owner-side lookup of a row-sharded DeepFM table: BOTH embeddings of a row from its ONE record line
//   rec [N, stride] = W(D) | W1 | ...   ->  out_w [n, D], out_w1 [n]
// (two rec_emb_gather launches read every line twice).  PS tables are born lazily: with init_range > 0 a row whose
// state float rec[row*stride + state_off] is 0 reads as its creation values — embed_w (= W1, element 0) always,
// embedx (= W, element 1+d) only when init_embedx — keyed by the GLOBAL row = row * row_mul + row_add.
template <int VEC, int LANES>
__global__ __launch_bounds__(kBlock) void record_gather_kernel(
    int64_t n, int D, int stride, int64_t N, const int64_t* __restrict__ rows, const float* __restrict__ rec,
    float* __restrict__ out_w, float* __restrict__ out_w1, int state_off, float init_range, int init_embedx,
    uint64_t seed, int64_t row_mul, int64_t row_add, int32_t* __restrict__ status) {
  const int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / LANES;
  const int lg = threadIdx.x % LANES;
  const int d0 = lg * VEC;
  if (i >= n) return;
  const int64_t row = rows[i];
  const bool ok = row >= 0 && row < N;
  if (!ok && lg == 0) atomicOr(status, REC_FLAG_INDEX_OOB);
  const float* r = rec + (ok ? row : 0) * stride;
  const bool unborn = ok && state_off >= 0 && r[state_off] == 0.f;
  const int64_t grow = row * row_mul + row_add;
  if (d0 < D) {
    float e[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) e[v] = 0.f;
    if (ok) {
      if (unborn) {
#pragma unroll
        for (int v = 0; v < VEC; ++v)
          e[v] = (init_embedx && d0 + v < D) ? ps_init_value(seed, grow, 1 + d0 + v, init_range) : 0.f;
      } else {
        vload<VEC>(e, r + d0);
      }
    }
    vstore<VEC>(out_w + i * D + d0, e);
  }
  if (lg == 0) out_w1[i] = !ok ? 0.f : (unborn ? ps_init_value(seed, grow, 0, init_range) : r[D]);
}

constexpr int kPoolCH = 8;

template <int VEC, int LANES>
__global__ __launch_bounds__(kBlock) void emb_sumpool_kernel(
    int64_t B, int D, int stride, int64_t N, int64_t pad, const int64_t* __restrict__ ids,
    const int64_t* __restrict__ lod, const float* __restrict__ W, float* __restrict__ out,
    int32_t* __restrict__ counts, int32_t* __restrict__ status) {
  const int64_t b = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / LANES;
  const int lg = threadIdx.x % LANES;
  const int d0 = lg * VEC;
  if (b >= B) return;
  const bool dvalid = d0 < D;
  const int64_t beg = lod[b], end = lod[b + 1];
  float acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
  int cnt = 0;
  for (int64_t k0 = beg; k0 < end; k0 += kPoolCH) {
    int64_t row[kPoolCH];
#pragma unroll
    for (int c = 0; c < kPoolCH; ++c) {
      row[c] = -1;
      if (k0 + c < end) {
        const int64_t id = ids[k0 + c];
        if (id != pad || pad < 0) {
          if (id >= 0 && id < N) { row[c] = id; ++cnt; }
          else if (lg == 0) atomicOr(status, REC_FLAG_INDEX_OOB);
        }
      }
    }
    float e[kPoolCH][VEC];
#pragma unroll
    for (int c = 0; c < kPoolCH; ++c) {
      if (dvalid && row[c] >= 0) vload<VEC>(e[c], W + row[c] * stride + d0);
      else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) e[c][v] = 0.f;
      }
    }
#pragma unroll
    for (int c = 0; c < kPoolCH; ++c) {  // ascending-k order, same as the oracle
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[v] += e[c][v];
    }
  }
  if (dvalid) vstore<VEC>(out + b * D + d0, acc);
  if (lg == 0 && counts) counts[b] = cnt;
}

template <int VEC, int LANES>
__global__ __launch_bounds__(kBlock) void emb_sumpool_bwd_kernel(
    int64_t B, int D, const int64_t* __restrict__ lod, const float* __restrict__ d_out,
    float* __restrict__ row_grad) {
  const int64_t b = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / LANES;
  const int d0 = (threadIdx.x % LANES) * VEC;
  if (b >= B || d0 >= D) return;
  float g[VEC];
  vload<VEC>(g, d_out + b * D + d0);
  for (int64_t k = lod[b]; k < lod[b + 1]; ++k) vstore<VEC>(row_grad + k * D + d0, g);
}

}  // namespace rec

using namespace rec;

extern "C" int rec_emb_gather(int64_t n, int32_t emb_dim, int32_t row_stride, int64_t num_rows,
                              int64_t padding_idx, const int64_t* ids, const float* W, float* out,
                              int32_t out_group, int64_t out_group_stride, int32_t* status,
                              void* stream) {
  REC_REQUIRE(n >= 0 && emb_dim > 0 && row_stride >= emb_dim && num_rows > 0, REC_EINVAL,
              "bad sizes");
  if (n == 0) return REC_OK;
  REC_REQUIRE(ids && W && out && status, REC_EINVAL, "null pointer argument");
  REC_REQUIRE(out_group <= 0 || out_group_stride >= (int64_t)out_group * emb_dim, REC_EINVAL,
              "out_group_stride too small");
  // float4 stores need 16-B aligned output rows: fall back to scalar lanes otherwise
  const bool out_vec = out_group <= 0 || (out_group_stride % 4 == 0 && ((uintptr_t)out) % 16 == 0);
  return dispatch_row_shape(emb_dim, out_vec ? row_stride : row_stride | 1, [&](auto vec, auto lanes) -> int {
    constexpr int VEC = decltype(vec)::value, LANES = decltype(lanes)::value;
    const int64_t grid = (n * LANES + kBlock - 1) / kBlock;
    REC_REQUIRE(grid < (1ll << 31), REC_ESHAPE, "n too large");
    hipLaunchKernelGGL((emb_gather_kernel<VEC, LANES>), dim3((unsigned)grid), dim3(kBlock), 0,
                       (hipStream_t)stream, n, emb_dim, row_stride, num_rows, padding_idx, ids, W,
                       out, out_group, out_group_stride, status);
    return check_launch("rec_emb_gather");
  });
}

extern "C" int rec_emb_gather_sumpool(int64_t batch, int32_t emb_dim, int32_t row_stride,
                                      int64_t num_rows, int64_t padding_idx, const int64_t* ids,
                                      const int64_t* lod, const float* W, float* out,
                                      int32_t* counts, int32_t* status, void* stream) {
  REC_REQUIRE(batch >= 0 && emb_dim > 0 && row_stride >= emb_dim && num_rows > 0, REC_EINVAL,
              "bad sizes");
  if (batch == 0) return REC_OK;
  REC_REQUIRE(lod && W && out && status, REC_EINVAL, "null pointer argument");
  return dispatch_row_shape(emb_dim, row_stride, [&](auto vec, auto lanes) -> int {
    constexpr int VEC = decltype(vec)::value, LANES = decltype(lanes)::value;
    const int64_t grid = (batch * LANES + kBlock - 1) / kBlock;
    REC_REQUIRE(grid < (1ll << 31), REC_ESHAPE, "batch too large");
    hipLaunchKernelGGL((emb_sumpool_kernel<VEC, LANES>), dim3((unsigned)grid), dim3(kBlock), 0,
                       (hipStream_t)stream, batch, emb_dim, row_stride, num_rows, padding_idx, ids,
                       lod, W, out, counts, status);
    return check_launch("rec_emb_gather_sumpool");
  });
}

extern "C" int rec_emb_sumpool_bwd(int64_t batch, int32_t emb_dim, const int64_t* lod,
                                   const float* d_out, float* row_grad, void* stream) {
  REC_REQUIRE(batch >= 0 && emb_dim > 0, REC_EINVAL, "bad sizes");
  if (batch == 0) return REC_OK;
  REC_REQUIRE(lod && d_out && row_grad, REC_EINVAL, "null pointer argument");
  return dispatch_row_shape(emb_dim, emb_dim, [&](auto vec, auto lanes) -> int {
    constexpr int VEC = decltype(vec)::value, LANES = decltype(lanes)::value;
    const int64_t grid = (batch * LANES + kBlock - 1) / kBlock;
    REC_REQUIRE(grid < (1ll << 31), REC_ESHAPE, "batch too large");
    hipLaunchKernelGGL((emb_sumpool_bwd_kernel<VEC, LANES>), dim3((unsigned)grid), dim3(kBlock), 0,
                       (hipStream_t)stream, batch, emb_dim, lod, d_out, row_grad);
    return check_launch("rec_emb_sumpool_bwd");
  });
}

extern "C" int rec_record_gather(int64_t n, int32_t emb_dim, int32_t rec_stride, int64_t num_rows,
                                 const int64_t* rows, const float* rec, float* out_w, float* out_w1,
                                 const rec_lazy_init* lazy, int32_t* status, void* stream) {
  REC_REQUIRE(n >= 0 && emb_dim > 0 && rec_stride > emb_dim && num_rows > 0, REC_EINVAL,
              "bad sizes (the record holds W(D) | W1 | ...)");
  if (n == 0) return REC_OK;
  REC_REQUIRE(rows && rec && out_w && out_w1 && status, REC_EINVAL, "null pointer argument");
  rec_lazy_init lz = {-1, 0, 0.f, 0, 1, 0};
  if (lazy && lazy->init_range > 0.f) lz = *lazy;
  REC_REQUIRE(lz.state_offset < rec_stride && (lz.state_offset < 0 || lz.state_offset > emb_dim), REC_EINVAL,
              "state float must sit behind W | W1 inside the record");
  const bool vec = ((uintptr_t)rec) % 16 == 0 && ((uintptr_t)out_w) % 16 == 0;
  return dispatch_row_shape(emb_dim, vec ? rec_stride : rec_stride | 1, [&](auto vec_, auto lanes) -> int {
    constexpr int VEC = decltype(vec_)::value, LANES = decltype(lanes)::value;
    const int64_t grid = (n * LANES + kBlock - 1) / kBlock;
    REC_REQUIRE(grid < (1ll << 31), REC_ESHAPE, "n too large");
    hipLaunchKernelGGL((record_gather_kernel<VEC, LANES>), dim3((unsigned)grid), dim3(kBlock), 0,
                       (hipStream_t)stream, n, emb_dim, rec_stride, num_rows, rows, rec, out_w, out_w1,
                       lz.state_offset, lz.init_range, lz.init_dims > 1 ? 1 : 0, lz.seed, lz.row_mul, lz.row_add,
                       status);
    return check_launch("rec_record_gather");
  });
}

int rec::gather_jobs_make(int32_t count, const GatherJob* jobs, GatherJobs* out) {
  REC_REQUIRE(count >= 0 && count <= kGatherJobsMax && (count == 0 || jobs) && out, REC_EINVAL, "bad arguments");
  out->count = 0;
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    const GatherJob& a = jobs[i];
    REC_REQUIRE(a.n >= 0 && a.emb_dim > 0 && a.row_stride >= a.emb_dim && a.num_rows > 0, REC_EINVAL, "job %d: bad sizes", i);
    if (a.n == 0) continue;
    REC_REQUIRE(a.ids && a.W && a.out, REC_EINVAL, "job %d: null pointer argument", i);
    out->block0[out->count] = blocks;
    out->j[out->count++] = a;
    blocks += (int)((a.n * a.emb_dim + kBlock - 1) / kBlock);
  }
  out->blocks = blocks;
  return REC_OK;
}

int rec::emb_gather_multi(int32_t count, const GatherJob* jobs, int32_t* status, void* stream) {
  REC_REQUIRE(status, REC_EINVAL, "status is NULL");
  GatherJobs js;
  if (int rc = gather_jobs_make(count, jobs, &js)) return rc;
  if (js.count == 0) return REC_OK;
  hipLaunchKernelGGL(emb_gather_multi_kernel, dim3((unsigned)js.blocks), dim3(kBlock), 0, (hipStream_t)stream, js, status);
  return check_launch("emb_gather_multi");
}
