// Row-sharded table routing (gfx950): which shard owns each lookup, and in what order it is sent.
//
// Reference counterpart: the key-sharded pull/push of core.PSGPU inside exe.train_from_dataset
// (/root/reference/tools/static_gpubox_trainer.py:152-160,256; models/rank/dnn/net.py:71-79) — the
// HeterPS code itself is not in the reference repository [EXT].  Here the table is split row-wise,
// owner(r) = r mod G, local row = r div G; one stable partition of the B*S lookups
// by owner produces everything the three all-to-all exchanges of a step need:
//   send_local_row : the ids message, grouped by owner, ascending position inside a group
//   send_pos       : position b*S+s of every entry (gathers row-grads into send order for the bwd)
//   send_sample    : b of every entry (gathers dy1 for the first-order table)
//   slot_of_pos    : 1 + index of a position in send order (0 for padding) — the rows come back in
//                    send order, so this is the `ids` the fused FM kernel reads the reply with.
// Integer work end to end: bit-exact targets.
#include <string.h>
#include <cstring>

#include "radix_sort.h"
#include "rec_common.h"

namespace rec {

constexpr int kMaxShards = 1024;

__global__ __launch_bounds__(kBlock) void route_keys_kernel(
    int64_t n, int S, int64_t N, int64_t pad, int G, const int64_t* __restrict__ ids,
    const int64_t* __restrict__ slot_off, uint32_t* __restrict__ keys, int32_t* __restrict__ vals,
    unsigned long long* __restrict__ counts, int32_t* __restrict__ status) {
  extern __shared__ int hist[];  // [G+1]
  for (int i = threadIdx.x; i <= G; i += kBlock) hist[i] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * kBlock) {
    const int64_t id = ids[i];
    uint32_t k = (uint32_t)G;  // dropped: padding or out of range
    if (id != pad || pad < 0) {
      const int64_t r = slot_off ? id + slot_off[i % S] : id;
      if (r >= 0 && r < N) k = (uint32_t)(r % G); else atomicOr(status, REC_FLAG_INDEX_OOB);
    }
    keys[i] = k;
    vals[i] = (int32_t)i;
    atomicAdd(&hist[k], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i <= G; i += kBlock)
    if (hist[i]) atomicAdd(&counts[i], (unsigned long long)hist[i]);
}

__global__ __launch_bounds__(kBlock) void route_emit_kernel(
    int64_t n, int S, int G, const int64_t* __restrict__ ids, const int64_t* __restrict__ slot_off,
    const uint32_t* __restrict__ keys, const int32_t* __restrict__ sorted_pos,
    int64_t* __restrict__ send_local_row, int64_t* __restrict__ send_pos,
    int64_t* __restrict__ send_sample, int64_t* __restrict__ slot_of_pos) {
  const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k >= n) return;
  const int32_t pos = sorted_pos[k];
  if (keys[k] < (uint32_t)G) {
    const int64_t r = slot_off ? ids[pos] + slot_off[pos % S] : ids[pos];
    send_local_row[k] = r / G;
    send_pos[k] = pos;
    send_sample[k] = pos / S;
    slot_of_pos[pos] = k + 1;
  } else {
    slot_of_pos[pos] = 0;
  }
}

static int route_bits(int G) {
  int b = 1;
  while ((1 << b) <= G) ++b;  // keys 0..G
  return b;
}

struct RoutePlan {
  rsort::Plan sort;
  size_t off_keys_in, off_keys_out, off_vals_in, off_vals_out, off_keys_tmp, off_vals_tmp, off_hist, off_totals, total;
};

static int plan_route(int64_t n, int G, RoutePlan* p) {
  p->sort = rsort::make_plan(n, route_bits(G));     // G <= 1024: one pass
  size_t o = 0;
  p->off_keys_in = o;  o += align_up((size_t)n * 4, 256);
  p->off_keys_out = o; o += align_up((size_t)n * 4, 256);
  p->off_vals_in = o;  o += align_up((size_t)n * 4, 256);
  p->off_vals_out = o; o += align_up((size_t)n * 4, 256);
  p->off_keys_tmp = o; o += p->sort.passes > 1 ? align_up((size_t)n * 4, 256) : 0;
  p->off_vals_tmp = o; o += p->sort.passes > 1 ? align_up((size_t)n * 4, 256) : 0;
  p->off_hist = o;     o += p->sort.hist_bytes;
  p->off_totals = o;   o += p->sort.totals_bytes;
  p->total = o;
  return REC_OK;
}

}  // namespace rec

using namespace rec;

extern "C" int rec_shard_route_workspace_bytes(int64_t n, int32_t num_shards, size_t* bytes) {
  REC_REQUIRE(bytes && n >= 0 && num_shards >= 1 && num_shards <= kMaxShards, REC_EINVAL,
              "bad arguments");
  REC_REQUIRE(n < (1ll << 31) - 1, REC_ESHAPE, "n too large for int32 positions");
  if (n == 0) { *bytes = 256; return REC_OK; }
  RoutePlan p;
  if (int rc = plan_route(n, num_shards, &p)) return rc;
  *bytes = p.total;
  return REC_OK;
}

extern "C" int rec_shard_route(int64_t n, int32_t num_slots, int64_t num_rows, int64_t padding_idx,
                               int32_t num_shards, const int64_t* ids, const int64_t* slot_offset,
                               int64_t* send_local_row, int64_t* send_pos, int64_t* send_sample,
                               int64_t* slot_of_pos, int64_t* send_counts, int32_t* status,
                               void* workspace, size_t workspace_bytes, void* stream) {
  REC_REQUIRE(n >= 0 && num_slots > 0 && num_rows > 0 && num_shards >= 1 &&
                  num_shards <= kMaxShards, REC_EINVAL, "bad sizes");
  REC_REQUIRE(n < (1ll << 31) - 1, REC_ESHAPE, "n too large for int32 positions");
  REC_REQUIRE(send_counts && status, REC_EINVAL, "null pointer argument");
  hipStream_t st = (hipStream_t)stream;
  const int G = num_shards;
  if (hipMemsetAsync(send_counts, 0, (size_t)(G + 1) * sizeof(int64_t), st) != hipSuccess) {
    set_error("memset of send_counts failed");
    return REC_EHIP;
  }
  if (n == 0) return REC_OK;
  REC_REQUIRE(ids && send_local_row && send_pos && send_sample && slot_of_pos, REC_EINVAL,
              "null pointer argument");
  RoutePlan p;
  if (int rc = plan_route(n, G, &p)) return rc;
  REC_REQUIRE(workspace && workspace_bytes >= p.total, REC_EWORKSPACE, "workspace %zu < %zu",
              workspace_bytes, p.total);
  char* base = (char*)workspace;
  uint32_t* keys_in = (uint32_t*)(base + p.off_keys_in);
  uint32_t* keys_out = (uint32_t*)(base + p.off_keys_out);
  int32_t* vals_in = (int32_t*)(base + p.off_vals_in);
  int32_t* vals_out = (int32_t*)(base + p.off_vals_out);
  int64_t grid = (n + kBlock - 1) / kBlock;
  if (grid > kNumCU * 8) grid = kNumCU * 8;
  hipLaunchKernelGGL(route_keys_kernel, dim3((unsigned)grid), dim3(kBlock),
                     (size_t)(G + 1) * sizeof(int), st, n, num_slots, num_rows, padding_idx, G, ids,
                     slot_offset, keys_in, vals_in, (unsigned long long*)send_counts, status);
  // stable partition by owner: one radix pass over the owner key (hand-written, csrc/radix_sort.h)
  rsort::BufSrc<uint32_t> src{keys_in, vals_in};
  if (int rc = rsort::sort_pairs<uint32_t>(n, p.sort, src, (uint32_t*)(base + p.off_keys_tmp),
                                           (int32_t*)(base + p.off_vals_tmp), keys_out, vals_out,
                                           base + p.off_hist, base + p.off_totals, st))
    return rc;
  const unsigned g2 = (unsigned)((n + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(route_emit_kernel, dim3(g2), dim3(kBlock), 0, st, n, num_slots, G, ids,
                     slot_offset, keys_out, vals_out, send_local_row, send_pos, send_sample,
                     slot_of_pos);
  return check_launch("rec_shard_route");
}
