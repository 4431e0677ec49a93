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

// =====================================================================================================================
// Deduplicated lookup plan (round 6; paddlerec_amd/sharded.py REC_SHARD_DEDUP).  HeterPS dedups the keys of a pass before
// it builds the per-GPU tables (tools/static_gpubox_trainer.py:237-246 load_into_memory -> PSGPU.begin_pass [EXT]); here a
// rank asks every owner for the DISTINCT rows of its batch only.  On top of rec_ids_group over the shard-major key
//     key(row) = (row % G) * local_rows + row / G          (owner-major, ascending local row inside an owner)
// four small kernels turn the grouping into fixed-capacity send slots (owner o owns slots [o * cap, (o + 1) * cap)):
//   send_rows    [G * cap] i64 : local row per slot, `local_rows` (the sentinel) in the empty ones
//   slot_of_uniq [n] i64       : slot of distinct row u (G * cap: behind the capacity), first n_uniq[0] entries
//   slot_of_pos  [n] i64       : 1 + slot of the position's row, 0 for padding / dropped / overflowed positions — the
//                                `ids` rec_deepfm_fm_fwd reads the reply table with
//   counts       [G] i64       : distinct rows per owner (before the capacity cut)
// Nothing is read back: sizes are the capacity; an owner that needs more sets REC_FLAG_EXCHANGE_OVERFLOW.
namespace rec {

__global__ __launch_bounds__(kBlock) void dedup_keys_kernel(int64_t n, int S, int64_t N, int64_t pad, int G, int64_t L,
                                                            const int64_t* __restrict__ ids,
                                                            const int64_t* __restrict__ slot_off, int64_t* __restrict__ keys,
                                                            int64_t* __restrict__ slot_of_pos, int32_t* __restrict__ status) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int64_t id = ids[i];
  int64_t k = (int64_t)G * L;                           // "no lookup": dropped by the grouping as its padding key
  if (id != pad || pad < 0) {
    const int64_t r = slot_off ? id + slot_off[i % S] : id;
    if (r >= 0 && r < N) k = (r % G) * L + r / G; else atomicOr(status, REC_FLAG_INDEX_OOB);
  }
  keys[i] = k;
  slot_of_pos[i] = 0;
}

// start[o] = index of the first distinct row of owner o (uniq is owner-major ascending); owners without rows are filled
// by dedup_fill_kernel.  start has G + 1 entries, preset to -1.
__global__ __launch_bounds__(kBlock) void dedup_starts_kernel(int64_t n, int64_t L, const int32_t* __restrict__ n_uniq,
                                                              const int64_t* __restrict__ uniq, int64_t* __restrict__ start) {
  const int64_t u = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t U = n_uniq[0];
  if (u >= U || u >= n) return;
  const int64_t o = uniq[u] / L;
  if (u == 0 || uniq[u - 1] / L != o) start[o] = u;
}
__global__ void dedup_fill_kernel(int G, const int32_t* __restrict__ n_uniq, int64_t* __restrict__ start,
                                  int64_t* __restrict__ counts, int cap, int32_t* __restrict__ status) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  start[G] = n_uniq[0];
  for (int o = G - 1; o >= 0; --o)
    if (start[o] < 0) start[o] = start[o + 1];
  bool over = false;
  for (int o = 0; o < G; ++o) {
    counts[o] = start[o + 1] - start[o];
    over |= counts[o] > cap;
  }
  if (over) atomicOr(status, REC_FLAG_EXCHANGE_OVERFLOW);
}
// thread t: slot t of the send buffer (gather form: no races, no prefill) and distinct row t's slot
__global__ __launch_bounds__(kBlock) void dedup_slots_kernel(int64_t n, int G, int64_t L, int cap,
                                                             const int32_t* __restrict__ n_uniq,
                                                             const int64_t* __restrict__ uniq,
                                                             const int64_t* __restrict__ start,
                                                             int64_t* __restrict__ send_rows,
                                                             int64_t* __restrict__ slot_of_uniq) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t slots = (int64_t)G * cap;
  if (t < slots) {
    const int o = (int)(t / cap);
    const int64_t j = t - (int64_t)o * cap;
    send_rows[t] = j < start[o + 1] - start[o] ? uniq[start[o] + j] - (int64_t)o * L : L;
  }
  if (t < n) {
    int64_t s = slots;
    if (t < n_uniq[0]) {
      const int64_t o = uniq[t] / L, j = t - start[o];
      if (j < cap) s = o * cap + j;
    }
    slot_of_uniq[t] = s;
  }
}
// thread k: the k-th lookup in sorted order belongs to distinct row u = #{u' : seg_offset[u' + 1] <= k}
__global__ __launch_bounds__(kBlock) void dedup_positions_kernel(int64_t n, int64_t slots, const int32_t* __restrict__ n_uniq,
                                                                 const int32_t* __restrict__ seg_offset,
                                                                 const int32_t* __restrict__ sorted_pos,
                                                                 const int64_t* __restrict__ slot_of_uniq,
                                                                 int64_t* __restrict__ slot_of_pos) {
  const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k >= n || k >= n_uniq[1]) return;
  int64_t lo = 0, hi = n_uniq[0];                      // first u with seg_offset[u + 1] > k
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (seg_offset[mid + 1] > k) hi = mid; else lo = mid + 1;
  }
  const int64_t s = slot_of_uniq[lo];
  slot_of_pos[sorted_pos[k]] = s < slots ? s + 1 : 0;
}

}  // namespace rec

extern "C" int rec_dedup_plan_workspace_bytes(int64_t n, int32_t num_shards, int64_t local_rows, size_t* bytes) {
  REC_REQUIRE(bytes && n >= 0 && num_shards >= 1 && num_shards <= kMaxShards && local_rows >= 1, REC_EINVAL, "bad arguments");
  size_t g = 0;
  if (int rc = rec_ids_group_workspace_bytes(n, (int64_t)num_shards * local_rows + 1, &g)) return rc;
  *bytes = align_up((size_t)(n > 0 ? n : 1) * 8, 256) + align_up((size_t)(num_shards + 1) * 8, 256) + align_up(g, 256);
  return REC_OK;
}

extern "C" int rec_dedup_plan(int64_t n, int32_t num_slots, int64_t num_rows, int64_t padding_idx, int32_t num_shards,
                              int64_t local_rows, int32_t cap, const int64_t* ids, const int64_t* slot_offset,
                              int32_t* sorted_pos, int64_t* uniq_rows, int32_t* seg_offset, int32_t* n_uniq,
                              int64_t* send_rows, int64_t* slot_of_pos, int64_t* slot_of_uniq, int64_t* counts,
                              int32_t* status, void* workspace, size_t workspace_bytes, void* stream) {
  REC_REQUIRE(n >= 0 && num_slots > 0 && num_rows > 0 && num_shards >= 1 && num_shards <= kMaxShards && local_rows >= 1 &&
                  cap >= 1, REC_EINVAL, "bad sizes");
  REC_REQUIRE((int64_t)num_shards * local_rows >= num_rows, REC_EINVAL, "num_shards x local_rows < num_rows");
  REC_REQUIRE(n < (1ll << 31) - 1 && (int64_t)num_shards * cap < (1ll << 31), REC_ESHAPE, "n / slots too large");
  REC_REQUIRE(sorted_pos && uniq_rows && seg_offset && n_uniq && send_rows && slot_of_pos && slot_of_uniq && counts && status,
              REC_EINVAL, "null pointer argument");
  size_t need = 0;
  if (int rc = rec_dedup_plan_workspace_bytes(n, num_shards, local_rows, &need)) return rc;
  REC_REQUIRE(workspace && workspace_bytes >= need, REC_EWORKSPACE, "workspace %zu < %zu", workspace_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  const int G = num_shards;
  const int64_t slots = (int64_t)G * cap, kpad = (int64_t)G * local_rows;
  char* base = (char*)workspace;
  int64_t* keys = (int64_t*)base;
  int64_t* start = (int64_t*)(base + align_up((size_t)(n > 0 ? n : 1) * 8, 256));
  void* gws = base + align_up((size_t)(n > 0 ? n : 1) * 8, 256) + align_up((size_t)(G + 1) * 8, 256);
  const size_t gws_bytes = workspace_bytes - (size_t)((char*)gws - base);
  REC_REQUIRE(hipMemsetAsync(start, 0xff, (size_t)(G + 1) * 8, st) == hipSuccess, REC_EHIP, "memset failed");
  const unsigned gn = (unsigned)((n + kBlock - 1) / kBlock);
  if (n > 0) {
    REC_REQUIRE(ids, REC_EINVAL, "ids is NULL");
    hipLaunchKernelGGL(dedup_keys_kernel, dim3(gn), dim3(kBlock), 0, st, n, num_slots, num_rows, padding_idx, G, local_rows, ids,
                       slot_offset, keys, slot_of_pos, status);
  }
  if (int rc = rec_ids_group(n, 1, kpad + 1, kpad, keys, nullptr, sorted_pos, uniq_rows, seg_offset, n_uniq, status, gws,
                             gws_bytes, stream))
    return rc;
  if (n > 0) hipLaunchKernelGGL(dedup_starts_kernel, dim3(gn), dim3(kBlock), 0, st, n, local_rows, n_uniq, uniq_rows, start);
  hipLaunchKernelGGL(dedup_fill_kernel, dim3(1), dim3(64), 0, st, G, n_uniq, start, counts, cap, status);
  const int64_t tmax = slots > n ? slots : n;
  hipLaunchKernelGGL(dedup_slots_kernel, dim3((unsigned)((tmax + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, n, G, local_rows,
                     cap, n_uniq, uniq_rows, start, send_rows, slot_of_uniq);
  if (n > 0)
    hipLaunchKernelGGL(dedup_positions_kernel, dim3(gn), dim3(kBlock), 0, st, n, slots, n_uniq, seg_offset, sorted_pos,
                       slot_of_uniq, slot_of_pos);
  return check_launch("rec_dedup_plan");
}
