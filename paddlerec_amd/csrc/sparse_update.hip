// SelectedRows merge (integer grouping) + lazy sparse Adam + dense Adam (gfx950).
//
// Reference behaviour replaced (all inside Paddle core [EXT], driven from
// /root/reference/models/rank/deepfm/net.py:62-70,80 `sparse=True` and
// deepfm/dygraph_model.py:61-65 / static_model.py:83-84 paddle.optimizer.Adam):
//   lookup_table_v2_grad -> SelectedRows{rows=ids, value} -> MergeAdd -> adam (lazy rows).
// Here: one stable key sort of the B*S lookups by row (integer, bit-exact, hidden behind the
// top-MLP GEMMs on a side stream), then ONE kernel that sums each row's duplicate gradients in
// ascending-position order (deterministic, unlike atomics) and applies Adam to P/M/V in place —
// the merged [U,D] gradient never goes to HBM.
#include <string.h>
#include <cstring>

#include <cmath>

#include "radix_sort.h"
#include "slot_group.h"
#include <stdlib.h>
#include <type_traits>

#include "rec_common.h"
#include "tail_roles.h"

namespace rec {

constexpr int kSegTile = REC_SEG_TILE, kSegLong = REC_SEG_LONG;

// ------------------------------------------------------------------------------- ids grouping
// keys of the B*S lookups, computed on the fly by the first radix pass (no key materialisation pass):
// row = id + slot_offset[pos % S]; padding and out-of-range lookups get the sentinel N (sorts behind every row)
template <class KeyT>
struct IdsSrc {
  const int64_t* ids;
  const int64_t* slot_off;
  int S;
  int64_t N, pad;
  int32_t* status;
  const int32_t* payload;   // what travels with the key: payload[i], or the position i itself when null
  bool drop_invalid;        // the sort leaves the sentinel keys out after its first pass (rsort::sort_pairs n_live)
  int64_t slot_rows = 0;    // > 0 (rec_ids_group_slots): slot s owns rows [s * slot_rows, (s + 1) * slot_rows)
  __device__ __forceinline__ bool drop(KeyT k) const { return drop_invalid && k == (KeyT)N; }
  __device__ __forceinline__ KeyT key(int64_t i) const {
    const int64_t id = ids[i];
    KeyT k = (KeyT)N;
    if (id != pad || pad < 0) {
      int64_t r = slot_off ? id + slot_off[(int)i % S] : id;
      if (slot_rows > 0) r = (id >= 0 && id < slot_rows) ? id + (int64_t)((int)i % S) * slot_rows : -1;
      if (r >= 0 && r < N) k = (KeyT)r; else atomicOr(status, REC_FLAG_INDEX_OOB);
    }
    return k;
  }
  __device__ __forceinline__ int32_t val(int64_t i) const { return payload ? payload[i] : (int32_t)i; }
};

// Heads of the sorted key runs.  A block owns a tile of kHeadsTile consecutive sorted positions, a wave walks it
// in 64-position chunks (coalesced): head flags by comparing with the left neighbour (shuffle; lane 0 reads one key
// more), counted / ranked with ballots — no per-thread serial scans.
constexpr int kHeadsChunks = 8;                                            // chunks per wave
constexpr int kHeadsWaves = rsort::kThreads / kWave;
constexpr int kHeadsTile = kHeadsWaves * kHeadsChunks * kWave;             // 2048 positions per block

template <class KeyT>
__device__ __forceinline__ unsigned long long head_mask(int64_t n, KeyT sentinel, const KeyT* __restrict__ keys,
                                                        int64_t i, int lane, bool* valid, KeyT* key_out) {
  const bool in = i < n;
  const KeyT k = in ? keys[i] : sentinel;
  KeyT left = __shfl_up(k, 1, kWave);
  if (lane == 0) left = (i > 0 && i <= n) ? keys[i - 1] : sentinel;
  const bool v = in && k != sentinel;
  *valid = v;
  *key_out = k;
  return __ballot(v && (i == 0 || k != left));
}

// per tile: {number of heads, number of valid positions}; also raises n_uniq[2] when some row owns >= REC_SEG_LONG
// positions (keys[i] == keys[i + REC_SEG_LONG - 1]) — the hot-row partial sums are skipped when none does
template <class KeyT>
__global__ __launch_bounds__(rsort::kThreads) void heads_count_kernel(int64_t n, KeyT sentinel,
                                                                      const KeyT* __restrict__ keys,
                                                                      int32_t* __restrict__ cnt,
                                                                      int32_t* __restrict__ n_uniq) {
  __shared__ int red[2][kHeadsWaves];
  const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
  const int64_t base = (int64_t)blockIdx.x * kHeadsTile;
  int heads = 0, valid = 0, has_long = 0;
#pragma unroll
  for (int c = 0; c < kHeadsChunks; ++c) {
    const int64_t i = base + (int64_t)(c * kHeadsWaves + wave) * kWave + lane;
    bool v;
    KeyT k;
    const unsigned long long hm = head_mask<KeyT>(n, sentinel, keys, i, lane, &v, &k);
    heads += __popcll(hm);
    valid += __popcll(__ballot(v));
    if (v && i + kSegLong - 1 < n && keys[i + kSegLong - 1] == k) has_long = 1;
  }
  if (__ballot(has_long) != 0 && lane == 0) atomicOr(&n_uniq[2], 1);
  if (lane == 0) { red[0][wave] = heads; red[1][wave] = valid; }
  __syncthreads();
  if (threadIdx.x == 0) {
    int h = 0, v = 0;
    for (int w = 0; w < kHeadsWaves; ++w) { h += red[0][w]; v += red[1][w]; }
    cnt[blockIdx.x * 2] = h;
    cnt[blockIdx.x * 2 + 1] = v;
  }
}

// one block: exclusive scan of the per-tile head counts (in place, cnt[2*blk]); n_uniq[0..1] = {U, n_valid};
// seg_off[U] = n_valid (the end of the last segment)
__global__ __launch_bounds__(rsort::kThreads) void heads_scan_kernel(int nblk, int32_t* __restrict__ cnt,
                                                                     int32_t* __restrict__ seg_off,
                                                                     int32_t* __restrict__ n_uniq) {
  __shared__ int part[rsort::kThreads], vpart[rsort::kThreads];
  const int per = (nblk + rsort::kThreads - 1) / rsort::kThreads;
  const int lo = threadIdx.x * per, hi = min(lo + per, nblk);
  int s = 0, v = 0;
  for (int i = lo; i < hi; ++i) { s += cnt[2 * i]; v += cnt[2 * i + 1]; }
  part[threadIdx.x] = s;
  vpart[threadIdx.x] = v;
  __syncthreads();
  for (int o = 1; o < rsort::kThreads; o <<= 1) {
    const int a = (int)threadIdx.x >= o ? part[threadIdx.x - o] : 0;
    const int b = (int)threadIdx.x >= o ? vpart[threadIdx.x - o] : 0;
    __syncthreads();
    part[threadIdx.x] += a;
    vpart[threadIdx.x] += b;
    __syncthreads();
  }
  int run = threadIdx.x ? part[threadIdx.x - 1] : 0;
  for (int i = lo; i < hi; ++i) {
    const int t = cnt[2 * i];
    cnt[2 * i] = run;
    run += t;
  }
  if (threadIdx.x == rsort::kThreads - 1) {
    const int U = part[rsort::kThreads - 1], nv = vpart[rsort::kThreads - 1];
    n_uniq[0] = U;
    n_uniq[1] = nv;
    seg_off[U] = nv;
  }
}

template <class KeyT>
__global__ __launch_bounds__(rsort::kThreads) void heads_emit_kernel(int64_t n, KeyT sentinel,
                                                                     const KeyT* __restrict__ keys,
                                                                     const int32_t* __restrict__ cnt,
                                                                     int64_t* __restrict__ uniq,
                                                                     int32_t* __restrict__ seg_off) {
  __shared__ int ccnt[kHeadsChunks * kHeadsWaves];
  const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
  const int64_t base = (int64_t)blockIdx.x * kHeadsTile;
  unsigned long long hm[kHeadsChunks];
  KeyT kk[kHeadsChunks];
#pragma unroll
  for (int c = 0; c < kHeadsChunks; ++c) {
    const int64_t i = base + (int64_t)(c * kHeadsWaves + wave) * kWave + lane;
    bool v;
    hm[c] = head_mask<KeyT>(n, sentinel, keys, i, lane, &v, &kk[c]);
    if (lane == 0) ccnt[c * kHeadsWaves + wave] = __popcll(hm[c]);
  }
  __syncthreads();
  if (threadIdx.x == 0) {   // exclusive scan over the tile's 32 chunks (in position order)
    int run = cnt[blockIdx.x * 2];
    for (int q = 0; q < kHeadsChunks * kHeadsWaves; ++q) {
      const int t = ccnt[q];
      ccnt[q] = run;
      run += t;
    }
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < kHeadsChunks; ++c) {
    if ((hm[c] >> lane) & 1ull) {
      const int64_t i = base + (int64_t)(c * kHeadsWaves + wave) * kWave + lane;
      const int u = ccnt[c * kHeadsWaves + wave] + __popcll(hm[c] & ((1ull << lane) - 1ull));
      uniq[u] = (int64_t)kk[c];
      seg_off[u] = (int32_t)i;
    }
  }
}

static int key_bits(int64_t N) {
  int b = 1;
  while (b < 63 && (1ll << b) <= N) ++b;  // keys take values 0..N (N = sentinel)
  return b;
}

template <class KeyT>
struct GroupPlan {
  rsort::Plan sort;
  size_t off_keys_tmp, off_keys_dst, off_vals_tmp, off_hist, off_totals, off_nlive, off_cnt, total;
};

template <class KeyT>
static int plan_group(int64_t n, int64_t N, GroupPlan<KeyT>* p) {
  p->sort = rsort::make_plan(n, key_bits(N));
  size_t o = 0;
  p->off_keys_tmp = o; o += align_up((size_t)n * sizeof(KeyT), 256);
  p->off_keys_dst = o; o += align_up((size_t)n * sizeof(KeyT), 256);
  // (32-bit keys, >= 2 passes: the intermediate passes ping-pong PACKED pairs — pairs_a = the keys_tmp | keys_dst region,
  // pairs_b = this region, n uint2 — see rsort::sort_pairs_packed)
  p->off_vals_tmp = o; o += align_up((size_t)n * (sizeof(KeyT) == 4 ? 8 : sizeof(int32_t)), 256);
  p->off_hist = o;     o += p->sort.hist_bytes;
  p->off_totals = o;   o += p->sort.totals_bytes;
  p->off_nlive = o;    o += 256;
  p->off_cnt = o;      o += align_up((size_t)((n + kHeadsTile - 1) / kHeadsTile + 1) * 2 * sizeof(int32_t), 256);
  p->total = o;
  return REC_OK;
}

template <class KeyT>
static int run_group(int64_t n, int S, int64_t N, int64_t pad, const int64_t* ids,
                     const int64_t* slot_off, const int32_t* payload, int32_t* sorted_pos, int64_t* uniq,
                     int32_t* seg_off, int32_t* n_uniq, int32_t* status, void* ws, size_t ws_bytes,
                     hipStream_t st, int64_t slot_rows = 0) {
  GroupPlan<KeyT> p;
  if (int rc = plan_group<KeyT>(n, N, &p)) return rc;
  REC_REQUIRE(ws && ws_bytes >= p.total, REC_EWORKSPACE, "workspace %zu < %zu", ws_bytes, p.total);
  char* base = (char*)ws;
  KeyT* keys_tmp = (KeyT*)(base + p.off_keys_tmp);
  KeyT* keys_dst = (KeyT*)(base + p.off_keys_dst);
  int32_t* vals_tmp = (int32_t*)(base + p.off_vals_tmp);
  int32_t* cnt = (int32_t*)(base + p.off_cnt);
  // padding and out-of-range lookups leave the sort after its first pass (REC_GROUP_DROP=0: carried through every pass)
  static const bool drop = [] { const char* v = getenv("REC_GROUP_DROP"); return !(v && *v == '0'); }();
  const bool dr = drop && n < (1ll << 31);
  IdsSrc<KeyT> src{ids, slot_off, S, N, pad, status, payload, dr, slot_rows};
  // packed intermediate passes: measured on the gpubox model's 40 M lookups (sort alone 1.425 -> 1.361 ms, step -0.1 ms)
  // and on the 1.7 M lookups of a DeepFM batch on one shared table (step +8 us): taken from 4 M lookups on
  // (REC_RSORT_PACK=0 / 1: never / always)
  static const int pack = [] { const char* v = getenv("REC_RSORT_PACK"); return v && *v ? atoi(v) : -1; }();
  bool packed = false;
  if constexpr (sizeof(KeyT) == 4) {
    if (p.sort.passes >= 2 && (pack == 1 || (pack < 0 && n >= (4ll << 20)))) {
      packed = true;
      if (int rc = rsort::sort_pairs_packed(n, p.sort, src, (uint2*)keys_tmp, (uint2*)vals_tmp, (uint32_t*)keys_dst,
                                            sorted_pos, base + p.off_hist, base + p.off_totals, st,
                                            dr ? (int32_t*)(base + p.off_nlive) : nullptr, (uint32_t)N))
        return rc;
    }
  }
  if (!packed)
    if (int rc = rsort::sort_pairs<KeyT>(n, p.sort, src, keys_tmp, vals_tmp, keys_dst, sorted_pos,
                                         base + p.off_hist, base + p.off_totals, st,
                                         dr ? (int32_t*)(base + p.off_nlive) : nullptr, (KeyT)N))
      return rc;
  const int nblk = (int)((n + kHeadsTile - 1) / kHeadsTile);
  (void)hipMemsetAsync(n_uniq + 2, 0, 2 * sizeof(int32_t), st);
  hipLaunchKernelGGL(heads_count_kernel<KeyT>, dim3(nblk), dim3(rsort::kThreads), 0, st, n, (KeyT)N, keys_dst, cnt,
                     n_uniq);
  hipLaunchKernelGGL(heads_scan_kernel, dim3(1), dim3(rsort::kThreads), 0, st, nblk, cnt, seg_off, n_uniq);
  hipLaunchKernelGGL(heads_emit_kernel<KeyT>, dim3(nblk), dim3(rsort::kThreads), 0, st, n, (KeyT)N, keys_dst, cnt,
                     uniq, seg_off);
  return check_launch("rec_ids_group");
}

// element offset of the gradient row of lookup position `pos` (see rec_grad_layout)
__device__ __forceinline__ int64_t grad_offset(const rec_grad_layout& gl, int pos, int D) {
  const int p = gl.index ? gl.index[pos] : pos;   // multi-slot CSR: value k -> its (sample, slot) segment
  const int q = gl.div > 1 ? p / gl.div : p;
  return gl.group > 0 ? (int64_t)(q / gl.group) * gl.group_stride + (int64_t)(q % gl.group) * D
                      : (int64_t)q * D;
}
// ... of SORTED position k: the k-th row of grad when the producer wrote its rows in sorted order (rec_grad_layout.sorted:
// rec_deepfm_fm_bwd_sorted through the rank of rec_ids_group_slots) — consecutive segments then read consecutive
// memory and the dependent sorted_pos read drops out of the chain
__device__ __forceinline__ int64_t grad_at(const rec_grad_layout& gl, const int32_t* __restrict__ spos, int k, int D) {
  return gl.sorted ? (int64_t)k * D : grad_offset(gl, spos[k], D);
}


// ---------------------------------------------------------------- long-segment partial sums
// A hot row (Zipf-distributed ids: one row can own tens of thousands of the B*S lookups) would turn
// the per-row duplicate loop below into a serial chain of that many dependent HBM/L2 reads.
// rec_segment_partials pre-reduces, one wave per tile of REC_SEG_TILE sorted positions, the pieces of
// every segment of >= REC_SEG_LONG positions; the per-row loops then add one partial per tile instead
// of one gradient row per position.  partials[(tile*2 + slot)*D ..]: slot 0 = the piece of the long
// segment that contains the tile's first position, slot 1 = the piece of a different long segment
// that contains its last position (a long segment spans >= 2 tiles, so no tile holds a third piece).

template <int VEC, int LANES>
__global__ __launch_bounds__(kBlock) void segment_partials_kernel(
    int D, const int32_t* __restrict__ n_uniq, const int32_t* __restrict__ seg_off,
    const int32_t* __restrict__ spos, const float* __restrict__ grad, rec_grad_layout gl,
    float* __restrict__ partials) {
  static_assert(kSegTile == 64, "one wave per tile");
  constexpr int G = 64 / LANES;  // row groups per wave
  const int lane = threadIdx.x % 64;
  const int64_t tile = (int64_t)blockIdx.x * (kBlock / 64) + threadIdx.x / 64;
  const int U = n_uniq[0];
  if (U <= 0 || n_uniq[2] == 0) return;   // no long segment in this batch: nothing to pre-reduce
  const int nvalid = seg_off[U];
  const int64_t s64 = tile * kSegTile;
  if (s64 >= nvalid) return;
  const int s = (int)s64, e = min(s + kSegTile, nvalid);
  auto find = [&](int x) {  // the segment that holds sorted position x: 64-ary search, one probe per lane
    int lo = 0, hi = U;     // invariant: seg_off[lo] <= x < seg_off[hi]
    while (hi - lo > 1) {
      const int step = (hi - lo + 63) / 64;
      const int idx = lo + lane * step;
      const bool le = idx < hi && seg_off[idx] <= x;      // monotone over the lanes, lane 0 always true
      const int c = __popcll(__ballot(le));
      lo += (c - 1) * step;
      hi = min(lo + step, hi);
    }
    return lo;
  };
  const int grp = lane / LANES, d0 = (lane % LANES) * VEC;
  auto piece = [&](int a, int b, int slot) {
    float g[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) g[i] = 0.f;
    if (d0 < D)
      for (int k = a + grp; k < b; k += G) {
        float t[VEC];
        vload<VEC>(t, grad + grad_at(gl, spos, k, D) + d0);
#pragma unroll
        for (int i = 0; i < VEC; ++i) g[i] += t[i];
      }
#pragma unroll
    for (int off = LANES; off < 64; off <<= 1)
#pragma unroll
      for (int i = 0; i < VEC; ++i) g[i] += __shfl_xor(g[i], off, 64);
    if (grp == 0 && d0 < D) vstore<VEC>(partials + (tile * 2 + slot) * D + d0, g);
  };
  const int u0 = find(s);
  const int b0 = seg_off[u0], e0 = seg_off[u0 + 1];
  if (e0 - b0 >= kSegLong) piece(s, min(e, e0), 0);
  if (e0 < e) {
    const int u1 = find(e - 1);
    const int b1 = seg_off[u1];
    if (seg_off[u1 + 1] - b1 >= kSegLong) piece(b1, e, 1);
  }
}

// g += the gradient rows of sorted positions [beg,end) (ascending; four loads in flight), through the
// tile partials when the segment is long and the caller supplied them
template <int VEC>
__device__ __forceinline__ void segment_sum(float (&g)[VEC], int beg, int end,
                                            const int32_t* __restrict__ spos,
                                            const float* __restrict__ grad,
                                            const rec_grad_layout& gl, int D, int d0) {
  if (gl.partials && end - beg >= kSegLong) {
    const float* __restrict__ pp = gl.partials;
    const int t1 = (end - 1) / kSegTile;
    int t = beg / kSegTile;
    auto at = [&](int tt) {
      return pp + ((int64_t)tt * 2 + (beg <= tt * kSegTile ? 0 : 1)) * D + d0;
    };
    for (; t + 4 <= t1 + 1; t += 4) {
      float a[VEC], b[VEC], c[VEC], d[VEC];
      vload<VEC>(a, at(t)); vload<VEC>(b, at(t + 1)); vload<VEC>(c, at(t + 2)); vload<VEC>(d, at(t + 3));
#pragma unroll
      for (int i = 0; i < VEC; ++i) g[i] = (((g[i] + a[i]) + b[i]) + c[i]) + d[i];
    }
    for (; t <= t1; ++t) {
      float a[VEC];
      vload<VEC>(a, at(t));
#pragma unroll
      for (int i = 0; i < VEC; ++i) g[i] += a[i];
    }
    return;
  }
  int k = beg;
  if (end - beg >= 8)   // short rows (the common case) stay on the plain loop: no divergence inside a wave
  for (; k + 4 <= end; k += 4) {
    float a[VEC], b[VEC], c[VEC], d[VEC];
    vload<VEC>(a, grad + grad_at(gl, spos, k, D) + d0);
    vload<VEC>(b, grad + grad_at(gl, spos, k + 1, D) + d0);
    vload<VEC>(c, grad + grad_at(gl, spos, k + 2, D) + d0);
    vload<VEC>(d, grad + grad_at(gl, spos, k + 3, D) + d0);
#pragma unroll
    for (int i = 0; i < VEC; ++i) g[i] = (((g[i] + a[i]) + b[i]) + c[i]) + d[i];
  }
  for (; k < end; ++k) {
    float a[VEC];
    vload<VEC>(a, grad + grad_at(gl, spos, k, D) + d0);
#pragma unroll
    for (int i = 0; i < VEC; ++i) g[i] += a[i];
  }
}

// One Adam element update with the rounding points pinned (no compiler-chosen fma contraction): every Adam kernel
// of this file goes through it, so rec_sparse_adam_record is bit-identical to two rec_sparse_adam_rows passes and
// rec_adam_rows_all to the lazy kernel on the touched rows.
__device__ __forceinline__ float scale_grad(float g, float sc) {
#pragma clang fp contract(off)
  return g * sc;
}

__device__ __forceinline__ void adam_elem(float& p, float& m, float& v, float g, float lr_t, float eps_t,
                                          float b1, float b2) {
#pragma clang fp contract(off)
  const float m1 = b1 * m;
  const float m2 = (1.f - b1) * g;
  const float v1 = b2 * v;
  const float v2 = ((1.f - b2) * g) * g;
  m = m1 + m2;
  v = v1 + v2;
  const float den = sqrtf(v) + eps_t;
  const float step = lr_t * (m / den);
  p = p - step;
}

// --------------------------------------------------------------------------- lazy sparse Adam
template <int VEC, int LANES>
__global__ __launch_bounds__(kBlock) void sparse_adam_rows_kernel(
    int D, int stride, int sstride, const int32_t* __restrict__ n_uniq, const int64_t* __restrict__ uniq,
    const int32_t* __restrict__ seg_off, const int32_t* __restrict__ spos,
    const float* __restrict__ grad, rec_grad_layout gl, const float* __restrict__ grad_scale,
    float* __restrict__ P, float* __restrict__ M, float* __restrict__ V, float lr_t, float eps_t,
    float b1, float b2) {
  const int64_t u = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / LANES;
  const int d0 = (threadIdx.x % LANES) * VEC;
  if (u >= n_uniq[0] || d0 >= D) return;
  const int64_t row = uniq[u];
  const int beg = seg_off[u], end = seg_off[u + 1];
  float p[VEC], m[VEC], v[VEC], g[VEC];
  const int64_t ro = row * stride + d0;
  const int64_t so = row * sstride + d0;
  vload<VEC>(p, P + ro);
  vload<VEC>(m, M + so);
  vload<VEC>(v, V + so);
#pragma unroll
  for (int i = 0; i < VEC; ++i) g[i] = 0.f;
  segment_sum<VEC>(g, beg, end, spos, grad, gl, D, d0);
  if (grad_scale) {   // global-norm clipping factor (device scalar)
    const float sc = grad_scale[0];
#pragma unroll
    for (int i = 0; i < VEC; ++i) g[i] = scale_grad(g[i], sc);
  }
#pragma unroll
  for (int i = 0; i < VEC; ++i) adam_elem(p[i], m[i], v[i], g[i], lr_t, eps_t, b1, b2);
  vstore<VEC>(P + ro, p);
  vstore<VEC>(M + so, m);
  vstore<VEC>(V + so, v);
}

// ------------------------------------------------------------- narrow rows: one LANE per touched row
// Rows whose width is not a multiple of 4 floats (slot_dnn: D 9, deepfm/config.yaml: D 9 / 10) have no float4 row
// groups: the kernel above then spends 16 lanes on a 9-float row — 4 rows per wave — and each wave walks the chain
// uniq -> seg_off -> sorted_pos -> grad_index -> gradient row -> P, M, V -> store one dependent memory round trip
// after the other: 9 M touched rows took 3.1 ms (profiles/r02c_slot_dnn_kernel_stats.csv), five times the time of
// their bytes.  Here a lane owns a whole row (<= 16 floats in registers): 64 rows per wave are in flight, the index
// chain is read coalesced across the lanes, and the row's own 36-64 bytes are the lane's private line.  Same
// arithmetic (adam_elem), same ascending-position summation order.
template <int NV, bool V4>
__device__ __forceinline__ void narrow_load(float (&x)[NV * 4], const float* __restrict__ p, int D) {
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    if (V4 && c * 4 + 4 <= D) {
      const float4 t = *reinterpret_cast<const float4*>(p + c * 4);
      x[c * 4] = t.x; x[c * 4 + 1] = t.y; x[c * 4 + 2] = t.z; x[c * 4 + 3] = t.w;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) x[c * 4 + i] = c * 4 + i < D ? p[c * 4 + i] : 0.f;
    }
  }
}
template <int NV, bool V4>
__device__ __forceinline__ void narrow_store(float* __restrict__ p, const float (&x)[NV * 4], int D) {
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    if (V4 && c * 4 + 4 <= D) {
      *reinterpret_cast<float4*>(p + c * 4) = make_float4(x[c * 4], x[c * 4 + 1], x[c * 4 + 2], x[c * 4 + 3]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (c * 4 + i < D) p[c * 4 + i] = x[c * 4 + i];
    }
  }
}

// x[0..D) = one narrow gradient row (zeros behind D).  Whole groups of four floats as ONE dwordx4 load at dword alignment
// (gfx950 global loads take it; a 9-float row is two of them and a dword instead of nine dwords — every load instruction
// of a lane-per-row kernel is an L2 round trip of its own: 64 different lines per wave instruction do not live in L1).
template <int NV>
__device__ __forceinline__ void narrow_row_load(float (&x)[NV * 4], const float* __restrict__ a, int D) {
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    if (4 * q + 4 <= D) {
      const f4u t = *reinterpret_cast<const f4u*>(a + 4 * q);
      x[4 * q] = t.x; x[4 * q + 1] = t.y; x[4 * q + 2] = t.z; x[4 * q + 3] = t.w;
    } else {
#pragma unroll
      for (int d = 4 * q; d < 4 * q + 4; ++d) x[d] = d < D ? a[d] : 0.f;
    }
  }
}

// g[0..D) += the gradient rows of sorted positions [beg,end), ascending (two positions in flight); long segments
// through the tile partials as segment_sum does
template <int NV>
__device__ __forceinline__ void narrow_segment_sum(float (&g)[NV * 4], int beg, int end,
                                                   const int32_t* __restrict__ spos,
                                                   const float* __restrict__ grad, const rec_grad_layout& gl,
                                                   int D) {
  if (gl.partials && end - beg >= kSegLong) {
    const float* __restrict__ pp = gl.partials;
    const int t1 = (end - 1) / kSegTile;
    for (int t = beg / kSegTile; t <= t1; ++t) {
      const float* a = pp + ((int64_t)t * 2 + (beg <= t * kSegTile ? 0 : 1)) * D;
#pragma unroll
      for (int d = 0; d < NV * 4; ++d) g[d] += d < D ? a[d] : 0.f;
    }
    return;
  }
  int k = beg;
  for (; k + 2 <= end; k += 2) {
    const float* a = grad + grad_at(gl, spos, k, D);
    const float* b = grad + grad_at(gl, spos, k + 1, D);
    float x[NV * 4], y[NV * 4];
    narrow_row_load<NV>(x, a, D);
    narrow_row_load<NV>(y, b, D);
#pragma unroll
    for (int d = 0; d < NV * 4; ++d) g[d] = (g[d] + x[d]) + y[d];
  }
  if (k < end) {
    float x[NV * 4];
    narrow_row_load<NV>(x, grad + grad_at(gl, spos, k, D), D);
#pragma unroll
    for (int d = 0; d < NV * 4; ++d) g[d] += x[d];
  }
}

template <int NV, bool V4>
__global__ __launch_bounds__(kBlock) void sparse_adam_rows_narrow_kernel(
    int D, int stride, int sstride, const int32_t* __restrict__ n_uniq, const int64_t* __restrict__ uniq,
    const int32_t* __restrict__ seg_off, const int32_t* __restrict__ spos,
    const float* __restrict__ grad, rec_grad_layout gl, const float* __restrict__ grad_scale,
    float* __restrict__ P, float* __restrict__ M, float* __restrict__ V, float lr_t, float eps_t,
    float b1, float b2) {
  // the launch is sized for n_max (a capacity: the number of distinct rows is only known on the device), so the
  // grid is capped at a few resident rounds and strides over the rows that exist
  const int nu = n_uniq[0];
  for (int64_t u = (int64_t)blockIdx.x * kBlock + threadIdx.x; u < nu; u += (int64_t)gridDim.x * kBlock) {
  const int64_t row = uniq[u];
  const int beg = seg_off[u], end = seg_off[u + 1];
  float p[NV * 4], m[NV * 4], v[NV * 4], g[NV * 4];
  narrow_load<NV, V4>(p, P + row * stride, D);
  narrow_load<NV, V4>(m, M + row * sstride, D);
  narrow_load<NV, V4>(v, V + row * sstride, D);
#pragma unroll
  for (int d = 0; d < NV * 4; ++d) g[d] = 0.f;
  narrow_segment_sum<NV>(g, beg, end, spos, grad, gl, D);
  const float sc = grad_scale ? grad_scale[0] : 1.f;
#pragma unroll
  for (int d = 0; d < NV * 4; ++d) {
    if (d < D) {
      if (grad_scale) g[d] = scale_grad(g[d], sc);
      adam_elem(p[d], m[d], v[d], g[d], lr_t, eps_t, b1, b2);
    }
  }
  narrow_store<NV, V4>(P + row * stride, p, D);
  narrow_store<NV, V4>(M + row * sstride, m, D);
  narrow_store<NV, V4>(V + row * sstride, v, D);
  }
}

// Both embeddings of a DeepFM row in ONE pass (DESIGN.md "table layout"): the record line holds
//   rec [N, stride] = W(D) | W1 | m1 | v1 | pad      and      mv [N, sstride] = m(D) | v(D)
// so the first-order table (deepfm/net.py:62-70 `embedding_one`) costs no extra HBM line: the row group that
// updates W/m/v also updates W1/m1/v1 (lane 0 of the group, same 128-B record line) from the first-order
// gradient dz[pos / S].  Replaces two launches of sparse_adam_rows_kernel (the second one re-read and re-wrote
// the record line for 12 useful bytes).
template <int VEC, int LANES>
__global__ __launch_bounds__(kBlock) void sparse_adam_record_kernel(
    int D, int stride, int sstride, int v_off, const int32_t* __restrict__ n_uniq,
    const int64_t* __restrict__ uniq, const int32_t* __restrict__ seg_off,
    const int32_t* __restrict__ spos, const float* __restrict__ grad, rec_grad_layout gl,
    const float* __restrict__ grad1, rec_grad_layout gl1, const float* __restrict__ grad_scale,
    float* __restrict__ rec, float* __restrict__ MV, float lr_t, float eps_t, float b1, float b2, int nt) {
  const int64_t u = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / LANES;
  const int lg = threadIdx.x % LANES;
  const int d0 = lg * VEC;
  if (u >= n_uniq[0]) return;
  const int64_t row = uniq[u];
  const int beg = seg_off[u], end = seg_off[u + 1];
  const float sc = grad_scale ? grad_scale[0] : 1.f;
  float* r = rec + row * stride;
  if (d0 < D) {
    float p[VEC], m[VEC], v[VEC], g[VEC];
    float* mv = MV + row * sstride;
    vload<VEC>(p, r + d0);
    vload<VEC>(m, mv + d0);
    vload<VEC>(v, mv + v_off + d0);
#pragma unroll
    for (int i = 0; i < VEC; ++i) g[i] = 0.f;
    segment_sum<VEC>(g, beg, end, spos, grad, gl, D, d0);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      if (grad_scale) g[i] = scale_grad(g[i], sc);
      adam_elem(p[i], m[i], v[i], g[i], lr_t, eps_t, b1, b2);
    }
    // nt: the updated lines are not wanted in L2 / Infinity Cache (the next reader is some later step's gather
    // of a random subset) — streamed out, the next kernel does not pay for evicting them (profiles/r03_fm_instep.txt)
    if (nt) {
      vstore_nt<VEC>(r + d0, p);
      vstore_nt<VEC>(mv + d0, m);
      vstore_nt<VEC>(mv + v_off + d0, v);
    } else {
      vstore<VEC>(r + d0, p);
      vstore<VEC>(mv + d0, m);
      vstore<VEC>(mv + v_off + d0, v);
    }
  }
  if (lg == 0) {   // first-order weight + its moments: rec[D], rec[D+1], rec[D+2]
    // (VEC 4: D and the stride are multiples of 4 — the three floats and the pad float behind them are ONE aligned
    // float4: one load and one store instead of three of each)
    float q[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (VEC == 4) vload<4>(q, r + D);
    else { q[0] = r[D]; q[1] = r[D + 1]; q[2] = r[D + 2]; }
    float g1[1] = {0.f};
    segment_sum<1>(g1, beg, end, spos, grad1, gl1, 1, 0);
    const float g = grad_scale ? scale_grad(g1[0], sc) : g1[0];
    adam_elem(q[0], q[1], q[2], g, lr_t, eps_t, b1, b2);
    if constexpr (VEC == 4) vstore<4>(r + D, q);
    else { r[D] = q[0]; r[D + 1] = q[1]; r[D + 2] = q[2]; }
  }
}

// paddle.optimizer.Adam with lazy_mode=False on a SelectedRows gradient (the dygraph default,
// deepfm/dygraph_model.py:61-65): EVERY row's moments decay and every row moves, rows absent
// from the merged gradient use g = 0.  One pass over the whole table (6*N*D*4 B of traffic — the reason the
// engine defaults to the lazy variant); a block owns kBlock/LANES consecutive rows, finds the slice of the
// sorted unique-row list that falls inside with two binary searches and spreads it into an LDS slot map.
template <int VEC, int LANES>
__global__ __launch_bounds__(kBlock) void adam_rows_all_kernel(
    int64_t N, int D, int stride, int sstride, const int32_t* __restrict__ n_uniq,
    const int64_t* __restrict__ uniq, const int32_t* __restrict__ seg_off,
    const int32_t* __restrict__ spos, const float* __restrict__ grad, rec_grad_layout gl,
    const float* __restrict__ grad_scale, float* __restrict__ P, float* __restrict__ M,
    float* __restrict__ V, float lr_t, float eps_t, float b1, float b2) {
  constexpr int RB = kBlock / LANES;
  __shared__ int slot[RB];
  __shared__ int range[2];
  const int64_t r0 = (int64_t)blockIdx.x * RB;
  for (int i = threadIdx.x; i < RB; i += kBlock) slot[i] = -1;
  if (threadIdx.x < 2) {
    const int64_t key = r0 + (threadIdx.x ? RB : 0);
    int lo = 0, hi = n_uniq[0];
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (uniq[mid] < key) lo = mid + 1; else hi = mid;
    }
    range[threadIdx.x] = lo;
  }
  __syncthreads();
  for (int u = range[0] + threadIdx.x; u < range[1]; u += kBlock) slot[(int)(uniq[u] - r0)] = u;
  __syncthreads();
  const int lr_ = threadIdx.x / LANES;
  const int64_t row = r0 + lr_;
  const int d0 = (threadIdx.x % LANES) * VEC;
  if (row >= N || d0 >= D) return;
  float p[VEC], m[VEC], v[VEC], g[VEC];
  const int64_t ro = row * stride + d0, so = row * sstride + d0;
  vload<VEC>(p, P + ro);
  vload<VEC>(m, M + so);
  vload<VEC>(v, V + so);
#pragma unroll
  for (int i = 0; i < VEC; ++i) g[i] = 0.f;
  const int u = slot[lr_];
  if (u >= 0) {
    segment_sum<VEC>(g, seg_off[u], seg_off[u + 1], spos, grad, gl, D, d0);
    if (grad_scale) {
      const float sc = grad_scale[0];
#pragma unroll
      for (int i = 0; i < VEC; ++i) g[i] *= sc;
    }
  }
#pragma unroll
  for (int i = 0; i < VEC; ++i) adam_elem(p[i], m[i], v[i], g[i], lr_t, eps_t, b1, b2);
  vstore<VEC>(P + ro, p);
  vstore<VEC>(M + so, m);
  vstore<VEC>(V + so, v);
}

// The same on the record layout of sparse_adam_record_kernel — rec [N, stride] = W(D) | W1 | m1 | v1 | pad, MV [N, sstride] =
// m(D) | v(D) at v_off — so that the dygraph-default optimizer touches a row's two lines once: two adam_rows_all passes
// (W, then W1 / m1 / v1 as three strided 4-byte columns of the SAME record line) move 768 B per row, this one 512
// (26 M rows x D 16: 20 GB -> 13.3 GB per step).  Same arithmetic and summation order as the two passes: bit-identical.
template <int VEC, int LANES>
__global__ __launch_bounds__(kBlock) void adam_record_all_kernel(
    int64_t N, int D, int stride, int sstride, int v_off, const int32_t* __restrict__ n_uniq,
    const int64_t* __restrict__ uniq, const int32_t* __restrict__ seg_off, const int32_t* __restrict__ spos,
    const float* __restrict__ grad, rec_grad_layout gl, const float* __restrict__ grad1, rec_grad_layout gl1,
    const float* __restrict__ grad_scale, float* __restrict__ rec, float* __restrict__ MV, float lr_t, float eps_t,
    float b1, float b2) {
  constexpr int RB = kBlock / LANES;
  __shared__ int slot[RB];
  __shared__ int range[2];
  const int64_t r0 = (int64_t)blockIdx.x * RB;
  for (int i = threadIdx.x; i < RB; i += kBlock) slot[i] = -1;
  if (threadIdx.x < 2) {
    const int64_t key = r0 + (threadIdx.x ? RB : 0);
    int lo = 0, hi = n_uniq[0];
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (uniq[mid] < key) lo = mid + 1; else hi = mid;
    }
    range[threadIdx.x] = lo;
  }
  __syncthreads();
  for (int u = range[0] + threadIdx.x; u < range[1]; u += kBlock) slot[(int)(uniq[u] - r0)] = u;
  __syncthreads();
  const int lr_ = threadIdx.x / LANES, lg = threadIdx.x % LANES;
  const int64_t row = r0 + lr_;
  const int d0 = lg * VEC;
  if (row >= N) return;
  const int u = slot[lr_];
  const float sc = grad_scale ? grad_scale[0] : 1.f;
  float* r = rec + row * stride;
  if (d0 < D) {
    float p[VEC], m[VEC], v[VEC], g[VEC];
    float* mv = MV + row * sstride;
    vload<VEC>(p, r + d0);
    vload<VEC>(m, mv + d0);
    vload<VEC>(v, mv + v_off + d0);
#pragma unroll
    for (int i = 0; i < VEC; ++i) g[i] = 0.f;
    if (u >= 0) {
      segment_sum<VEC>(g, seg_off[u], seg_off[u + 1], spos, grad, gl, D, d0);
      if (grad_scale) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) g[i] *= sc;
      }
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) adam_elem(p[i], m[i], v[i], g[i], lr_t, eps_t, b1, b2);
    vstore_nt<VEC>(r + d0, p);                       // the whole table streams through: nothing of it is wanted in L2
    vstore_nt<VEC>(mv + d0, m);
    vstore_nt<VEC>(mv + v_off + d0, v);
  }
  if (lg == 0) {
    float q[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (VEC == 4) vload<4>(q, r + D);
    else { q[0] = r[D]; q[1] = r[D + 1]; q[2] = r[D + 2]; }
    float g1[1] = {0.f};
    if (u >= 0) {
      segment_sum<1>(g1, seg_off[u], seg_off[u + 1], spos, grad1, gl1, 1, 0);
      if (grad_scale) g1[0] *= sc;
    }
    adam_elem(q[0], q[1], q[2], g1[0], lr_t, eps_t, b1, b2);
    if constexpr (VEC == 4) vstore_nt<4>(r + D, q);
    else { r[D] = q[0]; r[D + 1] = q[1]; r[D + 2] = q[2]; }
  }
}

// paddle.optimizer.SGD on the touched rows (din/dygraph_model.py:64-73; rows with zero gradient do not move,
// so updating only the merged rows IS dense SGD): p -= lr * sum of the row's duplicate gradients
template <int VEC, int LANES>
__global__ __launch_bounds__(kBlock) void sparse_sgd_rows_kernel(
    int D, int stride, const int32_t* __restrict__ n_uniq, const int64_t* __restrict__ uniq,
    const int32_t* __restrict__ seg_off, const int32_t* __restrict__ spos,
    const float* __restrict__ grad, rec_grad_layout gl, float* __restrict__ P, float lr) {
  const int64_t u = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / LANES;
  const int d0 = (threadIdx.x % LANES) * VEC;
  if (u >= n_uniq[0] || d0 >= D) return;
  float p[VEC], g[VEC];
  const int64_t ro = uniq[u] * stride + d0;
  vload<VEC>(p, P + ro);
#pragma unroll
  for (int i = 0; i < VEC; ++i) g[i] = 0.f;
  segment_sum<VEC>(g, seg_off[u], seg_off[u + 1], spos, grad, gl, D, d0);
#pragma unroll
  for (int i = 0; i < VEC; ++i) p[i] -= lr * g[i];
  vstore<VEC>(P + ro, p);
}

__global__ void sgd_dense_kernel(int64_t n, float* __restrict__ p, const float* __restrict__ g, float lr) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    p[i] -= lr * g[i];
}

// ------------------------------------------------------------------ small batches: merge + SGD in ONE launch
// At the reference's own batch size (din/config.yaml: 32 samples x ~150 history positions = ~5 k lookups per table)
// the grouping sort + hot-row partials + row update of one table are 12-13 launches of a few microseconds each,
// seven tables per step: ~90 of the ~120 launches of a DIN train step, and the launches — not the work — are the
// step time (0.70 ms; replaying the same launches from a hipGraph: 0.60 ms).  For n this small the merge needs no
// sort: a wave per lookup compares its id with all n ids 64 at a time (ballots over the id list, which sits in L2);
// the wave of a row's FIRST occurrence adds the gradient rows of every occurrence in ascending position (the
// SelectedRows merge order) and applies p -= lr * g.  O(n^2 / 64) wave steps: 12 us at n = 8192.
constexpr int kSmallMergeMax = 15360;   // ids + per-wave lists stay inside the default 64 KB of LDS
constexpr int kSmallWaves = 16;        // lookups per block; the block stages the whole id list in LDS once (4 B per id)
constexpr int kSmallList = 32;         // occurrences of a row collected before their gradient rows are fetched together

// What the wave of a row's first occurrence does with the merged gradient: SGD on the row, or lazy Adam on BOTH
// embeddings of a DeepFM record (W, m, v and W1, m1, v1 — the record kernel's arithmetic, adam_elem).
struct SmallSgd {
  float* P;
  int stride;
  float lr;
};
struct SmallAdamRecord {
  float* rec;      // [N, stride]  = W(D) | W1 | m1 | v1 | pad
  float* MV;       // [N, sstride] = m(D) at 0, v(D) at v_off
  int stride, sstride, v_off;
  const float* grad1;          // first-order gradient source (dz), layout gl1
  rec_grad_layout gl1;
  const float* grad_scale;     // device scalar or null
  float lr_t, eps_t, b1, b2;
};

template <int NACC, class Update>   // floats per lane: D <= 64 * NACC
__device__ __forceinline__ void sparse_small_body(
    int blk, int n, int D, int S, int64_t N, int64_t pad, const int64_t* __restrict__ ids,
    const int64_t* __restrict__ slot_off, const float* __restrict__ grad, rec_grad_layout gl, Update up,
    int32_t* __restrict__ status, const int32_t* __restrict__ span_flag = nullptr, int fp_mode = 0) {
  constexpr bool kRecord = std::is_same<Update, SmallAdamRecord>::value;
  constexpr int FLY = kSmallList / NACC;   // gradient rows in flight per wave (32 / 16 / 8 rows of 64 / 128 / 256 floats)
  extern __shared__ __attribute__((aligned(16))) int small_lds[];
  int* wlists = small_lds;                               // [waves][kSmallList] positions of collected occurrences
  int* small_ids = small_lds + kSmallWaves * kSmallList; // [n] rows as int32, -1 = padding / out of range (never matches)
  // fp_mode (plain tables, small_lds_bytes): 16-bit fingerprints of the rows behind the id list, 512 positions per chunk and
  // position j of a chunk at half (j % 64) * 8 + j / 64 — one ds_read_b128 per lane then covers 512 positions (lane l holds
  // positions l, 64 + l, ..., 448 + l), and a chunk without a fingerprint match costs that read, a dozen VALU ops and a
  // ballot instead of eight reads and two ballots; candidates are confirmed against small_ids
  unsigned short* fps = reinterpret_cast<unsigned short*>(small_ids + ((n + 3) & ~3));
  const int lane = threadIdx.x % kWave;
  const bool slots = slot_off != nullptr && S > 1 && n % S == 0;
  const bool fp = fp_mode != 0 && !slots;
  auto fp_of = [](int r) { return (unsigned short)((r ^ (r >> 16)) & 0xffff); };
  auto fp_slot = [](int j) { return (j & ~511) | ((j & 63) << 3) | ((j >> 6) & 7); };
  int oob = 0, viol = 0;
  bool by_slot = false;
  int pos = blk * kSmallWaves + threadIdx.x / kWave;       // the lookup of this wave (re-mapped below in slot mode)
  int ts = 1, t0 = 0, tpos = pos, tn = n;                  // logical index in the compared list -> position t0 + j * ts
  if (slots) {
    // Slots whose ids stay inside their own span of rows cannot share a row with another slot: a lookup is then compared
    // with the lookups of ITS slot only (n / S of them: 512 instead of 13312 at the reference's batch size).  Every
    // block takes the decision on ALL ids — a plain scan, nothing staged: with the whole list staged per block the launch
    // cost grew with n^2 (48 us at 512 x 26 lookups, 40 of them staging) — and stages only the ids of the ONE slot its 16
    // waves belong to (blocks are dealt slot by slot).
    if (span_flag) {        // small_span_check_kernel looked at every id once for the whole launch
      by_slot = span_flag[0] == 0;
    } else {
      int sl = (int)(threadIdx.x % S);
      const int step = (kSmallWaves * kWave) % S;
      for (int i = threadIdx.x; i < n; i += kSmallWaves * kWave) {
        const int64_t id = ids[i];
        const bool isp = pad >= 0 && id == pad;
        const int64_t r = id + slot_off[sl];
        oob |= (!isp && !(r >= 0 && r < N)) ? 1 : 0;
        if (id < 0 || (sl + 1 < S && id >= slot_off[sl + 1] - slot_off[sl])) viol = 1;
        sl += step;
        if (sl >= S) sl -= S;
      }
      by_slot = __syncthreads_or(viol) == 0;
    }
  }
  if (by_slot) {
    tn = n / S;
    const int bps = (tn + kSmallWaves - 1) / kSmallWaves;   // blocks per slot
    t0 = blk / bps;                                         // this block's slot
    ts = S;
    tpos = (blk % bps) * kSmallWaves + threadIdx.x / kWave;
    pos = t0 + tpos * ts;
    if (t0 >= S) return;                                    // (grid = max of the two mappings)
    const int64_t off = slot_off[t0];
    for (int j = threadIdx.x; j < tn; j += kSmallWaves * kWave) {
      const int64_t id = ids[(int64_t)j * S + t0];
      const int64_t r = id + off;
      const bool isp = pad >= 0 && id == pad, inr = r >= 0 && r < N;
      if (span_flag) oob |= (!isp && !inr) ? 1 : 0;       // (every slot's ids pass through its blocks' staging)
      small_ids[j] = (!isp && inr) ? (int)r : -1;
    }
  } else {
    auto stage = [&](int i, int64_t id) {
      const bool isp = pad >= 0 && id == pad;
      const int64_t r = slot_off ? id + slot_off[i % S] : id;     // 26 tables as one: row = id + slot offset
      const bool inr = r >= 0 && r < N;
      if (!slots || span_flag) oob |= (!isp && !inr) ? 1 : 0;
      const int row = (!isp && inr) ? (int)r : -1;
      small_ids[i] = row;
      if (fp) fps[fp_slot(i)] = fp_of(row);
    };
    constexpr int NT = kSmallWaves * kWave;
    if (!slot_off && (reinterpret_cast<uintptr_t>(ids) & 15) == 0) {
      // a plain table: two ids per 16-byte load and four loads in flight per thread (every block of the launch stages the
      // whole list: one L2 round trip instead of one per 1024 ids)
      const longlong2* ids2 = reinterpret_cast<const longlong2*>(ids);
      const int np = n >> 1;
      for (int b0 = 0; b0 < np; b0 += 4 * NT) {
        longlong2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int q = b0 + u * NT + (int)threadIdx.x;
          v[u] = ids2[q < np ? q : 0];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int q = b0 + u * NT + (int)threadIdx.x;
          if (q < np) {
            stage(2 * q, v[u].x);
            stage(2 * q + 1, v[u].y);
          }
        }
      }
      if ((n & 1) && threadIdx.x == 0) stage(n - 1, ids[n - 1]);
    } else {
      for (int i = threadIdx.x; i < n; i += NT) stage(i, ids[i]);
    }
    if (fp)      // the rest of the last chunk: a fingerprint that is confirmed against nothing (positions >= n)
      for (int j = n + (int)threadIdx.x; j < ((n + 511) & ~511); j += NT) fps[fp_slot(j)] = 0;
    if (blk >= (n + kSmallWaves - 1) / kSmallWaves) pos = n;   // a block of the slot mapping only: nothing to do
  }
  if (oob) atomicOr(status, REC_FLAG_INDEX_OOB);
  __syncthreads();
  if (tpos >= tn || pos >= n) return;
  const int my = small_ids[tpos];
  if (my < 0) return;
  // an earlier occurrence owns the row (four 64-id chunks per trip: the LDS reads of a trip are in flight together).
  // Searched from the NEAREST earlier lookups backwards: only existence matters, and where duplicates come in runs (DIN's
  // target-seq tables: one row repeated over a sample's whole history) the first trip already finds one — searched from
  // position 0 the lookups of the last sample walked the whole list first.
  const unsigned myfp = fp_of(my), rep = myfp * 0x00010001u;
  // does this lane's eight fingerprints of chunk ck hold myfp?  (zero-half test of w ^ rep, exact for "any")
  auto chunk_any = [&](int ck, uint4& w) {
    w = *reinterpret_cast<const uint4*>(fps + ck * 512 + lane * 8);
    const unsigned x0 = w.x ^ rep, x1 = w.y ^ rep, x2 = w.z ^ rep, x3 = w.w ^ rep;
    const unsigned z = ((x0 - 0x00010001u) & ~x0) | ((x1 - 0x00010001u) & ~x1) | ((x2 - 0x00010001u) & ~x2) |
                       ((x3 - 0x00010001u) & ~x3);
    return (z & 0x80008000u) != 0u;
  };
  auto half_of = [](const uint4& w, int h) {
    const unsigned d = h < 2 ? w.x : h < 4 ? w.y : h < 6 ? w.z : w.w;
    return (d >> (16 * (h & 1))) & 0xffffu;
  };
  if (fp) {
    for (int ck = tpos >> 9; ck >= 0; --ck) {
      uint4 w;
      if (__ballot(chunk_any(ck, w)) == 0) continue;
      bool hit = false;
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        const int j = ck * 512 + h * kWave + lane;
        if (half_of(w, h) == myfp && j < tpos) hit |= small_ids[j] == my;
      }
      if (__ballot(hit) != 0) return;
    }
  } else {
    for (int c1 = tpos; c1 > 0; c1 -= 4 * kWave) {
      const int c0 = c1 - 4 * kWave;
      bool hit = false;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = c0 + u * kWave + lane;
        hit |= j >= 0 && j < tpos && small_ids[j >= 0 ? j : 0] == my;
      }
      if (__ballot(hit) != 0) return;
    }
  }
  // The occurrences (ascending) are collected FLY at a time across the 64-id chunks and their gradient rows fetched
  // together: a row with a handful of scattered duplicates costs one memory round trip, a hot row (the target item
  // of a sample: one occurrence per history position) ceil(occurrences / FLY).
  int* wl = wlists + (threadIdx.x / kWave) * kSmallList;
  float acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a) acc[a] = 0.f;
  float acc1 = 0.f;                                      // first-order gradient of the row (record update)
  int cnt = 0;
  const bool plain = gl.group == 1 && gl.div == 1 && gl.index == nullptr;
  // The gradient rows of the collected occurrences are fetched in straight-line code, in tiers of the count (1, 4, 16,
  // FLY rows; slots behind the count re-read the first row and add nothing): with a wave-uniform
  // `if (u < cnt)` around every fetch the compiler drained each load at the branch's join — a hot row's FLY fetches were
  // FLY memory round trips (tools/isa_load_waits.py: 387 of the kernel's 774 loads).  Rows are added in ascending order.
  bool no_index = gl.index == nullptr;
  if constexpr (kRecord) no_index = no_index && up.gl1.index == nullptr;
  auto fetch = [&](auto tier, auto noidx) {
    constexpr int TR = decltype(tier)::value;
    rec_grad_layout gq = gl;
    if constexpr (decltype(noidx)::value) gq.index = nullptr;      // folds the position -> index load out of the tier
    float x[TR][NACC], x1[TR];
#pragma unroll
    for (int u = 0; u < TR; ++u) {
      const int q = wl[u < cnt ? u : 0];
      // (one gradient row per position at a fixed pitch is the common layout: no integer division)
      const float* g = grad + (plain ? (int64_t)q * gq.group_stride : grad_offset(gq, q, D));
#pragma unroll
      for (int a = 0; a < NACC; ++a) {
        const int d = lane + a * kWave;
        x[u][a] = g[d < D ? d : 0];
      }
      if constexpr (kRecord) {
        rec_grad_layout g1 = up.gl1;
        if constexpr (decltype(noidx)::value) g1.index = nullptr;
        x1[u] = up.grad1[grad_offset(g1, q, 1)];
      } else {
        x1[u] = 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < TR; ++u) {
      const bool live = u < cnt;                       // wave-uniform
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] += (live && lane + a * kWave < D) ? x[u][a] : 0.f;
      acc1 += live ? x1[u] : 0.f;
    }
  };
  auto flush = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (no_index) {
      if (cnt <= 1) fetch(std::integral_constant<int, 1>{}, std::true_type{});
      else if (cnt <= 4) fetch(std::integral_constant<int, (FLY < 4 ? FLY : 4)>{}, std::true_type{});
      else if (cnt <= 16) fetch(std::integral_constant<int, (FLY < 16 ? FLY : 16)>{}, std::true_type{});
      else fetch(std::integral_constant<int, FLY>{}, std::true_type{});
    } else {
      fetch(std::integral_constant<int, FLY>{}, std::false_type{});      // (position -> index -> row: a dependent chain anyway)
    }
    cnt = 0;
    __builtin_amdgcn_wave_barrier();     // the list is rewritten only after every lane has read it
  };
  // (four chunks per trip, like the search above: most rows have no second occurrence, and a trip without a match
  // is four LDS reads in flight, one ballot)
  // the occurrences among 64 consecutive positions (lane = position j), appended in ascending order
  auto emit = [&](bool mine, int j) {
    unsigned long long m = __ballot(mine);
    while (m) {
      const int rank = __popcll(m & ((1ull << lane) - 1ull));
      const int take = min((int)__popcll(m), FLY - cnt);
      if (mine && rank < take) wl[cnt + rank] = t0 + j * ts;
      cnt += take;
      mine = mine && rank >= take;
      m = __ballot(mine);
      if (cnt == FLY) flush();
    }
  };
  if (fp) {
    for (int ck = tpos >> 9; ck * 512 < tn; ++ck) {
      uint4 w;
      if (__ballot(chunk_any(ck, w)) == 0) continue;
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        const int j = ck * 512 + h * kWave + lane;
        emit(half_of(w, h) == myfp && j >= tpos && j < tn && small_ids[j] == my, j);
      }
    }
  } else {
    for (int c0 = (tpos / kWave) * kWave; c0 < tn; c0 += 4 * kWave) {
      bool mk[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = c0 + u * kWave + lane;
        mk[u] = j >= tpos && j < tn && small_ids[j < tn ? j : 0] == my;
      }
      if (__ballot(mk[0] || mk[1] || mk[2] || mk[3]) == 0) continue;
#pragma unroll
      for (int u = 0; u < 4; ++u) emit(mk[u], c0 + u * kWave + lane);
    }
  }
  if (cnt > 0) flush();
  if constexpr (!kRecord) {
    float* p = up.P + (int64_t)my * up.stride;
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
      const int d = lane + a * kWave;
      if (d < D) p[d] -= up.lr * acc[a];
    }
  } else {
    float* r = up.rec + (int64_t)my * up.stride;
    float* mv = up.MV + (int64_t)my * up.sstride;
    const float sc = up.grad_scale ? up.grad_scale[0] : 1.f;
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
      const int d = lane + a * kWave;
      if (d < D) {
        float p = r[d], m_ = mv[d], v_ = mv[up.v_off + d];
        const float g = up.grad_scale ? scale_grad(acc[a], sc) : acc[a];
        adam_elem(p, m_, v_, g, up.lr_t, up.eps_t, up.b1, up.b2);
        r[d] = p; mv[d] = m_; mv[up.v_off + d] = v_;
      }
    }
    if (lane == 0) {   // first-order weight + its moments: rec[D], rec[D+1], rec[D+2]
      float p1 = r[D], m1 = r[D + 1], v1 = r[D + 2];
      const float g = up.grad_scale ? scale_grad(acc1, sc) : acc1;
      adam_elem(p1, m1, v1, g, up.lr_t, up.eps_t, up.b1, up.b2);
      r[D] = p1; r[D + 1] = m1; r[D + 2] = v1;
    }
  }
}

template <int NACC, class Update>
__global__ __launch_bounds__(kSmallWaves* kWave) void sparse_small_kernel(
    int n, int D, int S, int64_t N, int64_t pad, const int64_t* __restrict__ ids,
    const int64_t* __restrict__ slot_off, const float* __restrict__ grad, rec_grad_layout gl, Update up,
    int32_t* __restrict__ status, const int32_t* __restrict__ span_flag, int fp_mode) {
  sparse_small_body<NACC, Update>(blockIdx.x, n, D, S, N, pad, ids, slot_off, grad, gl, up, status, span_flag, fp_mode);
}

// ------------------------------------------------------------------ one-launch merge by ROW BUCKETS (n > kBucketMin)
// The merge above compares a lookup with every other lookup of its list: O(n^2 / 64) wave steps and, worse, every block
// staging the whole id list — 22 us + a 6 us span-check launch for the 13 312 lookups of a DeepFM batch of 512 on 26 slot
// tables (compared slot by slot), 98 us on the reference's own ONE shared table, where every lookup meets every other.
// Here block b owns the rows that HASH to bucket b: every block still reads all n ids once (8 n bytes from L2, no
// staging), but keeps only its own ~n / P lookups — (row, position) pairs in ascending position, built in two steps so
// that the list is dense and ordered without atomics: each wave counts the members of its contiguous range of positions
// (the ballots stay in registers), the counts meet in LDS, and the wave writes its members behind the waves in front of
// it.  All occurrences of a row are in ONE bucket whatever the slots are, so nothing depends on ids staying inside
// their slot's span (no span check), and the duplicate search runs over ~16 entries instead of 512 or 13 312.  The wave
// of a row's first occurrence adds the gradient rows of every occurrence in ascending position (the SelectedRows merge
// order: bit-identical to the kernels above) and applies the update; its record and moment lines are requested BEFORE
// the gradient rows, so the two fetches overlap.  A bucket with more than kBucketCap members (a row that owns a quarter
// of the batch) does not fit its LDS list: that block walks the positions in global memory instead — same arithmetic,
// same order, slower.
constexpr int kBucketMin = 2048;       // below: sparse_small_kernel (few blocks, cheap staging)
constexpr int kBucketWaves = 16;
constexpr int kBucketCap = 4096;       // (row, position) pairs of one bucket in LDS: 32 KB
constexpr int kBucketIts = (kSmallMergeMax + kBucketWaves * kWave - 1) / (kBucketWaves * kWave);   // positions per lane: 15
constexpr int kQuadCap = kBucketWaves * kWave;   // a list of at most one entry per thread ...
constexpr int kQuadLanes = 16;                   // ... of rows at most this wide: a row per 16 lanes (step 2Q below)

__device__ __forceinline__ unsigned bucket_of(int row, unsigned P) {
  const unsigned h = (unsigned)row * 0x9E3779B1u;
  return (unsigned)(((unsigned long long)(h ^ (h >> 15)) * P) >> 32);
}

template <int NACC, class Update>
__device__ __forceinline__ void sparse_bucket_body(
    const unsigned me, const unsigned P, int n, int D, int S, int64_t N, int64_t pad, const int64_t* __restrict__ ids,
    const int64_t* __restrict__ slot_off, const float* __restrict__ grad, rec_grad_layout gl, Update up,
    int32_t* __restrict__ status) {
  constexpr bool kRecord = std::is_same<Update, SmallAdamRecord>::value;
  constexpr int FLY = NACC == 1 ? kSmallList / 2 : kSmallList / NACC;   // (16 rows in flight at most: 32 spilled beside step 1's registers)
  __shared__ int b_rows[kBucketCap];
  __shared__ int b_pos[kBucketCap];
  __shared__ int b_wl[kBucketWaves][kSmallList];
  __shared__ int b_cnt[kBucketWaves + 1];
  __shared__ int b_next[kQuadCap];     // step 2Q: the next entry of the same row, -1 = none
  __shared__ int b_own[kQuadCap];      // step 2Q: the entries that are their row's first occurrence (any order)
  __shared__ int b_nown;
  const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
  if (threadIdx.x == 0) b_nown = 0;
  auto row_of = [&](int p) -> int {                  // table row of position p, -1 = padding / out of range
    const int64_t id = ids[p];
    const bool isp = pad >= 0 && id == pad;
    const int64_t r = slot_off ? id + slot_off[p % S] : id;
    return (!isp && r >= 0 && r < N) ? (int)r : -1;
  };
  // ---- step 1: this wave's range of positions, members counted (ballots and rows stay in registers)
  const int per_wave = ((n + kBucketWaves - 1) / kBucketWaves + kWave - 1) / kWave * kWave;
  const int p0 = wave * per_wave;
  unsigned long long mask[kBucketIts];
  int rowv[kBucketIts];
  int mine_cnt = 0, oob = 0;
#pragma unroll
  for (int it = 0; it < kBucketIts; ++it) {
    const int p = p0 + it * kWave + lane;
    const bool in = it * kWave < per_wave && p < n;
    int r = -1;
    if (in) {
      const int64_t id = ids[p];
      const bool isp = pad >= 0 && id == pad;
      const int64_t rr = slot_off ? id + slot_off[p % S] : id;
      const bool inr = rr >= 0 && rr < N;
      oob |= (!isp && !inr) ? 1 : 0;
      r = (!isp && inr) ? (int)rr : -1;
    }
    rowv[it] = r;
    mask[it] = __ballot(r >= 0 && bucket_of(r, P) == me);
    mine_cnt += (int)__popcll(mask[it]);
  }
  if (me == 0 && oob) atomicOr(status, REC_FLAG_INDEX_OOB);      // (every block sees every id: one of them reports)
  if (lane == 0) b_cnt[wave] = mine_cnt;
  __syncthreads();
  int my_off = 0, m = 0;
#pragma unroll
  for (int w = 0; w < kBucketWaves; ++w) {
    const int c = b_cnt[w];
    my_off += w < wave ? c : 0;
    m += c;
  }
  const bool in_lds = m <= kBucketCap;
  if (in_lds) {
    int at = my_off;
#pragma unroll
    for (int it = 0; it < kBucketIts; ++it) {
      const unsigned long long mk = mask[it];
      if ((mk >> lane) & 1ull) {
        const int idx = at + (int)__popcll(mk & ((1ull << lane) - 1ull));
        b_rows[idx] = rowv[it];
        b_pos[idx] = p0 + it * kWave + lane;
      }
      at += (int)__popcll(mk);
    }
    if (threadIdx.x < 4 && m + (int)threadIdx.x < kBucketCap) b_rows[m + threadIdx.x] = -2;   // (step 2Q reads whole int4s)
  }
  __syncthreads();
  // ---- step 2Q: narrow rows (D <= 16) and a list of at most one entry per thread — the DeepFM record at the reference's
  // batch size: ~64 entries per bucket.  Step 2 below gives a wave ONE entry at a time — search, record lines, gradient
  // rows, stores: four dependent round trips per entry, four entries per wave, nine of 64 lanes carrying a D 9 row.  Here
  // thread j looks at entry j: one pass over the list tells it whether an earlier entry holds its row (then it is not
  // the row's owner) and which entry is the next of the same row; the owners are then dealt to groups of 16 lanes, which
  // walk their row's chain — ascending position, the same order of additions as step 2 — with four gradient rows in
  // flight.  Every row of the bucket is in flight at once: one round of latencies per block instead of four per wave.
  if constexpr (NACC == 1 && kRecord) {
    if (in_lds && m <= kQuadCap && D <= kQuadLanes) {
      const int j = threadIdx.x;
      if (j < m) {
        const int my = b_rows[j];
        bool first = true;
        int nxt = -1;
        const int4* r4 = reinterpret_cast<const int4*>(b_rows);
        for (int k4 = 0; k4 * 4 < m; ++k4) {                    // (every lane reads the same address: a broadcast)
          const int4 v = r4[k4];
          const int k = k4 * 4;
          const bool e0 = v.x == my, e1 = v.y == my, e2 = v.z == my, e3 = v.w == my;
          first = first && !((e0 && k < j) || (e1 && k + 1 < j) || (e2 && k + 2 < j) || (e3 && k + 3 < j));
          if (nxt < 0) nxt = (e0 && k > j) ? k : (e1 && k + 1 > j) ? k + 1 : (e2 && k + 2 > j) ? k + 2 : (e3 && k + 3 > j) ? k + 3 : -1;
        }
        b_next[j] = nxt;
        if (first) b_own[atomicAdd(&b_nown, 1)] = j;
      }
      __syncthreads();
      const int nown = b_nown;
      const int l = threadIdx.x % kQuadLanes, grp = threadIdx.x / kQuadLanes;
      const bool on = l < D;
      const int dc = on ? l : 0;
      const bool plain = gl.group == 1 && gl.div == 1 && gl.index == nullptr;
      const float sc = up.grad_scale ? up.grad_scale[0] : 1.f;
      for (int o = grp; o < nown; o += kQuadCap / kQuadLanes) {
        int k = b_own[o];
        const int my = b_rows[k];
        float* r = up.rec + (int64_t)my * up.stride;
        float* mv = up.MV + (int64_t)my * up.sstride;
        float pw = r[dc], pm = mv[dc], pv = mv[up.v_off + dc];
        float q1 = 0.f, q2 = 0.f, q3 = 0.f;
        if (l == 0) { q1 = r[D]; q2 = r[D + 1]; q3 = r[D + 2]; }
        float acc = 0.f, acc1 = 0.f;
        while (k >= 0) {
          int q[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            q[u] = k >= 0 ? b_pos[k] : -1;
            k = k >= 0 ? b_next[k] : -1;
          }
          float x[4], x1[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int qq = q[u] >= 0 ? q[u] : q[0];
            x[u] = grad[(plain ? (int64_t)qq * gl.group_stride : grad_offset(gl, qq, D)) + dc];
            x1[u] = up.grad1[grad_offset(up.gl1, qq, 1)];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            acc += (q[u] >= 0 && on) ? x[u] : 0.f;
            acc1 += q[u] >= 0 ? x1[u] : 0.f;
          }
        }
        if (on) {
          const float g = up.grad_scale ? scale_grad(acc, sc) : acc;
          adam_elem(pw, pm, pv, g, up.lr_t, up.eps_t, up.b1, up.b2);
          r[l] = pw; mv[l] = pm; mv[up.v_off + l] = pv;
        }
        if (l == 0) {
          const float g = up.grad_scale ? scale_grad(acc1, sc) : acc1;
          adam_elem(q1, q2, q3, g, up.lr_t, up.eps_t, up.b1, up.b2);
          r[D] = q1; r[D + 1] = q2; r[D + 2] = q3;
        }
      }
      return;
    }
  }
  // ---- step 2: the entries of the compared list.  LDS list: entry j = (b_rows[j], b_pos[j]), j < m, every entry a member.
  //      Overflow: the list is the n positions themselves, a candidate is a member of this bucket.
  const int tn = in_lds ? m : n;
  auto row_at = [&](int j) { return in_lds ? b_rows[j] : row_of(j); };
  auto pos_at = [&](int j) { return in_lds ? b_pos[j] : j; };
  const bool plain = gl.group == 1 && gl.div == 1 && gl.index == nullptr;
  bool no_index = gl.index == nullptr;
  if constexpr (kRecord) no_index = no_index && up.gl1.index == nullptr;
  int* wl = b_wl[wave];
  for (int j = wave; j < tn; j += kBucketWaves) {
    const int my = row_at(j);                                    // wave-uniform
    if (my < 0 || (!in_lds && bucket_of(my, P) != me)) continue;
    // an earlier occurrence owns the row: searched from the nearest earlier entries backwards (only existence matters)
    bool owned = false;
    for (int c1 = j; c1 > 0 && !owned; c1 -= 4 * kWave) {
      const int c0 = c1 - 4 * kWave;
      bool hit = false;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = c0 + u * kWave + lane;
        hit |= k >= 0 && k < j && row_at(k >= 0 ? k : 0) == my;
      }
      owned = __ballot(hit) != 0;
    }
    if (owned) continue;
    // the row's lines first, its gradient rows behind them (two fetches in flight together)
    float pw[NACC], pm[NACC], pv[NACC], q1 = 0.f, q2 = 0.f, q3 = 0.f;
    if constexpr (kRecord) {
      const float* r = up.rec + (int64_t)my * up.stride;
      const float* mv = up.MV + (int64_t)my * up.sstride;
#pragma unroll
      for (int a = 0; a < NACC; ++a) {
        const int d = lane + a * kWave;
        const int dc = d < D ? d : 0;
        pw[a] = r[dc]; pm[a] = mv[dc]; pv[a] = mv[up.v_off + dc];
      }
      if (lane == 0) { q1 = r[D]; q2 = r[D + 1]; q3 = r[D + 2]; }
    } else {
      const float* r = up.P + (int64_t)my * up.stride;
#pragma unroll
      for (int a = 0; a < NACC; ++a) {
        const int d = lane + a * kWave;
        pw[a] = r[d < D ? d : 0];
        pm[a] = pv[a] = 0.f;
      }
    }
    float acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc[a] = 0.f;
    float acc1 = 0.f;
    int cnt = 0;
    // (the fetch tiers, flush and emit are those of sparse_small_body: rows are added in ascending position)
    auto fetch = [&](auto tier, auto noidx) {
      constexpr int TR = decltype(tier)::value;
      rec_grad_layout gq = gl;
      if constexpr (decltype(noidx)::value) gq.index = nullptr;
      float x[TR][NACC], x1[TR];
#pragma unroll
      for (int u = 0; u < TR; ++u) {
        const int q = wl[u < cnt ? u : 0];
        const float* g = grad + (plain ? (int64_t)q * gq.group_stride : grad_offset(gq, q, D));
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
          const int d = lane + a * kWave;
          x[u][a] = g[d < D ? d : 0];
        }
        if constexpr (kRecord) {
          rec_grad_layout g1 = up.gl1;
          if constexpr (decltype(noidx)::value) g1.index = nullptr;
          x1[u] = up.grad1[grad_offset(g1, q, 1)];
        } else {
          x1[u] = 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < TR; ++u) {
        const bool live = u < cnt;
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] += (live && lane + a * kWave < D) ? x[u][a] : 0.f;
        acc1 += live ? x1[u] : 0.f;
      }
    };
    auto flush = [&]() {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (no_index) {
        if (cnt <= 1) fetch(std::integral_constant<int, 1>{}, std::true_type{});
        else if (cnt <= 4) fetch(std::integral_constant<int, (FLY < 4 ? FLY : 4)>{}, std::true_type{});
        else if (cnt <= 16) fetch(std::integral_constant<int, (FLY < 16 ? FLY : 16)>{}, std::true_type{});
        else fetch(std::integral_constant<int, FLY>{}, std::true_type{});
      } else {
        fetch(std::integral_constant<int, FLY>{}, std::false_type{});
      }
      cnt = 0;
      __builtin_amdgcn_wave_barrier();
    };
    auto emit = [&](bool mine, int k) {
      unsigned long long mk = __ballot(mine);
      while (mk) {
        const int rank = __popcll(mk & ((1ull << lane) - 1ull));
        const int take = min((int)__popcll(mk), FLY - cnt);
        if (mine && rank < take) wl[cnt + rank] = pos_at(k);
        cnt += take;
        mine = mine && rank >= take;
        mk = __ballot(mine);
        if (cnt == FLY) flush();
      }
    };
    for (int c0 = (j / kWave) * kWave; c0 < tn; c0 += 4 * kWave) {
      bool mk[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = c0 + u * kWave + lane;
        mk[u] = k >= j && k < tn && row_at(k < tn ? k : 0) == my;
      }
      if (__ballot(mk[0] || mk[1] || mk[2] || mk[3]) == 0) continue;
#pragma unroll
      for (int u = 0; u < 4; ++u) emit(mk[u], c0 + u * kWave + lane);
    }
    if (cnt > 0) flush();
    if constexpr (!kRecord) {
      float* r = up.P + (int64_t)my * up.stride;
#pragma unroll
      for (int a = 0; a < NACC; ++a) {
        const int d = lane + a * kWave;
        if (d < D) r[d] = pw[a] - up.lr * acc[a];
      }
    } else {
      float* r = up.rec + (int64_t)my * up.stride;
      float* mv = up.MV + (int64_t)my * up.sstride;
      const float sc = up.grad_scale ? up.grad_scale[0] : 1.f;
#pragma unroll
      for (int a = 0; a < NACC; ++a) {
        const int d = lane + a * kWave;
        if (d < D) {
          float p_ = pw[a], m_ = pm[a], v_ = pv[a];
          const float g = up.grad_scale ? scale_grad(acc[a], sc) : acc[a];
          adam_elem(p_, m_, v_, g, up.lr_t, up.eps_t, up.b1, up.b2);
          r[d] = p_; mv[d] = m_; mv[up.v_off + d] = v_;
        }
      }
      if (lane == 0) {
        const float g = up.grad_scale ? scale_grad(acc1, sc) : acc1;
        adam_elem(q1, q2, q3, g, up.lr_t, up.eps_t, up.b1, up.b2);
        r[D] = q1; r[D + 1] = q2; r[D + 2] = q3;
      }
    }
    __builtin_amdgcn_wave_barrier();       // the wave's occurrence list is reused by its next entry
  }
}

template <int NACC, class Update>
__global__ __launch_bounds__(kBucketWaves* kWave) void sparse_bucket_kernel(
    int n, int D, int S, int64_t N, int64_t pad, const int64_t* __restrict__ ids, const int64_t* __restrict__ slot_off,
    const float* __restrict__ grad, rec_grad_layout gl, Update up, int32_t* __restrict__ status) {
  sparse_bucket_body<NACC, Update>(blockIdx.x, gridDim.x, n, D, S, N, pad, ids, slot_off, grad, gl, up, status);
}

// ------------------------------------------------------------------ the whole tail of a launch-bound step in ONE launch
// Blocks [0, buckets): the row-bucket merge + update above.  Behind them the roles of tail_roles.h: the fused head's
// partial rows folded (+ Adam on the last Linear), the FM backward's partial rows folded (+ the folded layer 0's share,
// Adam on dense_w / dense_w_one and on the dense rows of W_0), and two elementwise roles — the sparse rows of a folded W_0
// (gradient: dW_0' as it is) and every other dense parameter (gradient: the flat gradient buffer, the ranges owned by the
// roles above skipped).  Four launches of the step (ctr_head_fold, fold_partials, dense_fold_bwd_full, adam_dense) ride
// in the fifth: 16 -> 12 launches at the reference's batch size.
__global__ __launch_bounds__(kBucketWaves* kWave) void small_tail_kernel(
    unsigned buckets, int n, int D, int S, int64_t N, int64_t pad, const int64_t* __restrict__ ids,
    const int64_t* __restrict__ slot_off, const float* __restrict__ grad, rec_grad_layout gl, SmallAdamRecord up,
    int32_t* __restrict__ status, TailRoles t) {
  constexpr int NT = kBucketWaves * kWave;
  int b = (int)blockIdx.x;
  if ((unsigned)b < buckets) {
    sparse_bucket_body<1, SmallAdamRecord>((unsigned)b, buckets, n, D, S, N, pad, ids, slot_off, grad, gl, up, status);
    return;
  }
  b -= (int)buckets;
  const int tid = (int)threadIdx.x;
  if (b < t.head_blocks) {
    ctr_head_fold_role<true>(b, tid, t.head_nblk, t.head_n2, t.head_partial, t.head_invB, t.head_dw, t.head_db, t.loss,
                             t.adam, t.head_w_off, t.head_b_off);
    return;
  }
  b -= t.head_blocks;
  if (b < t.fm_blocks) {
    fm_fold_role<true>(b, tid, NT, t.fm_partial, t.fm_nblk, t.fm_split, t.ddw, t.ddw1, t.f, t.adam, t.dw_off, t.dw1_off,
                       t.w0_off);
    return;
  }
  b -= t.fm_blocks;
  if (b < t.w0_blocks) {                      // the sparse rows of a folded W_0: dW_0[e] = dW_0'[e], Adam
    const int64_t total = (int64_t)t.f.S * t.f.D * t.f.NO;
    for (int64_t e = (int64_t)b * NT + tid; e < total; e += (int64_t)t.w0_blocks * NT) {
      const float g = t.f.dW0f[e];
      t.f.dW0[e] = g;
      adam_dense_at(t.adam, t.w0_off + e, g);
    }
    return;
  }
  b -= t.w0_blocks;
  for (int64_t e = (int64_t)b * NT + tid; e < t.flat_numel; e += (int64_t)t.rest_blocks * NT) {
    bool owned = false;
#pragma unroll
    for (int r = 0; r < 5; ++r) owned = owned || (r < t.n_skip && e >= t.skip_lo[r] && e < t.skip_hi[r]);
    if (!owned) adam_dense_at(t.adam, e, t.flat_grad[e]);
  }
}

// LDS bytes of a one-launch merge over n lookups; *fp_mode = 1 when the fingerprint array fits beside the id list in the
// default 64 KB (plain tables only: the slot-local mode compares n / S lookups, a chunk or two)
static size_t small_lds_bytes(size_t n, bool plain, int* fp_mode) {
  static const bool on = [] { const char* v = getenv("REC_SMALL_FP"); return !(v && *v == '0'); }();
  const size_t base = (n + kSmallWaves * kSmallList) * sizeof(int);
  const size_t with_fp = (((n + 3) & ~(size_t)3) + kSmallWaves * kSmallList) * sizeof(int) + ((n + 511) & ~(size_t)511) * 2;
  *fp_mode = on && plain && n >= 1024 && with_fp <= 64 * 1024 ? 1 : 0;
  return *fp_mode ? with_fp : base;
}

// flag[0] = 1 when some id leaves its slot's span of rows (id < 0 or id >= slot_off[s + 1] - slot_off[s]): ONE block looks
// at every id once, instead of every block of the merge launch doing so (832 blocks x 13312 ids at the reference's batch
// size: 40 of the launch's 48 us).
__global__ __launch_bounds__(kSmallWaves* kWave) void small_span_check_kernel(int n, int S, const int64_t* __restrict__ ids,
                                                                             const int64_t* __restrict__ slot_off,
                                                                             int32_t* __restrict__ flag) {
  // eight ids in flight per thread: one block walking 13 312 ids one dependent load at a time took 9.5 us — on the critical
  // path of a 190 us step
  constexpr int NT = kSmallWaves * kWave, U = 8;
  int viol = 0;
  for (int i0 = threadIdx.x; i0 < n; i0 += U * NT) {
    int64_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * NT;
      v[u] = ids[i < n ? i : i0];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * NT;
      if (i < n) {
        const int sl = i % S;
        if (v[u] < 0 || (sl + 1 < S && v[u] >= slot_off[sl + 1] - slot_off[sl])) viol = 1;
      }
    }
  }
  viol = __syncthreads_or(viol);
  if (threadIdx.x == 0) flag[0] = viol ? 1 : 0;
}

// Several tables in ONE launch: DIN updates seven embedding tables per step (din/dygraph_model.py:64-73), independent of
// each other; as seven launches on one stream they run one after the other, two of them 50 us long (the target tables:
// one row per sample hit by all its ~150 history positions) — 150 of the 440 us of a batch-32 step.  A block finds its
// table from the block ranges; the tables then share the chip.
constexpr int kSmallJobsMax = 8;
struct SmallJob {
  int n, D, block0;
  int64_t N, pad;
  const int64_t* ids;
  const float* grad;
  rec_grad_layout gl;
  SmallSgd up;
};
struct SmallJobs {
  int count, fp_mode;
  SmallJob j[kSmallJobsMax];
  int merge_blocks;          // blocks of the merges; the blocks behind them: the dense parameters' SGD (tail_roles.h)
  int dense_blocks;
  int64_t dense_n;
  float* dense_p;
  const float* dense_g;
  float dense_lr;
};
template <int NACC>
__global__ __launch_bounds__(kSmallWaves* kWave) void sparse_small_multi_kernel(SmallJobs jobs, int32_t* __restrict__ status) {
  if ((int)blockIdx.x >= jobs.merge_blocks) {            // rec_sgd_dense's statement, in this launch
    const int64_t nt = (int64_t)jobs.dense_blocks * blockDim.x;
    for (int64_t i = (int64_t)((int)blockIdx.x - jobs.merge_blocks) * blockDim.x + threadIdx.x; i < jobs.dense_n; i += nt)
      jobs.dense_p[i] -= jobs.dense_lr * jobs.dense_g[i];
    return;
  }
  int k = 0;
#pragma unroll
  for (int i = 1; i < kSmallJobsMax; ++i)
    if (i < jobs.count && (int)blockIdx.x >= jobs.j[i].block0) k = i;
  const SmallJob& jb = jobs.j[k];
  sparse_small_body<NACC, SmallSgd>((int)blockIdx.x - jb.block0, jb.n, jb.D, 1, jb.N, jb.pad, jb.ids, nullptr, jb.grad, jb.gl,
                                    jb.up, status, nullptr, jobs.fp_mode);
}

// sum over the merged rows of |g_row|^2 (global-norm clipping needs the norm of the MERGED sparse grad)
template <int VEC, int LANES>
__global__ __launch_bounds__(kBlock) void sparse_rows_sumsq_kernel(
    int D, const int32_t* __restrict__ n_uniq, const int32_t* __restrict__ seg_off,
    const int32_t* __restrict__ spos, const float* __restrict__ grad, rec_grad_layout gl,
    float* __restrict__ partial) {
  __shared__ float red[kBlock];
  float acc = 0.f;
  const int d0 = (threadIdx.x % LANES) * VEC;
  const int U = n_uniq[0];
  for (int64_t u = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / LANES; u < U;
       u += (int64_t)gridDim.x * kBlock / LANES) {
    if (d0 < D) {
      float g[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) g[i] = 0.f;
      segment_sum<VEC>(g, seg_off[u], seg_off[u + 1], spos, grad, gl, D, d0);
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc += g[i] * g[i];
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(kBlock) void dense_sumsq_kernel(int64_t n, const float* __restrict__ x,
                                                             float* __restrict__ partial) {
  __shared__ float red[kBlock];
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
    acc += x[i] * x[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// out[0] = (accumulate ? out[0] : 0) + sum(partial[0..n)) in a fixed order
__global__ __launch_bounds__(kBlock) void sumsq_fold_kernel(const float* __restrict__ partial, int n,
                                                            int accumulate, float* __restrict__ out) {
  __shared__ float red[kBlock];
  float t = 0.f;
  for (int i = threadIdx.x; i < n; i += kBlock) t += partial[i];
  red[threadIdx.x] = t;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + red[0];
}

// paddle.nn.ClipGradByGlobalNorm [EXT]: every gradient is multiplied by clip / max(||g||, clip)
__global__ void clip_scale_kernel(const float* __restrict__ sumsq, float clip, float* __restrict__ scale) {
  const float nrm = sqrtf(sumsq[0]);
  scale[0] = clip / fmaxf(nrm, clip);
}

__global__ void adam_dense_kernel(int64_t n, float* __restrict__ p, float* __restrict__ m,
                                  float* __restrict__ v, const float* __restrict__ g,
                                  const float* __restrict__ grad_scale, float lr_t, float eps_t,
                                  float b1, float b2) {
  const float sc = grad_scale ? grad_scale[0] : 1.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * sc;
    float pi = p[i], mi = m[i], vi = v[i];
    adam_dense_elem(pi, mi, vi, gi, lr_t, eps_t, b1, b2);
    m[i] = mi;
    v[i] = vi;
    p[i] = pi;
  }
}

static void adam_scalars(const rec_adam_hyper* h, float* lr_t, float* eps_t) {
  const float b1p = powf(h->beta1, (float)h->step), b2p = powf(h->beta2, (float)h->step);
  *lr_t = h->lr * sqrtf(1.f - b2p) / (1.f - b1p);
  *eps_t = h->eps * sqrtf(1.f - b2p);
}

}  // namespace rec

using namespace rec;

static bool use_u32_keys(int64_t N) { return N < 0xFFFFFFFFll; }

extern "C" int rec_ids_group_workspace_bytes(int64_t n, int64_t num_rows, size_t* bytes) {
  REC_REQUIRE(bytes && n >= 0 && num_rows > 0, REC_EINVAL, "bad arguments");
  REC_REQUIRE(n < (1ll << 31) - 1, REC_ESHAPE, "n too large for int32 positions");
  if (n == 0) { *bytes = 256; return REC_OK; }
  if (use_u32_keys(num_rows)) {
    GroupPlan<uint32_t> p;
    if (int rc = plan_group<uint32_t>(n, num_rows, &p)) return rc;
    *bytes = p.total;
  } else {
    GroupPlan<uint64_t> p;
    if (int rc = plan_group<uint64_t>(n, num_rows, &p)) return rc;
    *bytes = p.total;
  }
  return REC_OK;
}

extern "C" int rec_ids_group_payload(int64_t n, int32_t num_slots, int64_t num_rows, int64_t padding_idx,
                                     const int64_t* ids, const int64_t* slot_offset, const int32_t* payload,
                                     int32_t* sorted_pos, int64_t* uniq_rows, int32_t* seg_offset,
                                     int32_t* n_uniq, int32_t* status, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  REC_REQUIRE(n >= 0 && num_slots > 0 && num_rows > 0, REC_EINVAL, "bad sizes");
  REC_REQUIRE(n < (1ll << 31) - 1, REC_ESHAPE, "n too large for int32 positions");
  REC_REQUIRE(sorted_pos && uniq_rows && seg_offset && n_uniq && status, REC_EINVAL,
              "null pointer argument");
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    (void)hipMemsetAsync(n_uniq, 0, 4 * sizeof(int32_t), st);
    (void)hipMemsetAsync(seg_offset, 0, sizeof(int32_t), st);
    return REC_OK;
  }
  REC_REQUIRE(ids, REC_EINVAL, "ids is NULL");
  if (use_u32_keys(num_rows))
    return run_group<uint32_t>(n, num_slots, num_rows, padding_idx, ids, slot_offset, payload, sorted_pos,
                               uniq_rows, seg_offset, n_uniq, status, workspace, workspace_bytes,
                               st);
  return run_group<uint64_t>(n, num_slots, num_rows, padding_idx, ids, slot_offset, payload, sorted_pos,
                             uniq_rows, seg_offset, n_uniq, status, workspace, workspace_bytes, st);
}

extern "C" int rec_ids_group(int64_t n, int32_t num_slots, int64_t num_rows, int64_t padding_idx,
                             const int64_t* ids, const int64_t* slot_offset, int32_t* sorted_pos,
                             int64_t* uniq_rows, int32_t* seg_offset, int32_t* n_uniq,
                             int32_t* status, void* workspace, size_t workspace_bytes,
                             void* stream) {
  return rec_ids_group_payload(n, num_slots, num_rows, padding_idx, ids, slot_offset, nullptr, sorted_pos,
                               uniq_rows, seg_offset, n_uniq, status, workspace, workspace_bytes, stream);
}

namespace rec {
__global__ void rank_fill_kernel(int64_t n, int32_t* __restrict__ rank) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) rank[i] = -1;
}
__global__ void rank_scatter_kernel(const int32_t* __restrict__ n_uniq, const int32_t* __restrict__ spos,
                                    int32_t* __restrict__ rank) {
  const int nv = n_uniq[1];
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nv; k += (int64_t)gridDim.x * blockDim.x)
    rank[spos[k]] = (int32_t)k;
}
}  // namespace rec

extern "C" int rec_ids_rank(int64_t n, const int32_t* n_uniq, const int32_t* sorted_pos, int32_t* rank, void* stream) {
  REC_REQUIRE(n >= 0 && n < (1ll << 31) - 1, REC_EINVAL, "bad n");
  if (n == 0) return REC_OK;
  REC_REQUIRE(n_uniq && sorted_pos && rank, REC_EINVAL, "null pointer argument");
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)((n + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(rank_fill_kernel, dim3(grid), dim3(kBlock), 0, st, n, rank);
  hipLaunchKernelGGL(rank_scatter_kernel, dim3(grid < 2048u ? grid : 2048u), dim3(kBlock), 0, st, n_uniq, sorted_pos, rank);
  return check_launch("rec_ids_rank");
}

extern "C" int rec_ids_group_slots_workspace_bytes(int64_t batch, int32_t num_slots, int64_t slot_rows, size_t* bytes) {
  REC_REQUIRE(bytes && batch >= 0 && num_slots > 0 && slot_rows > 0, REC_EINVAL, "bad arguments");
  REC_REQUIRE(batch * num_slots < (1ll << 31) - 1, REC_ESHAPE, "batch x slots too large for int32 positions");
  if (batch == 0) { *bytes = 256; return REC_OK; }
  if (sg::eligible(batch, num_slots, slot_rows)) { *bytes = sg::workspace_bytes(batch, num_slots, slot_rows); return REC_OK; }
  return rec_ids_group_workspace_bytes(batch * num_slots, slot_rows * num_slots, bytes);
}

extern "C" int rec_ids_group_slots(int64_t batch, int32_t num_slots, int64_t slot_rows, int64_t padding_idx,
                                   const int64_t* ids, int32_t* sorted_pos, int64_t* uniq_rows, int32_t* seg_offset,
                                   int32_t* n_uniq, int32_t* rank, int32_t* status, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  REC_REQUIRE(batch >= 0 && num_slots > 0 && slot_rows > 0, REC_EINVAL, "bad sizes");
  const int64_t n = batch * num_slots;
  REC_REQUIRE(n < (1ll << 31) - 1, REC_ESHAPE, "batch x slots too large for int32 positions");
  REC_REQUIRE(sorted_pos && uniq_rows && seg_offset && n_uniq && status, REC_EINVAL, "null pointer argument");
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    (void)hipMemsetAsync(n_uniq, 0, 4 * sizeof(int32_t), st);
    (void)hipMemsetAsync(seg_offset, 0, sizeof(int32_t), st);
    return REC_OK;
  }
  REC_REQUIRE(ids, REC_EINVAL, "ids is NULL");
  if (sg::eligible(batch, num_slots, slot_rows))
    return sg::run(batch, num_slots, slot_rows, padding_idx, ids, sorted_pos, uniq_rows, seg_offset, n_uniq, rank, status,
                   workspace, workspace_bytes, st);
  const int64_t N = slot_rows * num_slots;
  int rc;
  if (use_u32_keys(N))
    rc = run_group<uint32_t>(n, num_slots, N, padding_idx, ids, nullptr, nullptr, sorted_pos, uniq_rows, seg_offset,
                             n_uniq, status, workspace, workspace_bytes, st, slot_rows);
  else
    rc = run_group<uint64_t>(n, num_slots, N, padding_idx, ids, nullptr, nullptr, sorted_pos, uniq_rows, seg_offset,
                             n_uniq, status, workspace, workspace_bytes, st, slot_rows);
  if (rc != REC_OK || !rank) return rc;
  return rec_ids_rank(n, n_uniq, sorted_pos, rank, stream);
}

extern "C" int rec_segment_partials_bytes(int64_t n_max, int32_t emb_dim, size_t* bytes) {
  REC_REQUIRE(n_max >= 0 && emb_dim > 0 && bytes, REC_EINVAL, "bad arguments");
  *bytes = (size_t)((n_max + kSegTile - 1) / kSegTile) * 2 * (size_t)emb_dim * sizeof(float);
  return REC_OK;
}

extern "C" int rec_segment_partials(int64_t n_max, int32_t emb_dim, const int32_t* n_uniq,
                                    const int32_t* seg_offset, const int32_t* sorted_pos,
                                    const float* grad, const rec_grad_layout* grad_layout,
                                    float* partials, void* stream) {
  rec_grad_layout gl = {1, 0, 0, nullptr, nullptr};
  if (grad_layout) gl = *grad_layout;
  gl.partials = nullptr;
  REC_REQUIRE(n_max >= 0 && emb_dim > 0 && gl.div >= 1, REC_EINVAL, "bad sizes");
  REC_REQUIRE(gl.group <= 0 || gl.group_stride >= (int64_t)gl.group * emb_dim, REC_EINVAL,
              "grad group_stride too small");
  REC_REQUIRE(((uintptr_t)gl.partials) % 16 == 0, REC_EINVAL, "grad_layout.partials must be 16-byte aligned");
  REC_REQUIRE(n_uniq && seg_offset && sorted_pos && grad && partials, REC_EINVAL,
              "null pointer argument");
  if (n_max == 0) return REC_OK;
  REC_REQUIRE(((uintptr_t)partials) % 16 == 0, REC_EINVAL, "partials must be 16-byte aligned");
  const bool gvec = ((uintptr_t)grad) % 16 == 0 && (gl.group <= 0 || gl.group_stride % 4 == 0);
  return dispatch_row_shape(emb_dim, gvec ? 4 : 1, [&](auto vec, auto lanes) -> int {
    constexpr int VEC = decltype(vec)::value, LANES = decltype(lanes)::value;
    const int64_t tiles = (n_max + kSegTile - 1) / kSegTile;
    const int64_t grid = (tiles + kBlock / 64 - 1) / (kBlock / 64);
    REC_REQUIRE(grid < (1ll << 31), REC_ESHAPE, "too many positions");
    hipLaunchKernelGGL((segment_partials_kernel<VEC, LANES>), dim3((unsigned)grid), dim3(kBlock), 0,
                       (hipStream_t)stream, emb_dim, n_uniq, seg_offset, sorted_pos, grad, gl, partials);
    return check_launch("rec_segment_partials");
  });
}

extern "C" int rec_sparse_adam_rows(int64_t n_max, int32_t emb_dim, int32_t row_stride,
                                    int32_t state_stride, const int32_t* n_uniq, const int64_t* uniq_rows,
                                    const int32_t* seg_offset, const int32_t* sorted_pos,
                                    const float* grad, const rec_grad_layout* grad_layout,
                                    const float* grad_scale, float* P, float* M, float* V,
                                    const rec_adam_hyper* hyper, void* stream) {
  rec_grad_layout gl = {1, 0, 0, nullptr, nullptr};
  if (grad_layout) gl = *grad_layout;
  REC_REQUIRE(n_max >= 0 && emb_dim > 0 && row_stride >= emb_dim && gl.div >= 1, REC_EINVAL,
              "bad sizes");
  REC_REQUIRE(gl.group <= 0 || gl.group_stride >= (int64_t)gl.group * emb_dim, REC_EINVAL,
              "grad group_stride too small");
  REC_REQUIRE(((uintptr_t)gl.partials) % 16 == 0, REC_EINVAL, "grad_layout.partials must be 16-byte aligned");
  if (state_stride <= 0) state_stride = row_stride;
  REC_REQUIRE(state_stride >= emb_dim, REC_EINVAL, "state_stride < emb_dim");
  REC_REQUIRE(n_uniq && uniq_rows && seg_offset && sorted_pos && grad && P && M && V && hyper,
              REC_EINVAL, "null pointer argument");
  REC_REQUIRE(hyper->step >= 1, REC_EINVAL, "Adam step must be >= 1");
  if (n_max == 0) return REC_OK;
  float lr_t, eps_t;
  adam_scalars(hyper, &lr_t, &eps_t);
  hipStream_t st = (hipStream_t)stream;
  // float4 gradient loads need 16-B aligned gradient rows
  const bool gvec = ((uintptr_t)grad) % 16 == 0 && (gl.group <= 0 || gl.group_stride % 4 == 0);
  const bool rows4 = emb_dim % 4 == 0 && row_stride % 4 == 0 && gvec && state_stride % 4 == 0;
  static const bool narrow_ok = [] { const char* v = getenv("REC_NARROW_ROWS"); return !(v && *v == '0'); }();
  if (!rows4 && emb_dim <= 16 && narrow_ok) {   // no float4 row groups: one lane per row (see the narrow kernel)
    const bool v4 = row_stride % 4 == 0 && state_stride % 4 == 0 && ((uintptr_t)P) % 16 == 0 &&
                    ((uintptr_t)M) % 16 == 0 && ((uintptr_t)V) % 16 == 0;
    int64_t grid = (n_max + kBlock - 1) / kBlock;
    if (grid > (int64_t)kNumCU * 32) grid = (int64_t)kNumCU * 32;   // grid-stride loop in the kernel
    const int nv = (emb_dim + 3) / 4;
#define REC_NARROW(NV_, V4_)                                                                                 \
  hipLaunchKernelGGL((sparse_adam_rows_narrow_kernel<NV_, V4_>), dim3((unsigned)grid), dim3(kBlock), 0, st,   \
                     emb_dim, row_stride, state_stride, n_uniq, uniq_rows, seg_offset, sorted_pos, grad, gl,   \
                     grad_scale, P, M, V, lr_t, eps_t, hyper->beta1, hyper->beta2)
    if (v4) {
      if (nv == 1) REC_NARROW(1, true); else if (nv == 2) REC_NARROW(2, true);
      else if (nv == 3) REC_NARROW(3, true); else REC_NARROW(4, true);
    } else {
      if (nv == 1) REC_NARROW(1, false); else if (nv == 2) REC_NARROW(2, false);
      else if (nv == 3) REC_NARROW(3, false); else REC_NARROW(4, false);
    }
#undef REC_NARROW
    return check_launch("rec_sparse_adam_rows (narrow)");
  }
  return dispatch_row_shape(emb_dim, (gvec && state_stride % 4 == 0) ? row_stride : row_stride | 1,
                            [&](auto vec, auto lanes) -> int {
    constexpr int VEC = decltype(vec)::value, LANES = decltype(lanes)::value;
    const int64_t grid = (n_max * LANES + kBlock - 1) / kBlock;
    REC_REQUIRE(grid < (1ll << 31), REC_ESHAPE, "too many rows");
    hipLaunchKernelGGL((sparse_adam_rows_kernel<VEC, LANES>), dim3((unsigned)grid), dim3(kBlock),
                       0, st, emb_dim, row_stride, state_stride, n_uniq, uniq_rows, seg_offset, sorted_pos, grad,
                       gl, grad_scale, P, M, V, lr_t, eps_t, hyper->beta1, hyper->beta2);
    return check_launch("rec_sparse_adam_rows");
  });
}

extern "C" int rec_sparse_adam_record(int64_t n_max, int32_t emb_dim, int32_t rec_stride,
                                      int32_t state_stride, int32_t v_offset, const int32_t* n_uniq,
                                      const int64_t* uniq_rows, const int32_t* seg_offset,
                                      const int32_t* sorted_pos, const float* grad,
                                      const rec_grad_layout* grad_layout, const float* grad1,
                                      const rec_grad_layout* grad1_layout, const float* grad_scale,
                                      float* rec, float* MV, const rec_adam_hyper* hyper, void* stream) {
  rec_grad_layout gl = {1, 0, 0, nullptr, nullptr}, gl1 = {1, 0, 0, nullptr, nullptr};
  if (grad_layout) gl = *grad_layout;
  if (grad1_layout) gl1 = *grad1_layout;
  REC_REQUIRE(n_max >= 0 && emb_dim > 0 && rec_stride >= emb_dim + 3 && gl.div >= 1 && gl1.div >= 1,
              REC_EINVAL, "bad sizes (the record holds W(D) | W1 | m1 | v1)");
  REC_REQUIRE(v_offset >= emb_dim && state_stride >= v_offset + emb_dim, REC_EINVAL,
              "state line must hold m(D) at 0 and v(D) at v_offset");
  REC_REQUIRE(gl.group <= 0 || gl.group_stride >= (int64_t)gl.group * emb_dim, REC_EINVAL,
              "grad group_stride too small");
  REC_REQUIRE(((uintptr_t)gl.partials) % 16 == 0 && ((uintptr_t)gl1.partials) % 4 == 0, REC_EINVAL,
              "grad_layout.partials must be 16-byte aligned");
  REC_REQUIRE(n_uniq && uniq_rows && seg_offset && sorted_pos && grad && grad1 && rec && MV && hyper,
              REC_EINVAL, "null pointer argument");
  REC_REQUIRE(hyper->step >= 1, REC_EINVAL, "Adam step must be >= 1");
  if (n_max == 0) return REC_OK;
  float lr_t, eps_t;
  adam_scalars(hyper, &lr_t, &eps_t);
  const bool vec = ((uintptr_t)grad) % 16 == 0 && (gl.group <= 0 || gl.group_stride % 4 == 0) &&
                   state_stride % 4 == 0 && v_offset % 4 == 0 && ((uintptr_t)rec) % 16 == 0 &&
                   ((uintptr_t)MV) % 16 == 0;
  static const int sparse_nt = [] { const char* v = getenv("REC_SPARSE_NT"); return (v && *v) ? atoi(v) : 1; }();
  return dispatch_row_shape(emb_dim, vec ? rec_stride : rec_stride | 1, [&](auto vec_, auto lanes) -> int {
    constexpr int VEC = decltype(vec_)::value, LANES = decltype(lanes)::value;
    const int64_t grid = (n_max * LANES + kBlock - 1) / kBlock;
    REC_REQUIRE(grid < (1ll << 31), REC_ESHAPE, "too many rows");
    hipLaunchKernelGGL((sparse_adam_record_kernel<VEC, LANES>), dim3((unsigned)grid), dim3(kBlock), 0,
                       (hipStream_t)stream, emb_dim, rec_stride, state_stride, v_offset, n_uniq, uniq_rows,
                       seg_offset, sorted_pos, grad, gl, grad1, gl1, grad_scale, rec, MV, lr_t, eps_t,
                       hyper->beta1, hyper->beta2, sparse_nt);
    return check_launch("rec_sparse_adam_record");
  });
}

extern "C" int rec_adam_record_all(int64_t num_rows, int32_t emb_dim, int32_t rec_stride, int32_t state_stride,
                                   int32_t v_offset, const int32_t* n_uniq, const int64_t* uniq_rows,
                                   const int32_t* seg_offset, const int32_t* sorted_pos, const float* grad,
                                   const rec_grad_layout* grad_layout, const float* grad1,
                                   const rec_grad_layout* grad1_layout, const float* grad_scale, float* rec, float* MV,
                                   const rec_adam_hyper* hyper, void* stream) {
  rec_grad_layout gl = {1, 0, 0, nullptr, nullptr}, gl1 = {1, 0, 0, nullptr, nullptr};
  if (grad_layout) gl = *grad_layout;
  if (grad1_layout) gl1 = *grad1_layout;
  REC_REQUIRE(num_rows >= 0 && emb_dim > 0 && rec_stride >= emb_dim + 3 && gl.div >= 1 && gl1.div >= 1, REC_EINVAL,
              "bad sizes (the record holds W(D) | W1 | m1 | v1)");
  REC_REQUIRE(v_offset >= emb_dim && state_stride >= v_offset + emb_dim, REC_EINVAL,
              "state line must hold m(D) at 0 and v(D) at v_offset");
  REC_REQUIRE(gl.group <= 0 || gl.group_stride >= (int64_t)gl.group * emb_dim, REC_EINVAL, "grad group_stride too small");
  REC_REQUIRE(((uintptr_t)gl.partials) % 16 == 0 && ((uintptr_t)gl1.partials) % 4 == 0, REC_EINVAL,
              "grad_layout.partials must be 16-byte aligned");
  REC_REQUIRE(n_uniq && uniq_rows && seg_offset && sorted_pos && grad && grad1 && rec && MV && hyper, REC_EINVAL,
              "null pointer argument");
  REC_REQUIRE(hyper->step >= 1, REC_EINVAL, "Adam step must be >= 1");
  if (num_rows == 0) return REC_OK;
  float lr_t, eps_t;
  adam_scalars(hyper, &lr_t, &eps_t);
  const bool vec = ((uintptr_t)grad) % 16 == 0 && (gl.group <= 0 || gl.group_stride % 4 == 0) && state_stride % 4 == 0 &&
                   v_offset % 4 == 0 && ((uintptr_t)rec) % 16 == 0 && ((uintptr_t)MV) % 16 == 0;
  return dispatch_row_shape(emb_dim, vec ? rec_stride : rec_stride | 1, [&](auto vec_, auto lanes) -> int {
    constexpr int VEC = decltype(vec_)::value, LANES = decltype(lanes)::value;
    constexpr int RB = kBlock / LANES;
    const int64_t grid = (num_rows + RB - 1) / RB;
    REC_REQUIRE(grid < (1ll << 31), REC_ESHAPE, "too many rows");
    hipLaunchKernelGGL((adam_record_all_kernel<VEC, LANES>), dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream,
                       num_rows, emb_dim, rec_stride, state_stride, v_offset, n_uniq, uniq_rows, seg_offset, sorted_pos,
                       grad, gl, grad1, gl1, grad_scale, rec, MV, lr_t, eps_t, hyper->beta1, hyper->beta2);
    return check_launch("rec_adam_record_all");
  });
}

// ---------------------------------------------------------------------------------------------
// PS / gpubox accessor rule (slot_dnn/config_online.yaml:57-79): a feature value is
//   [show, click, embed_w, embedx(D-1)]  (dnn/net.py:71-79: sparse_embedding(size=[N, D+2]) + CVM strips 2)
// with ONE AdaGrad scalar g2sum per part (embed_w / embedx) — D+4 floats per row instead of Adam's 3D:
//   scale = sqrt(initial_g2sum / (initial_g2sum + g2sum));  w -= lr * g * scale, clipped to the bounds;
//   g2sum += mean(g^2) over the part;  show += occurrences, click += sum of their labels.
// Record (one 128-B line for D <= 28): [show | click | g2sum_w | g2sum_x | W(D) | pad] — W starts 16-B
// aligned so the lookup kernels keep their float4 path.  One thread per
// touched row; the duplicate gradients of the row are summed in ascending-position order first.
__global__ __launch_bounds__(kBlock) void sparse_adagrad_rows_kernel(
    int D, int stride, int S, const int32_t* __restrict__ n_uniq, const int64_t* __restrict__ uniq,
    const int32_t* __restrict__ seg_off, const int32_t* __restrict__ spos,
    const float* __restrict__ grad, rec_grad_layout gl, const int64_t* __restrict__ label,
    float* __restrict__ rec, rec_adagrad_hyper h) {
  const int64_t u = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (u >= n_uniq[0]) return;
  float* r = rec + uniq[u] * stride;
  const int beg = seg_off[u], end = seg_off[u + 1];
  float clicks = 0.f;
  auto smp = [&](int pos) { return (gl.index ? gl.index[pos] : pos) / S; };   // sample of a lookup position
  if (label) {
    int k = beg;
    for (; k + 4 <= end; k += 4) {   // four independent loads in flight
      const int64_t l0 = label[smp(spos[k])], l1 = label[smp(spos[k + 1])], l2 = label[smp(spos[k + 2])],
                    l3 = label[smp(spos[k + 3])];
      clicks += (float)(l0 + l1 + l2 + l3);
    }
    for (; k < end; ++k) clicks += (float)label[smp(spos[k])];
  }
  r[0] += (float)(end - beg);   // show: every lookup of the row is one impression (dnn/static_model.py:86-94)
  r[1] += clicks;
  const float g2w = r[2], g2x = r[3];
  const float sw = sqrtf(h.initial_g2sum / (h.initial_g2sum + g2w));
  const float sx = sqrtf(h.initial_g2sum / (h.initial_g2sum + g2x));
  float addw = 0.f, addx = 0.f;
  for (int d = 0; d < D; ++d) {
    float gv[1] = {0.f};
    segment_sum<1>(gv, beg, end, spos, grad, gl, D, d);
    const float g = gv[0];
    float w = r[4 + d] - h.lr * g * (d == 0 ? sw : sx);
    w = fminf(fmaxf(w, h.min_bound), h.max_bound);
    r[4 + d] = w;
    if (d == 0) addw = g * g; else addx += g * g;
  }
  r[2] = g2w + addw;
  if (D > 1) r[3] = g2x + addx / (float)(D - 1);
}

constexpr int kSumsqBlocks = 1024;

extern "C" int rec_sparse_adagrad_rows(int64_t n_max, int32_t emb_dim, int32_t row_stride,
                                       int32_t num_slots, const int32_t* n_uniq,
                                       const int64_t* uniq_rows, const int32_t* seg_offset,
                                       const int32_t* sorted_pos, const float* grad,
                                       const rec_grad_layout* grad_layout, const int64_t* label,
                                       float* rec, const rec_adagrad_hyper* hyper, void* stream) {
  rec_grad_layout gl = {1, 0, 0, nullptr, nullptr};
  if (grad_layout) gl = *grad_layout;
  REC_REQUIRE(n_max >= 0 && emb_dim > 0 && row_stride >= emb_dim + 4 && num_slots > 0 && gl.div >= 1,
              REC_EINVAL, "bad sizes (row_stride must hold show, click, D weights and 2 g2sum)");
  REC_REQUIRE(n_uniq && uniq_rows && seg_offset && sorted_pos && grad && rec && hyper, REC_EINVAL,
              "null pointer argument");
  REC_REQUIRE(hyper->initial_g2sum > 0.f && hyper->min_bound <= hyper->max_bound, REC_EINVAL,
              "bad hyper-parameters");
  if (n_max == 0) return REC_OK;
  const int64_t grid = (n_max + kBlock - 1) / kBlock;
  REC_REQUIRE(grid < (1ll << 31), REC_ESHAPE, "too many rows");
  hipLaunchKernelGGL(sparse_adagrad_rows_kernel, dim3((unsigned)grid), dim3(kBlock), 0,
                     (hipStream_t)stream, emb_dim, row_stride, num_slots, n_uniq, uniq_rows, seg_offset,
                     sorted_pos, grad, gl, label, rec, *hyper);
  return check_launch("rec_sparse_adagrad_rows");
}

extern "C" int rec_sumsq_workspace_bytes(size_t* bytes) {
  REC_REQUIRE(bytes, REC_EINVAL, "bytes is NULL");
  *bytes = kSumsqBlocks * sizeof(float);
  return REC_OK;
}

extern "C" int rec_sumsq(int64_t n, const float* x, float* out, int32_t accumulate, void* workspace,
                         size_t workspace_bytes, void* stream) {
  REC_REQUIRE(n >= 0 && out && (n == 0 || x), REC_EINVAL, "bad arguments");
  REC_REQUIRE(workspace && workspace_bytes >= kSumsqBlocks * sizeof(float), REC_EWORKSPACE,
              "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  int64_t grid = (n + kBlock * 8 - 1) / (kBlock * 8);
  if (grid > kSumsqBlocks) grid = kSumsqBlocks;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(dense_sumsq_kernel, dim3((unsigned)grid), dim3(kBlock), 0, st, n, x,
                     (float*)workspace);
  hipLaunchKernelGGL(sumsq_fold_kernel, dim3(1), dim3(kBlock), 0, st, (const float*)workspace,
                     (int)grid, accumulate, out);
  return check_launch("rec_sumsq");
}

extern "C" int rec_sparse_rows_sumsq(int64_t n_max, int32_t emb_dim, const int32_t* n_uniq,
                                     const int32_t* seg_offset, const int32_t* sorted_pos,
                                     const float* grad, const rec_grad_layout* grad_layout,
                                     float* out, int32_t accumulate, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  rec_grad_layout gl = {1, 0, 0, nullptr, nullptr};
  if (grad_layout) gl = *grad_layout;
  REC_REQUIRE(n_max >= 0 && emb_dim > 0 && gl.div >= 1 && out, REC_EINVAL, "bad arguments");
  REC_REQUIRE(n_uniq && seg_offset && sorted_pos && grad, REC_EINVAL, "null pointer argument");
  REC_REQUIRE(workspace && workspace_bytes >= kSumsqBlocks * sizeof(float), REC_EWORKSPACE,
              "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const bool gvec = ((uintptr_t)grad) % 16 == 0 && (gl.group <= 0 || gl.group_stride % 4 == 0);
  return dispatch_row_shape(emb_dim, gvec ? emb_dim : emb_dim | 1, [&](auto vec, auto lanes) -> int {
    constexpr int VEC = decltype(vec)::value, LANES = decltype(lanes)::value;
    int64_t grid = (n_max * LANES + kBlock - 1) / kBlock;
    if (grid > kSumsqBlocks) grid = kSumsqBlocks;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL((sparse_rows_sumsq_kernel<VEC, LANES>), dim3((unsigned)grid), dim3(kBlock),
                       0, st, emb_dim, n_uniq, seg_offset, sorted_pos, grad, gl, (float*)workspace);
    hipLaunchKernelGGL(sumsq_fold_kernel, dim3(1), dim3(kBlock), 0, st, (const float*)workspace,
                       (int)grid, accumulate, out);
    return check_launch("rec_sparse_rows_sumsq");
  });
}

extern "C" int rec_clip_scale(const float* sumsq, float clip_norm, float* scale, void* stream) {
  REC_REQUIRE(sumsq && scale && clip_norm > 0.f, REC_EINVAL, "bad arguments");
  hipLaunchKernelGGL(clip_scale_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sumsq, clip_norm,
                     scale);
  return check_launch("rec_clip_scale");
}

extern "C" int rec_adam_dense(int64_t n, float* p, float* m, float* v, const float* g,
                              const float* grad_scale, const rec_adam_hyper* hyper, void* stream) {
  REC_REQUIRE(n >= 0 && p && m && v && g && hyper, REC_EINVAL, "bad arguments");
  REC_REQUIRE(hyper->step >= 1, REC_EINVAL, "Adam step must be >= 1");
  if (n == 0) return REC_OK;
  float lr_t, eps_t;
  adam_scalars(hyper, &lr_t, &eps_t);
  int64_t grid = (n + kBlock - 1) / kBlock;
  if (grid > kNumCU * 8) grid = kNumCU * 8;
  hipLaunchKernelGGL(adam_dense_kernel, dim3((unsigned)grid), dim3(kBlock), 0,
                     (hipStream_t)stream, n, p, m, v, g, grad_scale, lr_t, eps_t, hyper->beta1,
                     hyper->beta2);
  return check_launch("rec_adam_dense");
}

extern "C" int rec_sparse_sgd_rows(int64_t n_max, int32_t emb_dim, int32_t row_stride,
                                   const int32_t* n_uniq, const int64_t* uniq_rows,
                                   const int32_t* seg_offset, const int32_t* sorted_pos,
                                   const float* grad, const rec_grad_layout* grad_layout, float* P,
                                   float lr, void* stream) {
  rec_grad_layout gl = {1, 0, 0, nullptr, nullptr};
  if (grad_layout) gl = *grad_layout;
  REC_REQUIRE(n_max >= 0 && emb_dim > 0 && row_stride >= emb_dim && gl.div >= 1, REC_EINVAL, "bad sizes");
  REC_REQUIRE(gl.group <= 0 || gl.group_stride >= (int64_t)gl.group * emb_dim, REC_EINVAL,
              "grad group_stride too small");
  REC_REQUIRE(((uintptr_t)gl.partials) % 16 == 0, REC_EINVAL, "grad_layout.partials must be 16-byte aligned");
  REC_REQUIRE(n_uniq && uniq_rows && seg_offset && sorted_pos && grad && P, REC_EINVAL,
              "null pointer argument");
  if (n_max == 0) return REC_OK;
  const bool gvec = ((uintptr_t)grad) % 16 == 0 && (gl.group <= 0 || gl.group_stride % 4 == 0) &&
                    ((uintptr_t)P) % 16 == 0;
  return dispatch_row_shape(emb_dim, gvec ? row_stride : row_stride | 1, [&](auto vec, auto lanes) -> int {
    constexpr int VEC = decltype(vec)::value, LANES = decltype(lanes)::value;
    const int64_t grid = (n_max * LANES + kBlock - 1) / kBlock;
    REC_REQUIRE(grid < (1ll << 31), REC_ESHAPE, "too many rows");
    hipLaunchKernelGGL((sparse_sgd_rows_kernel<VEC, LANES>), dim3((unsigned)grid), dim3(kBlock), 0,
                       (hipStream_t)stream, emb_dim, row_stride, n_uniq, uniq_rows, seg_offset, sorted_pos,
                       grad, gl, P, lr);
    return check_launch("rec_sparse_sgd_rows");
  });
}

extern "C" int rec_sparse_sgd_small(int64_t n, int32_t emb_dim, int32_t row_stride, int64_t num_rows,
                                    int64_t padding_idx, const int64_t* ids, const float* grad,
                                    const rec_grad_layout* grad_layout, float* P, float lr, int32_t* status,
                                    void* stream) {
  rec_grad_layout gl = {1, 0, 0, nullptr, nullptr};
  if (grad_layout) gl = *grad_layout;
  gl.partials = nullptr;
  REC_REQUIRE(n >= 0 && emb_dim > 0 && row_stride >= emb_dim && num_rows > 0 && gl.div >= 1, REC_EINVAL, "bad sizes");
  REC_REQUIRE(n <= kSmallMergeMax, REC_ESHAPE, "n %lld > %d: use rec_ids_group + rec_sparse_sgd_rows", (long long)n,
              kSmallMergeMax);
  REC_REQUIRE(emb_dim <= 4 * kWave, REC_ESHAPE, "emb_dim %d > %d unsupported by the one-launch merge", emb_dim,
              4 * kWave);
  REC_REQUIRE(gl.group <= 0 || gl.group_stride >= (int64_t)gl.group * emb_dim, REC_EINVAL,
              "grad group_stride too small");
  if (n == 0) return REC_OK;
  REC_REQUIRE(ids && grad && P && status, REC_EINVAL, "null pointer argument");
  REC_REQUIRE(num_rows < (1ll << 31), REC_ESHAPE, "num_rows too large for the one-launch merge");
  const unsigned grid = (unsigned)((n + kSmallWaves - 1) / kSmallWaves);
  int fp_mode = 0;
  const size_t shmem = small_lds_bytes((size_t)n, true, &fp_mode);       // <= 64 KB
  hipStream_t st = (hipStream_t)stream;
  const SmallSgd up{P, row_stride, lr};
#define REC_SMALL(NACC_)                                                                                        \
  hipLaunchKernelGGL((sparse_small_kernel<NACC_, SmallSgd>), dim3(grid), dim3(kSmallWaves * kWave), shmem, st,   \
                     (int)n, emb_dim, 1, num_rows, padding_idx, ids, (const int64_t*)nullptr, grad, gl, up, status,   \
                     (const int32_t*)nullptr, fp_mode)
  if (emb_dim <= kWave) REC_SMALL(1); else if (emb_dim <= 2 * kWave) REC_SMALL(2); else REC_SMALL(4);
#undef REC_SMALL
  return check_launch("rec_sparse_sgd_small");
}

static int small_multi_impl(int32_t count, const rec_small_sgd_job* jobs, float lr, int32_t* status, void* stream,
                            int64_t dense_n, float* dense_p, const float* dense_g) {
  REC_REQUIRE(count >= 0 && count <= kSmallJobsMax && (count == 0 || jobs) && status, REC_EINVAL,
              "bad arguments (at most %d tables per call)", kSmallJobsMax);
  SmallJobs js;
  js.count = 0;
  int blocks = 0, dmax = 0;
  size_t nmax = 0;
  for (int i = 0; i < count; ++i) {
    const rec_small_sgd_job& a = jobs[i];
    REC_REQUIRE(a.n >= 0 && a.emb_dim > 0 && a.row_stride >= a.emb_dim && a.num_rows > 0 && a.grad_layout.div >= 1,
                REC_EINVAL, "job %d: bad sizes", i);
    REC_REQUIRE(a.n <= kSmallMergeMax, REC_ESHAPE, "job %d: n %lld > %d", i, (long long)a.n, kSmallMergeMax);
    REC_REQUIRE(a.emb_dim <= 4 * kWave && a.num_rows < (1ll << 31), REC_ESHAPE, "job %d: shape unsupported", i);
    REC_REQUIRE(a.grad_layout.group <= 0 || a.grad_layout.group_stride >= (int64_t)a.grad_layout.group * a.emb_dim,
                REC_EINVAL, "job %d: grad group_stride too small", i);
    if (a.n == 0) continue;
    REC_REQUIRE(a.ids && a.grad && a.P, REC_EINVAL, "job %d: null pointer argument", i);
    SmallJob& j = js.j[js.count++];
    j.n = (int)a.n; j.D = a.emb_dim; j.block0 = blocks; j.N = a.num_rows; j.pad = a.padding_idx;
    j.ids = a.ids; j.grad = a.grad; j.gl = a.grad_layout; j.gl.partials = nullptr;
    j.up = SmallSgd{a.P, a.row_stride, lr};
    blocks += (int)((a.n + kSmallWaves - 1) / kSmallWaves);
    dmax = a.emb_dim > dmax ? a.emb_dim : dmax;
    nmax = (size_t)a.n > nmax ? (size_t)a.n : nmax;
  }
  if (js.count == 0) return dense_n > 0 ? rec_sgd_dense(dense_n, dense_p, dense_g, lr, stream) : REC_OK;
  js.merge_blocks = blocks;
  js.dense_blocks = 0;
  js.dense_n = dense_n; js.dense_p = dense_p; js.dense_g = dense_g; js.dense_lr = lr;
  if (dense_n > 0) {
    const int64_t db = (dense_n + 4 * kSmallWaves * kWave - 1) / (4 * kSmallWaves * kWave);
    js.dense_blocks = (int)(db > 64 ? 64 : db);
  }
  const size_t shmem = small_lds_bytes(nmax, true, &js.fp_mode);   // (the layout is per job: its own n)
  hipStream_t st = (hipStream_t)stream;
#define REC_SMALLM(NACC_)                                                                                       \
  hipLaunchKernelGGL((sparse_small_multi_kernel<NACC_>), dim3((unsigned)(blocks + js.dense_blocks)), dim3(kSmallWaves * kWave), shmem, st, \
                     js, status)
  if (dmax <= kWave) REC_SMALLM(1); else if (dmax <= 2 * kWave) REC_SMALLM(2); else REC_SMALLM(4);
#undef REC_SMALLM
  return check_launch("rec_sparse_sgd_small_multi");
}

extern "C" int rec_sparse_sgd_small_multi(int32_t count, const rec_small_sgd_job* jobs, float lr, int32_t* status,
                                          void* stream) {
  return small_multi_impl(count, jobs, lr, status, stream, 0, nullptr, nullptr);
}

int rec::sparse_sgd_small_multi_dense(int32_t count, const rec_small_sgd_job* jobs, float lr, int32_t* status, void* stream,
                                      int64_t dense_n, float* dense_p, const float* dense_g) {
  REC_REQUIRE(dense_n >= 0 && (dense_n == 0 || (dense_p && dense_g)), REC_EINVAL, "bad dense arguments");
  return small_multi_impl(count, jobs, lr, status, stream, dense_n, dense_p, dense_g);
}

// REC_SMALL_BUCKET=0: the wave-per-lookup merge for every shape, =2 (lab): row buckets for slot tables of wide rows too;
// REC_SMALL_BUCKET_ROWS: lookups per bucket (64: 13 312 lookups = 208 blocks, one round on 256 CUs)
static bool bucket_path(int64_t n, int32_t emb_dim, const int64_t* slot_offset, int32_t num_slots) {
  static const int bucket_on = [] { const char* v = getenv("REC_SMALL_BUCKET"); return v ? atoi(v) : 1; }();
  return bucket_on && n > kBucketMin && (!slot_offset || num_slots <= 1 || emb_dim <= kQuadLanes || bucket_on == 2);
}
static unsigned bucket_count(int64_t n) {
  static const int per_bucket = [] { const char* v = getenv("REC_SMALL_BUCKET_ROWS"); const int x = v ? atoi(v) : 0; return x > 0 ? x : 64; }();
  return (unsigned)((n + per_bucket - 1) / per_bucket);
}

bool rec::small_tail_eligible(int64_t n, int32_t emb_dim, const int64_t* slot_offset, int32_t num_slots) {
  static const bool on = [] { const char* v = getenv("REC_SMALL_TAIL"); return !(v && *v == '0'); }();
  return on && n <= kSmallMergeMax && emb_dim <= kWave && bucket_path(n, emb_dim, slot_offset, num_slots);
}

int rec::sparse_adam_record_small_tail(int64_t n, int32_t num_slots, int32_t emb_dim, int32_t rec_stride,
                                       int32_t state_stride, int32_t v_offset, int64_t num_rows, int64_t padding_idx,
                                       const int64_t* ids, const int64_t* slot_offset, const float* grad,
                                       const rec_grad_layout* grad_layout, const float* grad1,
                                       const rec_grad_layout* grad1_layout, float* rec, float* MV,
                                       const rec_adam_hyper* hyper, int32_t* status, TailRoles roles, void* stream) {
  rec_grad_layout gl = {1, 0, 0, nullptr, nullptr}, gl1 = {1, 0, 0, nullptr, nullptr};
  if (grad_layout) gl = *grad_layout;
  if (grad1_layout) gl1 = *grad1_layout;
  gl.partials = gl1.partials = nullptr;
  REC_REQUIRE(small_tail_eligible(n, emb_dim, slot_offset, num_slots), REC_ESHAPE, "not a row-bucket shape");
  REC_REQUIRE(n > 0 && num_slots > 0 && emb_dim > 0 && rec_stride >= emb_dim + 3 && num_rows > 0 &&
                  num_rows < (1ll << 31) && gl.div >= 1 && gl1.div >= 1 && v_offset >= emb_dim &&
                  state_stride >= v_offset + emb_dim, REC_EINVAL, "bad sizes");
  REC_REQUIRE(ids && grad && grad1 && rec && MV && hyper && status && hyper->step >= 1, REC_EINVAL, "bad arguments");
  SmallAdamRecord up;
  up.rec = rec; up.MV = MV; up.stride = rec_stride; up.sstride = state_stride; up.v_off = v_offset;
  up.grad1 = grad1; up.gl1 = gl1; up.grad_scale = nullptr;
  adam_scalars(hyper, &up.lr_t, &up.eps_t);
  up.b1 = hyper->beta1; up.b2 = hyper->beta2;
  if (gl.group <= 0) { gl.group = 1; gl.group_stride = emb_dim; }
  roles.adam.lr_t = up.lr_t; roles.adam.eps_t = up.eps_t; roles.adam.b1 = up.b1; roles.adam.b2 = up.b2;
  const unsigned buckets = bucket_count(n);
  const unsigned grid = buckets + (unsigned)(roles.head_blocks + roles.fm_blocks + roles.w0_blocks + roles.rest_blocks);
  hipLaunchKernelGGL(small_tail_kernel, dim3(grid), dim3(kBucketWaves * kWave), 0, (hipStream_t)stream, buckets, (int)n,
                     emb_dim, num_slots, num_rows, padding_idx, ids, slot_offset, grad, gl, up, status, roles);
  return check_launch("small_tail_kernel");
}

extern "C" int rec_sparse_adam_record_small(int64_t n, int32_t num_slots, int32_t emb_dim, int32_t rec_stride,
                                            int32_t state_stride, int32_t v_offset, int64_t num_rows,
                                            int64_t padding_idx, const int64_t* ids, const int64_t* slot_offset,
                                            const float* grad, const rec_grad_layout* grad_layout,
                                            const float* grad1, const rec_grad_layout* grad1_layout,
                                            const float* grad_scale, float* rec, float* MV,
                                            const rec_adam_hyper* hyper, int32_t* status, int32_t* scratch, void* stream) {
  rec_grad_layout gl = {1, 0, 0, nullptr, nullptr}, gl1 = {1, 0, 0, nullptr, nullptr};
  if (grad_layout) gl = *grad_layout;
  if (grad1_layout) gl1 = *grad1_layout;
  gl.partials = gl1.partials = nullptr;
  REC_REQUIRE(n >= 0 && num_slots > 0 && emb_dim > 0 && rec_stride >= emb_dim + 3 && num_rows > 0 && gl.div >= 1 &&
                  gl1.div >= 1, REC_EINVAL, "bad sizes (the record holds W(D) | W1 | m1 | v1)");
  REC_REQUIRE(v_offset >= emb_dim && state_stride >= v_offset + emb_dim, REC_EINVAL,
              "state line must hold m(D) at 0 and v(D) at v_offset");
  REC_REQUIRE(n <= kSmallMergeMax, REC_ESHAPE, "n %lld > %d: use rec_ids_group + rec_sparse_adam_record",
              (long long)n, kSmallMergeMax);
  REC_REQUIRE(emb_dim <= 4 * kWave && num_rows < (1ll << 31), REC_ESHAPE, "shape unsupported by the one-launch merge");
  REC_REQUIRE(gl.group <= 0 || gl.group_stride >= (int64_t)gl.group * emb_dim, REC_EINVAL,
              "grad group_stride too small");
  if (n == 0) return REC_OK;
  REC_REQUIRE(ids && grad && grad1 && rec && MV && hyper && status, REC_EINVAL, "null pointer argument");
  REC_REQUIRE(hyper->step >= 1, REC_EINVAL, "Adam step must be >= 1");
  SmallAdamRecord up;
  up.rec = rec; up.MV = MV; up.stride = rec_stride; up.sstride = state_stride; up.v_off = v_offset;
  up.grad1 = grad1; up.gl1 = gl1; up.grad_scale = grad_scale;
  adam_scalars(hyper, &up.lr_t, &up.eps_t);
  up.b1 = hyper->beta1; up.b2 = hyper->beta2;
  if (gl.group <= 0) { gl.group = 1; gl.group_stride = emb_dim; }   // one D-wide row per position
  // by row buckets (sparse_bucket_kernel): ONE table whose lookups all meet each other, and slot tables of narrow rows
  // (the 16-lane form of the kernel's step 2: 26 x 512 lookups 0.115 -> 0.109 ms per step, and no span check — a row's
  // occurrences meet in its bucket whatever slot they come from).  Slot tables of wider rows keep the slot-major merge
  // below: a lookup is compared with its slot's n / S lookups only.
  if (bucket_path(n, emb_dim, slot_offset, num_slots)) {
    const unsigned buckets = bucket_count(n);
#define REC_BUCKET(NACC_)                                                                                             \
  hipLaunchKernelGGL((sparse_bucket_kernel<NACC_, SmallAdamRecord>), dim3(buckets), dim3(kBucketWaves * kWave), 0,      \
                     (hipStream_t)stream, (int)n, emb_dim, num_slots, num_rows, padding_idx, ids, slot_offset, grad, gl, \
                     up, status)
    if (emb_dim <= kWave) REC_BUCKET(1); else if (emb_dim <= 2 * kWave) REC_BUCKET(2); else REC_BUCKET(4);
#undef REC_BUCKET
    return check_launch("rec_sparse_adam_record_small (buckets)");
  }
  unsigned grid = (unsigned)((n + kSmallWaves - 1) / kSmallWaves);
  if (slot_offset && num_slots > 1 && n % num_slots == 0) {        // slot-major block mapping (sparse_small_body)
    const unsigned g2 = (unsigned)(num_slots * ((n / num_slots + kSmallWaves - 1) / kSmallWaves));
    if (g2 > grid) grid = g2;
  }
  int fp_mode = 0;
  const size_t shmem = small_lds_bytes((size_t)n, !(slot_offset && num_slots > 1 && n % num_slots == 0), &fp_mode);
  hipStream_t st = (hipStream_t)stream;
  int32_t* span_flag = nullptr;
  if (scratch && slot_offset && num_slots > 1 && n % num_slots == 0 && n > 2048) {    // (small lists: the scan is cheap)
    span_flag = scratch;
    hipLaunchKernelGGL(small_span_check_kernel, dim3(1), dim3(kSmallWaves * kWave), 0, st, (int)n, num_slots, ids,
                       slot_offset, span_flag);
  }
#define REC_SMALL(NACC_)                                                                                           \
  hipLaunchKernelGGL((sparse_small_kernel<NACC_, SmallAdamRecord>), dim3(grid), dim3(kSmallWaves * kWave), shmem,   \
                     st, (int)n, emb_dim, num_slots, num_rows, padding_idx, ids, slot_offset, grad, gl, up, status,    \
                     (const int32_t*)span_flag, fp_mode)
  if (emb_dim <= kWave) REC_SMALL(1); else if (emb_dim <= 2 * kWave) REC_SMALL(2); else REC_SMALL(4);
#undef REC_SMALL
  return check_launch("rec_sparse_adam_record_small");
}

extern "C" int rec_sgd_dense(int64_t n, float* p, const float* g, float lr, void* stream) {
  REC_REQUIRE(n >= 0 && (n == 0 || (p && g)), REC_EINVAL, "bad arguments");
  if (n == 0) return REC_OK;
  int64_t grid = (n + kBlock - 1) / kBlock;
  if (grid > kNumCU * 8) grid = kNumCU * 8;
  hipLaunchKernelGGL(sgd_dense_kernel, dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream, n, p, g, lr);
  return check_launch("rec_sgd_dense");
}

extern "C" int rec_adam_rows_all(int64_t num_rows, int32_t emb_dim, int32_t row_stride,
                                 int32_t state_stride, const int32_t* n_uniq, const int64_t* uniq_rows,
                                 const int32_t* seg_offset, const int32_t* sorted_pos, const float* grad,
                                 const rec_grad_layout* grad_layout, const float* grad_scale, float* P,
                                 float* M, float* V, const rec_adam_hyper* hyper, void* stream) {
  rec_grad_layout gl = {1, 0, 0, nullptr, nullptr};
  if (grad_layout) gl = *grad_layout;
  REC_REQUIRE(num_rows >= 0 && emb_dim > 0 && row_stride >= emb_dim && gl.div >= 1, REC_EINVAL, "bad sizes");
  if (state_stride <= 0) state_stride = row_stride;
  REC_REQUIRE(state_stride >= emb_dim, REC_EINVAL, "state_stride < emb_dim");
  REC_REQUIRE(n_uniq && uniq_rows && seg_offset && sorted_pos && grad && P && M && V && hyper, REC_EINVAL,
              "null pointer argument");
  REC_REQUIRE(hyper->step >= 1, REC_EINVAL, "Adam step must be >= 1");
  if (num_rows == 0) return REC_OK;
  float lr_t, eps_t;
  adam_scalars(hyper, &lr_t, &eps_t);
  const bool gvec = ((uintptr_t)grad) % 16 == 0 && (gl.group <= 0 || gl.group_stride % 4 == 0) &&
                    state_stride % 4 == 0;
  return dispatch_row_shape(emb_dim, gvec ? row_stride : row_stride | 1, [&](auto vec, auto lanes) -> int {
    constexpr int VEC = decltype(vec)::value, LANES = decltype(lanes)::value;
    constexpr int RB = kBlock / LANES;
    const int64_t grid = (num_rows + RB - 1) / RB;
    REC_REQUIRE(grid < (1ll << 31), REC_ESHAPE, "too many rows");
    hipLaunchKernelGGL((adam_rows_all_kernel<VEC, LANES>), dim3((unsigned)grid), dim3(kBlock), 0,
                       (hipStream_t)stream, num_rows, emb_dim, row_stride, state_stride, n_uniq, uniq_rows,
                       seg_offset, sorted_pos, grad, gl, grad_scale, P, M, V, lr_t, eps_t, hyper->beta1,
                       hyper->beta2);
    return check_launch("rec_adam_rows_all");
  });
}
