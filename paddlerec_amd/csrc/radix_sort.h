// Stable LSD radix sort of (key, int32 value) pairs for the engine's integer grouping work (gfx950) — hand-written
// replacement of the rocprim onesweep calls of round 1 (4 passes x 157 us for the 1.7 M lookups of a batch: those
// kernels are built for hundreds of millions of keys; here n is 10^6 .. 10^8 and the whole job is a few MB).
//
// One pass over digit [shift, shift+bits):
//   hist    : block = tile of kTile consecutive elements -> LDS histogram -> hist[bin][block]      (bin-major)
//   rowscan : block = one bin: exclusive scan of its row over the blocks, row total -> totals[bin]
//   scatter : block re-reads its tile; bin bases = scan(totals) (LDS); every wave owns a contiguous quarter of the
//             tile, counts it per bin, takes its base behind the earlier waves, then walks it 64 elements at a time
//             in order: peers of a digit are found with `bits` ballots, rank = popcount of the lower peers
//             -> stable, deterministic, no atomics on global memory.
// The first pass reads its keys through a functor (ids -> row keys computed on the fly: no key materialisation
// pass); passes ping-pong between two buffers so that the LAST pass writes the caller's destination.
#pragma once
#include <stdlib.h>
#include "rec_common.h"

namespace rec {
namespace rsort {

constexpr int kThreads = 256;          // rowscan / small helpers
constexpr int kChunks = 8;             // 64-element chunks per wave: a wave's keys live in registers
constexpr int kWaveSpan = kChunks * kWave;   // 512 consecutive elements per wave
constexpr int kMaxBits = 11;           // 2048 bins (4-wave blocks); 16-wave blocks take <= 9 bits

template <class KeyT>
struct BufSrc {
  const KeyT* keys;
  const int32_t* vals;
  __device__ __forceinline__ KeyT key(int64_t i) const { return keys[i]; }
  __device__ __forceinline__ int32_t val(int64_t i) const { return vals[i]; }
  __device__ __forceinline__ bool drop(KeyT) const { return false; }
};

// (key, value) pairs as ONE 8-byte element: what the intermediate passes of sort_pairs_packed read and write.  A
// scattered 4-byte store stream costs a pass ~20 us per 1.7 M elements on top of the first one (write combining breaks:
// profiles/r04_slot_group_probe.txt); a pass that scatters keys and values as two arrays pays that twice.
struct PairSrc {
  const uint2* pairs;
  __device__ __forceinline__ uint32_t key(int64_t i) const { return pairs[i].x; }
  __device__ __forceinline__ int32_t val(int64_t i) const { return (int32_t)pairs[i].y; }
  __device__ __forceinline__ bool drop(uint32_t) const { return false; }
};
template <class Src, class KeyT>
__device__ __forceinline__ void src_load(const Src& s, int64_t i, KeyT& k, int32_t& v) {
  k = s.key(i);
  v = s.val(i);
}
__device__ __forceinline__ void src_load(const PairSrc& s, int64_t i, uint32_t& k, int32_t& v) {
  const uint2 p = s.pairs[i];
  k = p.x;
  v = (int32_t)p.y;
}

// WAVES waves per block, tile = WAVES * 512 consecutive elements (2048 for small jobs: enough blocks to fill the
// chip at n ~ 10^6; 8192 for large ones: a smaller histogram matrix)
template <class KeyT, class Src, int WAVES>
__global__ __launch_bounds__(WAVES* kWave) void hist_kernel(int64_t n, int shift, int bits, int nblk, Src src,
                                                            int32_t* __restrict__ hist,
                                                            const int32_t* __restrict__ n_dev) {
  extern __shared__ int lh[];   // [1 << bits]
  if (n_dev) n = min(n, (int64_t)n_dev[0]);      // an earlier pass dropped the invalid keys: fewer elements are left
  constexpr int NT = WAVES * kWave;
  const int nb = 1 << bits;
  for (int i = threadIdx.x; i < nb; i += NT) lh[i] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * (WAVES * kWaveSpan);
  const KeyT mask = (KeyT)(nb - 1);
  int d[kChunks];
#pragma unroll
  for (int j = 0; j < kChunks; ++j) {   // all loads first, then the LDS atomics
    const int64_t i = base + j * NT + threadIdx.x;
    d[j] = -1;
    if (i < n) {
      const KeyT kk = src.key(i);
      if (!src.drop(kk)) d[j] = (int)((kk >> shift) & mask);
    }
  }
#pragma unroll
  for (int j = 0; j < kChunks; ++j)
    if (d[j] >= 0) atomicAdd(&lh[d[j]], 1);
  __syncthreads();
  for (int i = threadIdx.x; i < nb; i += NT) hist[(int64_t)i * nblk + blockIdx.x] = lh[i];
}

// exclusive scan of one bin's row (nblk entries) in place; totals[bin] = row sum
static __global__ __launch_bounds__(kThreads) void rowscan_kernel(int nblk, int32_t* __restrict__ hist,
                                                                  int32_t* __restrict__ totals) {
  __shared__ int part[kThreads];
  int32_t* row = hist + (int64_t)blockIdx.x * nblk;
  const int per = (nblk + kThreads - 1) / kThreads;
  const int lo = threadIdx.x * per, hi = min(lo + per, nblk);
  int s = 0;
  for (int i = lo; i < hi; ++i) s += row[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 1; o < kThreads; o <<= 1) {   // Hillis-Steele inclusive scan of the 256 partials
    const int v = (int)threadIdx.x >= o ? part[threadIdx.x - o] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int run = threadIdx.x ? part[threadIdx.x - 1] : 0;
  for (int i = lo; i < hi; ++i) {
    const int t = row[i];
    row[i] = run;
    run += t;
  }
  if (threadIdx.x == kThreads - 1) totals[blockIdx.x] = part[kThreads - 1];
}

template <class KeyT, class Src, int WAVES, bool PACK_OUT = false>
__global__ __launch_bounds__(WAVES* kWave) void scatter_kernel(int64_t n, int shift, int bits, int nblk, Src src,
                                                               const int32_t* __restrict__ hist,
                                                               const int32_t* __restrict__ totals,
                                                               KeyT* __restrict__ keys_out,   // PACK_OUT: a uint2 array
                                                               int32_t* __restrict__ vals_out,
                                                               const int32_t* __restrict__ n_dev,
                                                               int32_t* __restrict__ n_live_out) {
  extern __shared__ int sm[];
  if (n_dev) n = min(n, (int64_t)n_dev[0]);   // [nb] bin bases | [WAVES][nb] per-wave counters -> running offsets | [NT] scan
  constexpr int NT = WAVES * kWave;
  const int nb = 1 << bits;
  int* binbase = sm;
  int* whist = sm + nb;
  int* part = whist + WAVES * nb;
  const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
  const KeyT mask = (KeyT)(nb - 1);
  // this wave's 512 consecutive elements -> registers (coalesced, all loads in flight together)
  const int64_t wbase = (int64_t)blockIdx.x * (WAVES * kWaveSpan) + (int64_t)wave * kWaveSpan;
  KeyT k[kChunks];
  int32_t v[kChunks];
  int d[kChunks];
#pragma unroll
  for (int c = 0; c < kChunks; ++c) {
    const int64_t i = wbase + c * kWave + lane;
    const bool in = i < n;
    k[c] = (KeyT)0;
    v[c] = 0;
    if (in) src_load(src, i, k[c], v[c]);
    d[c] = (in && !src.drop(k[c])) ? (int)((k[c] >> shift) & mask) : -1;
  }
  // bin bases: exclusive scan of the totals (serial per thread + block scan)
  {
    const int per = (nb + NT - 1) / NT;
    const int lo = threadIdx.x * per, hi = min(lo + per, nb);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += totals[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < NT; o <<= 1) {
      const int t = (int)threadIdx.x >= o ? part[threadIdx.x - o] : 0;
      __syncthreads();
      part[threadIdx.x] += t;
      __syncthreads();
    }
    int run = threadIdx.x ? part[threadIdx.x - 1] : 0;
    for (int i = lo; i < hi; ++i) {
      binbase[i] = run;
      run += totals[i];
    }
    // the elements this pass keeps (all of them unless the source drops keys): what the later passes sort
    if (n_live_out && blockIdx.x == 0 && threadIdx.x == NT - 1) n_live_out[0] = part[NT - 1];
  }
  for (int i = threadIdx.x; i < WAVES * nb; i += NT) whist[i] = 0;
  __syncthreads();
  int* wh = whist + wave * nb;
#pragma unroll
  for (int c = 0; c < kChunks; ++c)
    if (d[c] >= 0) atomicAdd(&wh[d[c]], 1);
  __syncthreads();
  // counters -> first output index of (wave, bin): bin base + this block's prefix + earlier waves of the block
  for (int b = threadIdx.x; b < nb; b += NT) {
    int run = binbase[b] + hist[(int64_t)b * nblk + blockIdx.x];
    for (int w = 0; w < WAVES; ++w) {
      const int t = whist[w * nb + b];
      whist[w * nb + b] = run;
      run += t;
    }
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < kChunks; ++c) {
    const bool in = d[c] >= 0;
    // lanes holding the same digit (and a real element)
    unsigned long long peers = __ballot(in);
    for (int b = 0; b < bits; ++b) {
      const unsigned long long bal = __ballot((d[c] >> b) & 1);
      peers &= ((d[c] >> b) & 1) ? bal : ~bal;
    }
    const int rank = __popcll(peers & ((1ull << lane) - 1ull));
    int off = 0;
    if (in) off = wh[d[c]];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (in) {
      if constexpr (PACK_OUT) {
        reinterpret_cast<uint2*>(keys_out)[(int64_t)off + rank] = make_uint2((uint32_t)k[c], (uint32_t)v[c]);
      } else {
        keys_out[(int64_t)off + rank] = k[c];
        vals_out[(int64_t)off + rank] = v[c];
      }
      if (rank == __popcll(peers) - 1) wh[d[c]] = off + rank + 1;   // last peer advances the running offset
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

struct Plan {
  int passes, bits[8], shift[8];
  int waves;                         // 4 (tile 2048) or 16 (tile 8192)
  int nblk;
  size_t hist_bytes, totals_bytes;   // workspace: [max nb][nblk] i32 | [max nb] i32
};

inline Plan make_plan(int64_t n, int key_bits_) {
  Plan p;
  if (key_bits_ < 1) key_bits_ = 1;
  // digit width: 9 bits (512 bins) unless wider digits save a whole pass AND the keys are few enough for the 4-wave
  // blocks that take them (REC_RSORT_DIGIT=<9..11> forces one width: A/B runs)
  static const int digit_env = [] { const char* v = getenv("REC_RSORT_DIGIT"); return v && *v ? atoi(v) : 0; }();
  int digit = 9;
  if (digit_env >= 9 && digit_env <= kMaxBits) digit = digit_env;
  p.passes = key_bits_ <= kMaxBits ? 1 : (key_bits_ + digit - 1) / digit;
  int left = key_bits_, sh = 0, maxb = 0;
  for (int i = 0; i < p.passes; ++i) {
    const int b = (left + (p.passes - i) - 1) / (p.passes - i);
    p.bits[i] = b;
    p.shift[i] = sh;
    sh += b;
    left -= b;
    if (b > maxb) maxb = b;
  }
  p.waves = (n > (int64_t)8 * 1024 * 1024 && maxb <= 9) ? 16 : 4;
  const int64_t tile = (int64_t)p.waves * kWaveSpan;
  p.nblk = (int)((n + tile - 1) / tile);
  if (p.nblk < 1) p.nblk = 1;
  p.hist_bytes = align_up(((size_t)1 << maxb) * p.nblk * sizeof(int32_t), 256);
  p.totals_bytes = align_up(((size_t)1 << maxb) * sizeof(int32_t), 256);
  return p;
}

template <class KeyT, class Src, int WAVES, bool PACK_OUT = false>
inline void run_pass(int64_t n, const Plan& p, int i, Src src, KeyT* ko, int32_t* vo, int32_t* hist,
                     int32_t* totals, hipStream_t st, const int32_t* n_dev = nullptr, int32_t* n_live_out = nullptr) {
  const int nb = 1 << p.bits[i];
  const size_t sm_scatter = (size_t)(nb + WAVES * nb + WAVES * kWave) * sizeof(int);
  hipLaunchKernelGGL((hist_kernel<KeyT, Src, WAVES>), dim3(p.nblk), dim3(WAVES * kWave), nb * sizeof(int), st, n,
                     p.shift[i], p.bits[i], p.nblk, src, hist, n_dev);
  hipLaunchKernelGGL(rowscan_kernel, dim3(nb), dim3(kThreads), 0, st, p.nblk, hist, totals);
  hipLaunchKernelGGL((scatter_kernel<KeyT, Src, WAVES, PACK_OUT>), dim3(p.nblk), dim3(WAVES * kWave), sm_scatter, st, n,
                     p.shift[i], p.bits[i], p.nblk, src, hist, totals, ko, vo, n_dev, n_live_out);
}

// after a sort that dropped keys: positions [n_live, n) of the sorted keys read as `tail` (the callers' sentinel)
template <class KeyT>
__global__ void fill_tail_kernel(int64_t n, const int32_t* __restrict__ n_live, KeyT tail, KeyT* __restrict__ keys) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && i >= n_live[0]) keys[i] = tail;
}

// Sorts n pairs by key bits [0, key_bits).  Pass 0 reads `first` (functor); later passes ping-pong between
// (keys_tmp, vals_tmp) and (keys_dst, vals_dst) such that the last pass writes the *_dst buffers.
template <class KeyT, class FirstSrc>
inline int sort_pairs(int64_t n, const Plan& p, FirstSrc first, KeyT* keys_tmp, int32_t* vals_tmp, KeyT* keys_dst,
                      int32_t* vals_dst, void* ws_hist, void* ws_totals, hipStream_t st, int32_t* n_live = nullptr,
                      KeyT tail_key = 0) {
  // n_live (device int, optional): the first pass leaves out the keys `first.drop()` names (padding / out-of-range
  // lookups: 44 % of the ids of a multi-slot batch), writes how many it kept, the later passes sort only those, and the
  // tail of keys_dst reads as tail_key — what the callers' sentinel-sorts-last convention produced, at a fraction of
  // the traffic.  vals_dst behind the kept elements is unspecified.
  int32_t* hist = (int32_t*)ws_hist;
  int32_t* totals = (int32_t*)ws_totals;
  for (int i = 0; i < p.passes; ++i) {
    const bool to_dst = ((p.passes - 1 - i) % 2) == 0;
    KeyT* ko = to_dst ? keys_dst : keys_tmp;
    int32_t* vo = to_dst ? vals_dst : vals_tmp;
    if (i == 0) {
      if (p.waves == 16) run_pass<KeyT, FirstSrc, 16>(n, p, i, first, ko, vo, hist, totals, st, nullptr, n_live);
      else run_pass<KeyT, FirstSrc, 4>(n, p, i, first, ko, vo, hist, totals, st, nullptr, n_live);
    } else {
      BufSrc<KeyT> src{to_dst ? keys_tmp : keys_dst, to_dst ? vals_tmp : vals_dst};
      if (p.waves == 16) run_pass<KeyT, BufSrc<KeyT>, 16>(n, p, i, src, ko, vo, hist, totals, st, n_live);
      else run_pass<KeyT, BufSrc<KeyT>, 4>(n, p, i, src, ko, vo, hist, totals, st, n_live);
    }
  }
  if (n_live)
    hipLaunchKernelGGL(fill_tail_kernel<KeyT>, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, n,
                       (const int32_t*)n_live, tail_key, keys_dst);
  return check_launch("radix sort");
}

// sort_pairs for 32-bit keys and >= 2 passes with the intermediate results as packed (key, value) pairs: pass i < last
// writes ONE scattered 8-byte stream into pairs_a / pairs_b (n uint2 each, ping-pong; the last pass reads pairs_b, so
// keys_dst may alias pairs_a), the last pass writes keys_dst / vals_dst as before.  Same order, same results.
template <class FirstSrc>
inline int sort_pairs_packed(int64_t n, const Plan& p, FirstSrc first, uint2* pairs_a, uint2* pairs_b, uint32_t* keys_dst,
                             int32_t* vals_dst, void* ws_hist, void* ws_totals, hipStream_t st, int32_t* n_live = nullptr,
                             uint32_t tail_key = 0) {
  int32_t* hist = (int32_t*)ws_hist;
  int32_t* totals = (int32_t*)ws_totals;
  const int last = p.passes - 1;
  auto buf_of = [&](int i) { return ((last - 1 - i) % 2 == 0) ? pairs_b : pairs_a; };   // output of intermediate pass i
  for (int i = 0; i < p.passes; ++i) {
    if (i == 0) {          // first pass (never the last one here: passes >= 2)
      uint32_t* ko = (uint32_t*)buf_of(0);
      if (p.waves == 16) run_pass<uint32_t, FirstSrc, 16, true>(n, p, i, first, ko, nullptr, hist, totals, st, nullptr, n_live);
      else run_pass<uint32_t, FirstSrc, 4, true>(n, p, i, first, ko, nullptr, hist, totals, st, nullptr, n_live);
    } else if (i < last) {
      PairSrc src{buf_of(i - 1)};
      uint32_t* ko = (uint32_t*)buf_of(i);
      if (p.waves == 16) run_pass<uint32_t, PairSrc, 16, true>(n, p, i, src, ko, nullptr, hist, totals, st, n_live);
      else run_pass<uint32_t, PairSrc, 4, true>(n, p, i, src, ko, nullptr, hist, totals, st, n_live);
    } else {
      PairSrc src{buf_of(i - 1)};
      if (p.waves == 16) run_pass<uint32_t, PairSrc, 16, false>(n, p, i, src, keys_dst, vals_dst, hist, totals, st, n_live);
      else run_pass<uint32_t, PairSrc, 4, false>(n, p, i, src, keys_dst, vals_dst, hist, totals, st, n_live);
    }
  }
  if (n_live)
    hipLaunchKernelGGL(fill_tail_kernel<uint32_t>, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, n,
                       (const int32_t*)n_live, tail_key, keys_dst);
  return check_launch("radix sort");
}

}  // namespace rsort
}  // namespace rec
