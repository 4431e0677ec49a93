// SelectedRows merge keys of a [B, S] id batch whose slot s owns the rows [s * R, (s + 1) * R) of the table —
// BASELINE config 2a: 26 tables x 1 000 000 rows as one 26 M-row table (gfx950).
//
// Reference behaviour replaced: the duplicate-row merge of the `sparse=True` embedding gradient
// (/root/reference/models/rank/deepfm/net.py:62-70,80), integer part.
//
// rec_ids_group sorts the B*S lookups by their 25-bit global row: three LSD passes of hist / rowscan / scatter plus the
// head kernels — 13 launches that cost ~560 us of GPU time under the forward GEMMs of a DeepFM step.  The
// slot digit is free: column s of the batch IS slot s, so the sorted order is the concatenation of S independent sorts
// of B keys below 2^20.  Here:
//   keys    : [B,S] i64 -> [S][B] u32 slot-local keys (LDS transpose; padding / out-of-range -> no key, rank = -1)
//   hist    : block (slot, chunk of 8192) -> digit histogram [slot][chunk][bin]
//   scatter : the block scans its slot's [chunks][bins] matrix itself (8 x 1024 counters from L2: no rowscan launch,
//             no global bin totals), ranks its 8192 keys with ballots (stable) and scatters inside the slot's 256 KB
//             (L2-resident); two digits of <= 10 bits; the final pass writes sorted_pos, the global rows and — new —
//             rank[pos] = sorted index, the inverse permutation (fm_bwd writes its row gradients in SORTED order through
//             it, so the row-update kernel streams them: rec_grad_layout.sorted)
//   heads   : count per 2048-position tile, then emit — every emit block sums the counts in front of it (832 ints):
//             no scan launch.
// 7 launches, every block independent of the others inside a launch (no look-back, no spin): safe beside any other
// kernel.  Bit-identical to rec_ids_group with slot_offset[s] = s * R (same (row, position) order).
#include "slot_group.h"

namespace rec {
namespace sg {

constexpr int kWaves = 8;
constexpr int kThreads = kWaves * kWave;        // 512
constexpr int kChunks = 16;                     // 64-key chunks per wave (registers)
constexpr int kWaveSpan = kChunks * kWave;      // 1024 consecutive keys per wave
constexpr int kTile = kWaves * kWaveSpan;       // 8192 keys per block
constexpr uint32_t kNoKey = 0xFFFFFFFFu;
constexpr int kMaxDigit = 10;                   // 1024 bins: 8 waves x 1024 counters = 32 KB of LDS
constexpr int kKeyRows = 256;                   // samples per block of the transpose
constexpr int kMaxSlots = 60;                   // transpose tile [256][S | 1] u32 <= 64 KB
constexpr int kMaxChunks = 32;                  // every scatter block reads its slot's [chunks][bins] matrix
constexpr int kHeadThreads = 256, kHeadChunks = 8, kHeadWaves = kHeadThreads / kWave;
constexpr int kHeadTile = kHeadThreads * kHeadChunks;   // 2048 sorted positions per block
constexpr int kSegLong = REC_SEG_LONG;

static int key_bits(int64_t R) {   // keys 0 .. R-1
  int b = 1;
  while (b < 62 && (1ll << b) < R) ++b;
  return b;
}

struct Plan {
  int bits1, bits2, C, nblk_heads;
  size_t off_keyT, off_pairs1, off_pairs2, off_hist, off_bvalid, off_cnt, total;
};

static Plan make_plan(int64_t B, int S, int64_t R) {
  Plan p;
  const int kb = key_bits(R);
  p.bits1 = kb <= kMaxDigit ? kb : (kb + 1) / 2;
  p.bits2 = kb - p.bits1;
  p.C = (int)((B + kTile - 1) / kTile);
  const int64_t n = B * S;
  p.nblk_heads = (int)((n + kHeadTile - 1) / kHeadTile);
  const int nbmax = 1 << p.bits1;
  size_t o = 0;
  p.off_keyT = o; o += align_up((size_t)n * 4, 256);
  p.off_pairs1 = o; o += align_up((size_t)n * 8, 256);
  p.off_pairs2 = o; o += align_up((size_t)(n + 1) * 8, 256);
  p.off_hist = o; o += align_up((size_t)S * p.C * nbmax * 4, 256);
  p.off_bvalid = o; o += align_up((size_t)S * p.C * 4, 256);
  p.off_cnt = o; o += align_up((size_t)(p.nblk_heads + 1) * 4, 256);
  p.total = o;
  return p;
}

bool eligible(int64_t B, int32_t S, int64_t R) {
  static const bool on = [] { const char* v = getenv("REC_GROUP_SLOTS"); return !(v && *v == '0'); }();
  if (!on || B < kTile || S < 1 || S > kMaxSlots || R < 2) return false;
  if (B * S >= (1ll << 31) - 1 || (B + kTile - 1) / kTile > kMaxChunks) return false;
  if (key_bits(R) > 2 * kMaxDigit) return false;
  if ((double)S * (double)R >= 4294967295.0) return false;     // global rows travel as u32
  return true;
}

size_t workspace_bytes(int64_t B, int32_t S, int64_t R) { return make_plan(B, S, R).total; }

// ------------------------------------------------------------------------------------------------ keys
// ids [B,S] (coalesced) -> keyT [S][B] (coalesced, 256-sample runs); lookups without a key get rank -1.
__global__ __launch_bounds__(kKeyRows) void keys_kernel(int64_t B, int S, int64_t R, int64_t pad,
                                                        const int64_t* __restrict__ ids,
                                                        uint32_t* __restrict__ keyT, int32_t* __restrict__ rank,
                                                        int32_t* __restrict__ status, int32_t* __restrict__ n_uniq) {
  extern __shared__ uint32_t key_tile[];   // [kKeyRows][S | 1]
  const int pitch = S | 1;
  const int tid = threadIdx.x;
  if (blockIdx.x == 0 && tid < 2) n_uniq[2 + tid] = 0;
  const int64_t b0 = (int64_t)blockIdx.x * kKeyRows;
  const int nbr = (int)((B - b0 < kKeyRows) ? B - b0 : kKeyRows);
  const int total = nbr * S;
  const int64_t* src = ids + b0 * S;
  int r = tid / S, s = tid % S;
  const int dr = kKeyRows / S, ds = kKeyRows % S;
  int oob = 0;
  // eight ids in flight per thread (one dependent load per trip took 26 trips of a memory latency that is several
  // microseconds beside the lookup kernel this one runs next to)
  constexpr int U = 8;
  for (int i0 = tid; i0 < total; i0 += U * kKeyRows) {
    int64_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * kKeyRows;
      v[u] = src[i < total ? i : tid];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * kKeyRows;
      if (i < total) {
        const int64_t id = v[u];
        uint32_t key = kNoKey;
        if (id != pad || pad < 0) {
          if (id >= 0 && id < R) key = (uint32_t)id; else oob = 1;
        }
        key_tile[r * pitch + s] = key;
        if (rank && key == kNoKey) rank[b0 * S + i] = -1;
        r += dr; s += ds;
        if (s >= S) { s -= S; ++r; }
      }
    }
  }
  if (oob) atomicOr(status, REC_FLAG_INDEX_OOB);
  __syncthreads();
  if (tid < nbr)
    for (int q = 0; q < S; ++q) keyT[(int64_t)q * B + b0 + tid] = key_tile[tid * pitch + q];
}

// real keys of slot s = sum of its chunks' counts (written by the first histogram pass: no atomics anywhere)
__device__ __forceinline__ int slot_valid(const int32_t* __restrict__ bvalid, int C, int s) {
  int t = 0;
  for (int c = 0; c < C; ++c) t += bvalid[s * C + c];
  return t;
}

// ------------------------------------------------------------------------------------------------ digit histogram
// block (slot, chunk): hist[(slot * C + chunk) * nb + bin].  FIRST: input keyT, keys may be missing (kNoKey), the
// chunk's number of real keys goes to bvalid[slot * C + chunk]; otherwise input = the .x of the first pass's pairs, the
// slot's first slot_valid entries are all real.
template <bool FIRST>
__global__ __launch_bounds__(kThreads) void hist_kernel(int64_t B, int C, int shift, int bits,
                                                        const uint32_t* __restrict__ keys,
                                                        const uint2* __restrict__ pairs,
                                                        int32_t* __restrict__ hist, int32_t* __restrict__ bvalid) {
  extern __shared__ int lh[];      // [nb] + [kWaves]
  const int s = blockIdx.x / C, c = blockIdx.x % C;
  const int nb = 1 << bits;
  const int tid = threadIdx.x;
  for (int i = tid; i < nb; i += kThreads) lh[i] = 0;
  __syncthreads();
  const int64_t lim = FIRST ? B : (int64_t)slot_valid(bvalid, C, s);
  const int64_t base = (int64_t)c * kTile;
  uint32_t kk[kChunks];
#pragma unroll
  for (int j = 0; j < kChunks; ++j) {
    const int64_t idx = base + j * kThreads + tid;
    kk[j] = kNoKey;
    if (idx < lim) kk[j] = FIRST ? keys[(int64_t)s * B + idx] : pairs[(int64_t)s * B + idx].x;
  }
  int cnt = 0;
#pragma unroll
  for (int j = 0; j < kChunks; ++j)
    if (kk[j] != kNoKey) {
      atomicAdd(&lh[(kk[j] >> shift) & (uint32_t)(nb - 1)], 1);
      ++cnt;
    }
  if (FIRST) {
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, kWave);
    if (tid % kWave == 0) lh[nb + tid / kWave] = cnt;
  }
  __syncthreads();
  for (int i = tid; i < nb; i += kThreads) hist[(int64_t)blockIdx.x * nb + i] = lh[i];
  if (FIRST && tid == 0) {
    int t = 0;
    for (int w = 0; w < kWaves; ++w) t += lh[nb + w];
    bvalid[blockIdx.x] = t;
  }
}

// exclusive scan of arr[0, nb) (LDS, nb <= 2 * kThreads) in place by the whole block; wtot: kWaves ints of scratch
__device__ __forceinline__ void block_excl_scan(int* arr, int nb, int* wtot) {
  const int tid = threadIdx.x, lane = tid % kWave, wave = tid / kWave;
  const int per = nb > kThreads ? 2 : 1;
  const int lo = tid * per;
  int v0 = lo < nb ? arr[lo] : 0;
  int v1 = (per > 1 && lo + 1 < nb) ? arr[lo + 1] : 0;
  const int s = v0 + v1;
  int x = s;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const int y = __shfl_up(x, o, kWave);
    if (lane >= o) x += y;
  }
  if (lane == kWave - 1) wtot[wave] = x;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += wtot[w];
  const int run = base + x - s;
  if (lo < nb) arr[lo] = run;
  if (per > 1 && lo + 1 < nb) arr[lo + 1] = run + v0;
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------ scatter
// block (slot, chunk): stable scatter of its 8192 keys by digit [shift, shift + bits) inside the slot.
//   FIRST: input keyT (kNoKey = no key), the value of a key is its sample index; else input pairs_in {key, sample},
//          the slot's first slot_valid entries.
//   FINAL: output position = real keys of the slots in front + slot-local position, output pair {position b*S+s,
//          global row}; else {key, sample} inside the slot.
// One 8-byte store per key: two 4-byte streams to two arrays cost 20 us more per launch than one (measured,
// profiles/r04_slot_group_probe.txt) — the second stream breaks the write combining of the first.
template <bool FIRST, bool FINAL>
__global__ __launch_bounds__(kThreads) void scatter_kernel(int64_t B, int S, int C, int shift, int bits, int64_t R,
                                                           const uint32_t* __restrict__ keys_in,
                                                           const uint2* __restrict__ pairs_in,
                                                           const int32_t* __restrict__ hist,
                                                           const int32_t* __restrict__ bvalid,
                                                           uint2* __restrict__ pairs_out) {
  extern __shared__ int sm[];
  const int s = blockIdx.x / C, c = blockIdx.x % C;
  const int nb = 1 << bits;
  int* binbase = sm;                 // [nb]   bin totals of the slot -> exclusive scan
  int* pre = sm + nb;                // [nb]   keys of the bin in the chunks in front of this one
  int* whist = pre + nb;             // [kWaves][nb] per-wave counters -> running output offsets
  int* wtot = whist + kWaves * nb;   // [kWaves + 1]
  const int tid = threadIdx.x, lane = tid % kWave, wave = tid / kWave;
  const uint32_t mask = (uint32_t)(nb - 1);
  const int64_t lim = FIRST ? B : (int64_t)slot_valid(bvalid, C, s);
  // this wave's 1024 consecutive keys -> registers (coalesced, all loads in flight together)
  const int64_t wbase = (int64_t)c * kTile + (int64_t)wave * kWaveSpan;
  uint32_t k[kChunks];
  int32_t v[kChunks];
  int d[kChunks];
#pragma unroll
  for (int j = 0; j < kChunks; ++j) {
    const int64_t idx = wbase + j * kWave + lane;
    // (branch-free: a load inside `if (idx < lim)` is drained at the join — the wave's 16 fetches became 16 round trips)
    const bool in = idx < lim;
    const int64_t src = (int64_t)s * B + (in ? idx : 0);
    if (FIRST) {
      const uint32_t t = keys_in[src];
      k[j] = in ? t : kNoKey;
      v[j] = (int32_t)idx;
    } else {
      const uint2 t = pairs_in[src];
      k[j] = in ? t.x : kNoKey;
      v[j] = in ? (int32_t)t.y : (int32_t)idx;
    }
  }
#pragma unroll
  for (int j = 0; j < kChunks; ++j) d[j] = k[j] != kNoKey ? (int)((k[j] >> shift) & mask) : -1;
  // the slot's [C][nb] histogram matrix -> bin totals and the prefix of this chunk
  for (int b = tid; b < nb; b += kThreads) {
    int tot = 0, p = 0;
    const int32_t* h = hist + (int64_t)s * C * nb + b;
    for (int c0 = 0; c0 < C; c0 += 8) {                 // eight column entries in flight (C = 8 at batch 65536)
      int t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = h[(int64_t)(c0 + u < C ? c0 + u : c0) * nb];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int x = c0 + u < C ? t[u] : 0;
        tot += x;
        if (c0 + u < c) p += x;
      }
    }
    binbase[b] = tot;
    pre[b] = p;
  }
  for (int i = tid; i < kWaves * nb; i += kThreads) whist[i] = 0;
  if (FINAL && tid < kWave) {   // output base of the slot: the real keys of the slots in front of it
    int t = 0;
    for (int q = lane; q < s * C; q += kWave) t += bvalid[q];
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) t += __shfl_xor(t, o, kWave);
    if (lane == 0) wtot[kWaves] = t;
  }
  __syncthreads();
  block_excl_scan(binbase, nb, wtot);
  int* wh = whist + wave * nb;
#pragma unroll
  for (int j = 0; j < kChunks; ++j)
    if (d[j] >= 0) atomicAdd(&wh[d[j]], 1);
  __syncthreads();
  for (int b = tid; b < nb; b += kThreads) {
    int run = binbase[b] + pre[b];
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
      const int t = whist[w * nb + b];
      whist[w * nb + b] = run;
      run += t;
    }
  }
  __syncthreads();
  uint2* out = pairs_out + (FINAL ? (int64_t)wtot[kWaves] : (int64_t)s * B);
  const uint32_t row0 = (uint32_t)((int64_t)s * R);
#pragma unroll
  for (int j = 0; j < kChunks; ++j) {
    const bool in = d[j] >= 0;
    // lanes holding the same digit: kMaxDigit ballots (bits above `bits` are 0 in every real digit)
    unsigned long long peers = __ballot(in);
#pragma unroll
    for (int b = 0; b < kMaxDigit; ++b) {
      const unsigned long long bal = __ballot((d[j] >> b) & 1);
      peers &= ((d[j] >> b) & 1) ? bal : ~bal;
    }
    const int rk = __popcll(peers & ((1ull << lane) - 1ull));
    int off = 0;
    if (in) off = wh[d[j]];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (in) {
      const int o = off + rk;
      out[o] = FINAL ? make_uint2((uint32_t)(v[j] * S + s), row0 + k[j]) : make_uint2(k[j], (uint32_t)v[j]);
      if (rk == __popcll(peers) - 1) wh[d[j]] = o + 1;   // last peer advances the running offset
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// ------------------------------------------------------------------------------------------------ heads
__device__ __forceinline__ int total_valid(int nbv, const int32_t* __restrict__ bvalid, int* sh) {
  if (threadIdx.x < kWave) {
    int t = 0;
    for (int q = threadIdx.x; q < nbv; q += kWave) t += bvalid[q];
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) t += __shfl_xor(t, o, kWave);
    if (threadIdx.x == 0) *sh = t;
  }
  __syncthreads();
  return *sh;
}

// sorted pair i = {position, global row}.  The fetches are branch-free (clamped index, every lane): a load inside a
// divergent `if` is drained at the join, and the eight chunks of a block's tile were eight dependent round trips.
__device__ __forceinline__ void head_fetch(int n, const uint2* __restrict__ pairs, int64_t i, uint2* pair, uint32_t* prev) {
  const int64_t last = n > 0 ? n - 1 : 0;
  *pair = pairs[i < n ? i : last];
  *prev = pairs[(i > 0 && i < n) ? i - 1 : (i < n ? i : last)].y;
}
__device__ __forceinline__ unsigned long long head_eval(int n, int64_t i, int lane, uint2* pair, uint32_t prev) {
  const bool in = i < n;
  if (!in) *pair = make_uint2(0u, kNoKey);
  uint32_t left = __shfl_up(pair->y, 1, kWave);
  if (lane == 0) left = (i > 0 && in) ? prev : kNoKey;
  return __ballot(in && (i == 0 || pair->y != left));
}

__global__ __launch_bounds__(kHeadThreads) void heads_count_kernel(int nbv, const int32_t* __restrict__ bvalid,
                                                                   const uint2* __restrict__ pairs,
                                                                   int32_t* __restrict__ cnt,
                                                                   int32_t* __restrict__ n_uniq) {
  __shared__ int red[kHeadWaves];
  __shared__ int nsh;
  const int n = total_valid(nbv, bvalid, &nsh);
  const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
  const int64_t base = (int64_t)blockIdx.x * kHeadTile;
  int heads = 0, has_long = 0;
  if (base < n) {
    uint2 pp[kHeadChunks];
    uint32_t prev[kHeadChunks], far[kHeadChunks];
#pragma unroll
    for (int c = 0; c < kHeadChunks; ++c) {
      const int64_t i = base + (int64_t)(c * kHeadWaves + wave) * kWave + lane;
      head_fetch(n, pairs, i, &pp[c], &prev[c]);
      far[c] = pairs[i + kSegLong - 1 < n ? i + kSegLong - 1 : (n > 0 ? n - 1 : 0)].y;
    }
#pragma unroll
    for (int c = 0; c < kHeadChunks; ++c) {
      const int64_t i = base + (int64_t)(c * kHeadWaves + wave) * kWave + lane;
      heads += __popcll(head_eval(n, i, lane, &pp[c], prev[c]));
      if (i + kSegLong - 1 < n && far[c] == pp[c].y) has_long = 1;
    }
  }
  if (__ballot(has_long) != 0 && lane == 0) atomicOr(&n_uniq[2], 1);
  if (lane == 0) red[wave] = heads;
  __syncthreads();
  if (threadIdx.x == 0) {
    int h = 0;
    for (int w = 0; w < kHeadWaves; ++w) h += red[w];
    cnt[blockIdx.x] = h;
  }
}

// ... and the outputs: sorted_pos (coalesced), rank[pos] = sorted index (the one scattered 4-byte stream of this
// kernel), uniq / seg_off at the heads
__global__ __launch_bounds__(kHeadThreads) void heads_emit_kernel(int nbv, const int32_t* __restrict__ bvalid,
                                                                  const uint2* __restrict__ pairs,
                                                                  const int32_t* __restrict__ cnt,
                                                                  int32_t* __restrict__ sorted_pos,
                                                                  int32_t* __restrict__ rank,
                                                                  int64_t* __restrict__ uniq,
                                                                  int32_t* __restrict__ seg_off,
                                                                  int32_t* __restrict__ n_uniq) {
  __shared__ int ccnt[kHeadChunks * kHeadWaves];
  __shared__ int red[kHeadWaves];
  __shared__ int nsh;
  const int n = total_valid(nbv, bvalid, &nsh);
  const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
  const int64_t base = (int64_t)blockIdx.x * kHeadTile;
  const bool last = blockIdx.x == gridDim.x - 1;
  if (base >= n && !last) return;
  // heads in front of this tile: every block sums the per-tile counts itself (<= a few thousand ints from L2)
  int t = 0;
  for (int q = threadIdx.x; q < (int)blockIdx.x; q += kHeadThreads) t += cnt[q];
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) t += __shfl_xor(t, o, kWave);
  if (lane == 0) red[wave] = t;
  unsigned long long hm[kHeadChunks];
  uint2 pp[kHeadChunks];
  uint32_t prev[kHeadChunks];
#pragma unroll
  for (int c = 0; c < kHeadChunks; ++c)
    head_fetch(n, pairs, base + (int64_t)(c * kHeadWaves + wave) * kWave + lane, &pp[c], &prev[c]);
#pragma unroll
  for (int c = 0; c < kHeadChunks; ++c) {
    const int64_t i = base + (int64_t)(c * kHeadWaves + wave) * kWave + lane;
    hm[c] = head_eval(n, i, lane, &pp[c], prev[c]);
    if (lane == 0) ccnt[c * kHeadWaves + wave] = __popcll(hm[c]);
    if (i < n) {
      sorted_pos[i] = (int32_t)pp[c].x;
      if (rank) rank[pp[c].x] = (int32_t)i;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {   // exclusive scan over the tile's 32 chunks (in position order)
    int run = 0;
    for (int w = 0; w < kHeadWaves; ++w) run += red[w];
    for (int q = 0; q < kHeadChunks * kHeadWaves; ++q) {
      const int x = ccnt[q];
      ccnt[q] = run;
      run += x;
    }
    if (last) {
      n_uniq[0] = run;
      n_uniq[1] = n;
      seg_off[run] = n;
    }
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < kHeadChunks; ++c) {
    if ((hm[c] >> lane) & 1ull) {
      const int64_t i = base + (int64_t)(c * kHeadWaves + wave) * kWave + lane;
      const int u = ccnt[c * kHeadWaves + wave] + __popcll(hm[c] & ((1ull << lane) - 1ull));
      uniq[u] = (int64_t)pp[c].y;
      seg_off[u] = (int32_t)i;
    }
  }
}

int run(int64_t B, int32_t S, int64_t R, int64_t pad, const int64_t* ids, int32_t* sorted_pos, int64_t* uniq_rows,
        int32_t* seg_offset, int32_t* n_uniq, int32_t* rank, int32_t* status, void* workspace, size_t ws_bytes,
        hipStream_t st) {
  const Plan p = make_plan(B, S, R);
  REC_REQUIRE(workspace && ws_bytes >= p.total, REC_EWORKSPACE, "workspace %zu < %zu", ws_bytes, p.total);
  char* base = (char*)workspace;
  uint32_t* keyT = (uint32_t*)(base + p.off_keyT);
  uint2* pairs1 = (uint2*)(base + p.off_pairs1);
  uint2* pairs2 = (uint2*)(base + p.off_pairs2);
  int32_t* hist = (int32_t*)(base + p.off_hist);
  int32_t* bvalid = (int32_t*)(base + p.off_bvalid);
  int32_t* cnt = (int32_t*)(base + p.off_cnt);
  const unsigned nkb = (unsigned)((B + kKeyRows - 1) / kKeyRows);
  hipLaunchKernelGGL(keys_kernel, dim3(nkb), dim3(kKeyRows), (size_t)kKeyRows * (S | 1) * 4, st, B, S, R, pad, ids, keyT,
                     rank, status, n_uniq);
  const unsigned grid = (unsigned)(S * p.C);
  const int nb1 = 1 << p.bits1;
  auto lds_scatter = [](int nb) { return (size_t)(2 * nb + kWaves * nb + kWaves + 1) * sizeof(int); };
  auto lds_hist = [](int nb) { return (size_t)(nb + kWaves) * sizeof(int); };
  hipLaunchKernelGGL(hist_kernel<true>, dim3(grid), dim3(kThreads), lds_hist(nb1), st, B, p.C, 0, p.bits1, keyT,
                     (const uint2*)nullptr, hist, bvalid);
  if (p.bits2 == 0) {
    hipLaunchKernelGGL((scatter_kernel<true, true>), dim3(grid), dim3(kThreads), lds_scatter(nb1), st, B, S, p.C, 0,
                       p.bits1, R, keyT, (const uint2*)nullptr, hist, bvalid, pairs2);
  } else {
    const int nb2 = 1 << p.bits2;
    hipLaunchKernelGGL((scatter_kernel<true, false>), dim3(grid), dim3(kThreads), lds_scatter(nb1), st, B, S, p.C, 0,
                       p.bits1, R, keyT, (const uint2*)nullptr, hist, bvalid, pairs1);
    hipLaunchKernelGGL(hist_kernel<false>, dim3(grid), dim3(kThreads), lds_hist(nb2), st, B, p.C, p.bits1, p.bits2,
                       (const uint32_t*)nullptr, pairs1, hist, bvalid);
    hipLaunchKernelGGL((scatter_kernel<false, true>), dim3(grid), dim3(kThreads), lds_scatter(nb2), st, B, S, p.C,
                       p.bits1, p.bits2, R, (const uint32_t*)nullptr, pairs1, hist, bvalid, pairs2);
  }
  const int nbv = S * p.C;
  hipLaunchKernelGGL(heads_count_kernel, dim3(p.nblk_heads), dim3(kHeadThreads), 0, st, nbv, bvalid, pairs2, cnt, n_uniq);
  hipLaunchKernelGGL(heads_emit_kernel, dim3(p.nblk_heads), dim3(kHeadThreads), 0, st, nbv, bvalid, pairs2, cnt,
                     sorted_pos, rank, uniq_rows, seg_offset, n_uniq);
  return check_launch("rec_ids_group_slots");
}

}  // namespace sg
}  // namespace rec
