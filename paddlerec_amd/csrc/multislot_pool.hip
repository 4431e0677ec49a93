// Multi-slot embedding lookup + sum-pool in ONE launch (gfx950).
//
// Replaces, for every slot of a batch at once, the per-slot op pair of
//   /root/reference/models/rank/slot_dnn/net.py:63-75
//       for s_input in slot_inputs: emb = sparse_embedding(s_input, padding_idx=0, entry=ShowClickEntry)
//                                   bow = sequence_pool(emb, 'sum');  y = concat(bows, axis=1)
//   /root/reference/models/rank/dnn/static_model_lod.py:70-97   (same pattern, 26 slots)
// (Paddle: 2 ops x 408 slots per forward, each with its own LoD walk and a [nnz_s, D] intermediate).
//
// Input is the slot-major CSR rec_parse_feasign_slots produces (values | lod [S, B+1] | slot_base): the ids of one
// slot for consecutive samples are contiguous, so a wave that owns (slot s, 64 consecutive samples) reads its
// offsets with one coalesced load per lane and its ids in 512-B runs.
//
// Work decomposition (HBM-bound integer/byte work, no MFMA):
//   block = 8 waves = tile of TS = 64 samples x SS = 8*NSW slots; wave w owns NSW slots of the tile.
//   per slot:  offsets -> per-wave LDS; every lane (= sample) stamps its run into a per-wave byte table of
//              segment ids (LDS-staged segment table; binary search in the offsets if a run table would overflow)
//              ids, 64 per step, one per lane (coalesced) -> hash -> row; the LIVE ids (not padding, in range)
//              are compacted through LDS in id order, so padding costs no gather slot
//              rows gathered G = 64/LANES at a time (row group = LANES lanes x float4, lane = slice * G + group),
//              all sub-steps of a 64-id step issued back to back
//              WAVE-LEVEL SEGMENTED REDUCTION: the G row groups of a sub-step hold G consecutive live ids; a
//              segmented scan over the groups (DPP row shifts: no LDS traffic; steps skipped once no run reaches
//              that far) leaves each run's sum in its last group, which adds it to the block's LDS output tile
//              (one lane per address: no conflicts between waves - they own different slots - and program order
//              inside a wave)
//   the LDS tile [TS][SS*D] is written out once, in SS*D*4-byte runs per sample (concat(axis=1) layout), and
//   the pooled-id counts [TS][SS] likewise — a per-slot launch would write D*4 = 36-byte pieces.
// Work is proportional to the number of ids, not to B x S x longest segment: load balance does not depend on
// the segment-length distribution.  Fixed summation order (tree inside a sub-step, ascending across): results
// do not depend on the launch geometry.
#include <stdlib.h>

#include "rec_common.h"

namespace rec {

constexpr int kMsTS = 64;     // samples per tile (one offset per lane)
// waves per block W: 8 (512 threads) for D <= 20 — the LDS output tile is then shared by 8 waves and three resident
// blocks give 24 waves per CU (the kernel is latency-, not ALU-bound); 4 for wider rows (the tile grows with D)

__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// x % d for a divisor fixed per launch: q = mulhi(x, floor(2^64 / d)) is floor(x / d) or one less, so at most two
// conditional subtractions finish it — ~20 instructions instead of the ~150 of a generic 64-bit remainder
// (the hash runs once per id: 40 M times per batch on the slot_dnn shape).
struct FastMod {
  uint64_t d, magic;
};
__device__ __forceinline__ uint64_t fast_mod(uint64_t x, const FastMod& f) {
  const uint64_t q = __umul64hi(x, f.magic);
  uint64_t r = x - q * f.d;
  r = r >= f.d ? r - f.d : r;
  r = r >= f.d ? r - f.d : r;
  return r;
}

// lane shift inside a 16-lane row without touching the LDS crossbar (DPP row_shr); lanes whose source would be
// outside the row keep `old`
template <int O>
__device__ __forceinline__ int row_shr_i(int old, int x) {
  return __builtin_amdgcn_update_dpp(old, x, 0x110 | O, 0xF, 0xF, false);
}
template <int O>
__device__ __forceinline__ float row_shr_f(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x110 | O, 0xF, 0xF, false));
}

constexpr int kSegCap = 512;    // ids of (one slot, 64 samples) whose segment ids fit the per-wave byte table

// Lane layout inside a wave: lane = lg * G + g — the G row groups sit side by side (g), the LANES slices of a row
// are G lanes apart (lg).  With G <= 16 the segmented scan over g never leaves a 16-lane DPP row.
template <int VEC, int LANES, int NSW, int kMsWaves>
__global__ __launch_bounds__(kMsWaves * kWave, (kMsWaves == 8 || NSW >= 2) ? 6 : 3) void multislot_sumpool_kernel(
    int64_t B, int S, int D, int stride, int key_mode, int64_t N, int64_t pad, int64_t lod_stride,
    int64_t out_stride, int nbt, int pitch, int state_off, float init_range, int init_dims, uint64_t seed,
    FastMod fm, const int64_t* __restrict__ values,
    const int64_t* __restrict__ lod, const int64_t* __restrict__ slot_base,
    const float* __restrict__ W, float* __restrict__ out, int32_t* __restrict__ counts,
    int32_t* __restrict__ seg_of_value, int64_t* __restrict__ rows_out,
    int32_t* __restrict__ status) {
  constexpr int kMsBlock = kMsWaves * kWave;
  constexpr int G = kWave / LANES;        // row groups per wave instruction
  constexpr int SS = kMsWaves * NSW;      // slots per tile
  constexpr int U = (kWave / G) < 4 ? (kWave / G) : 4;  // sub-steps gathered back to back (<= a 64-id step)
  constexpr bool DPP = G <= 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* tile = reinterpret_cast<float*>(smem_raw);                 // [TS][pitch]
  int* cnt = reinterpret_cast<int*>(tile + kMsTS * pitch);          // [TS][SS]
  int64_t* crow = reinterpret_cast<int64_t*>(cnt + kMsTS * SS);     // [waves][64]
  int* cseg = reinterpret_cast<int*>(crow + kMsWaves * kWave);      // [waves][64]
  unsigned char* segtab = reinterpret_cast<unsigned char*>(cseg + kMsWaves * kWave);   // [waves][kSegCap]
  const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
  const int64_t b0 = (int64_t)(blockIdx.x % nbt) * kMsTS;
  const int s0 = (int)(blockIdx.x / nbt) * SS;
  const int ns_tile = min(SS, S - s0);

  for (int i = threadIdx.x; i < kMsTS * pitch; i += kMsBlock) tile[i] = 0.f;
  for (int i = threadIdx.x; i < kMsTS * SS; i += kMsBlock) cnt[i] = 0;
  __syncthreads();

  // ---- phase A: the tile's offsets of every slot this wave owns (2 coalesced loads per lane and slot)
  int64_t k0[NSW], base[NSW];
  int n[NSW], len[NSW], so_l[NSW];
  int64_t idv[NSW], idv2[NSW];
  int64_t* wrow = crow + wave * kWave;
  int* wseg = cseg + wave * kWave;
  unsigned char* wtab = segtab + wave * kSegCap;
#pragma unroll
  for (int i = 0; i < NSW; ++i) {
    const int sl = wave * NSW + i;
    n[i] = 0; k0[i] = 0; base[i] = 0; len[i] = 0; so_l[i] = 0;
    if (sl < ns_tile) {
      const int64_t* l = lod + (int64_t)(s0 + sl) * lod_stride;
      const int64_t ba = min(b0 + lane, B), bb = min(b0 + lane + 1, B);
      const int64_t lo = l[ba], hi = l[bb];
      k0[i] = __shfl(lo, 0, kWave);
      const int64_t kend = __shfl(hi, kWave - 1, kWave);
      n[i] = (int)(kend - k0[i]);
      base[i] = slot_base[s0 + sl];
      so_l[i] = (int)(lo - k0[i]);
      len[i] = (int)(hi - lo);
    }
  }
  // ---- phase B: first 64 ids of every slot (in flight together)
#pragma unroll
  for (int i = 0; i < NSW; ++i) {
    idv[i] = lane < n[i] ? values[base[i] + k0[i] + lane] : 0;
    idv2[i] = kWave + lane < n[i] ? values[base[i] + k0[i] + kWave + lane] : 0;
  }
  wave_fence();

  const int lg = lane / G, g = lane % G;
  const int d0 = lg * VEC;
  int oob = 0;
#pragma unroll
  for (int i = 0; i < NSW; ++i) {
    const int sl = wave * NSW + i;
    // segment id of every id of this (slot, 64 samples): each lane (= sample) stamps its own run into the byte
    // table; runs are short (<= 15 in the reference's data), so this is a couple of LDS stores per lane
    const bool tab = n[i] <= kSegCap;
    if (tab) {
      for (int j = 0; __ballot(j < len[i]) != 0; ++j)
        if (j < len[i]) wtab[so_l[i] + j] = (unsigned char)lane;
    }
    wave_fence();
    for (int c0 = 0; c0 < n[i]; c0 += kWave) {
      const int kk = c0 + lane;
      const bool in = kk < n[i];
      const int64_t id = c0 == 0 ? idv[i] : c0 == kWave ? idv2[i] : (in ? values[base[i] + k0[i] + kk] : 0);
      const bool live = in && (id != pad || pad < 0);
      const int64_t r = key_mode ? (id == 0 ? 0 : (int64_t)(1 + fast_mod(mix64((uint64_t)id), fm))) : id;
      const bool inr = r >= 0 && r < N;
      oob |= (live && !inr) ? 1 : 0;
      const bool hit = live && inr;
      int seg = -1;
      if (tab) {
        if (in) seg = wtab[kk];
      } else {   // oversized run table: binary search in the lanes' offsets (the last t with so[t] <= kk)
        int lo = 0, hi = kMsTS;
#pragma unroll
        for (int it = 0; it < 6; ++it) {
          const int mid = (lo + hi) >> 1;
          const bool le = __shfl(so_l[i], mid, kWave) <= kk;
          lo = le ? mid : lo;
          hi = le ? hi : mid;
        }
        seg = in ? lo : -1;
      }
      if (in) {
        const int64_t gk = base[i] + k0[i] + kk;
        if (seg_of_value) seg_of_value[gk] = (int32_t)((b0 + seg) * S + s0 + sl);
        if (rows_out) rows_out[gk] = hit ? r : (pad >= 0 ? (key_mode ? 0 : pad) : r);
      }
      if (hit) atomicAdd(&cnt[seg * SS + sl], 1);   // LDS integer add: exact, order-free
      // compaction: only the live ids are gathered; they stay in id order, so runs stay contiguous
      const unsigned long long hm = __ballot(hit);
      const int nlive = __popcll(hm);
      const int cpos = __popcll(hm & ((1ull << lane) - 1ull));
      if (hit) { wrow[cpos] = r; wseg[cpos] = seg; }
      wave_fence();
      const int nsub = (nlive + G - 1) / G;
      for (int j0 = 0; j0 < nsub; j0 += U) {
        float e[U][VEC], stf[U];
        int sg[U], sgn[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int q = (j0 + u) * G + g;          // compacted id this row group sums
          const bool ok = q < nlive;
          const int64_t rr = ok ? wrow[q] : 0;
          sg[u] = ok ? wseg[q] : -1;
          sgn[u] = (q + 1 < nlive && g != G - 1) ? wseg[q + 1] : -2;   // last group of a sub-step always flushes
#pragma unroll
          for (int v = 0; v < VEC; ++v) e[u][v] = 0.f;
          stf[u] = 1.f;
          if (ok && d0 < D) {
            vload<VEC>(e[u], W + rr * stride + d0);
            // PS rows are born lazily; the state float is only LOADED here (same record line as the weights) and
            // looked at below, so the gathers of all U sub-steps are in flight before the first wait
            if (state_off >= 0) stf[u] = W[rr * stride + state_off];
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if ((j0 + u) < nsub) {                  // wave-uniform
            if (state_off >= 0 && stf[u] == 0.f) {   // an unborn row (state 0, zero memory) reads as its creation values
              const int64_t rr = wrow[(j0 + u) * G + g];   // (stf is 1 where the group holds no row)
#pragma unroll
              for (int v = 0; v < VEC; ++v)
                e[u][v] = d0 + v < init_dims ? ps_init_value(seed, rr, d0 + v, init_range) : 0.f;
            }
            // segmented inclusive scan over the G groups (runs of equal seg are contiguous); a step is skipped
            // once no run reaches that far back (wave-uniform)
#define REC_SCAN_STEP(O)                                                                      \
  if (O < G && more) {                                                                        \
    int so_;                                                                                  \
    float t[VEC];                                                                             \
    if (DPP) {                                                                                \
      so_ = row_shr_i<O>(-3, sg[u]);                                                          \
      _Pragma("unroll") for (int v = 0; v < VEC; ++v) t[v] = row_shr_f<O>(e[u][v]);           \
    } else {                                                                                  \
      so_ = __shfl_up(sg[u], O, kWave);                                                       \
      _Pragma("unroll") for (int v = 0; v < VEC; ++v) t[v] = __shfl_up(e[u][v], O, kWave);    \
    }                                                                                         \
    const bool take = g >= O && so_ == sg[u] && sg[u] >= 0;                                   \
    more = __ballot(take) != 0;                                                               \
    _Pragma("unroll") for (int v = 0; v < VEC; ++v) e[u][v] += take ? t[v] : 0.f;             \
  }
            bool more = true;
            REC_SCAN_STEP(1) REC_SCAN_STEP(2) REC_SCAN_STEP(4) REC_SCAN_STEP(8)
            REC_SCAN_STEP(16) REC_SCAN_STEP(32)
#undef REC_SCAN_STEP
            if (sg[u] >= 0 && sgn[u] != sg[u]) {  // tail of a run: its sum goes to the output tile
              float* dst = tile + sg[u] * pitch + sl * D + d0;
#pragma unroll
              for (int v = 0; v < VEC; ++v)
                if (d0 + v < D) atomicAdd(dst + v, e[u][v]);   // ds_add_f32, one lane per address
            }
          }
        }
      }
      wave_fence();   // the compaction scratch is rewritten by the next 64-id step
    }
  }
  if (oob) atomicOr(status, REC_FLAG_INDEX_OOB);
  __syncthreads();

  // ---- the tile goes out in ns_tile*D-float runs per sample (concat(axis=1) layout); a wave per sample row,
  // lanes along the run (no integer division in the loop)
  const int run = ns_tile * D;
  const int nsamp = (int)min((int64_t)kMsTS, B - b0);
  for (int smp = wave; smp < nsamp; smp += kMsWaves) {
    float* dst = out + (b0 + smp) * out_stride + (int64_t)s0 * D;
    const float* src = tile + smp * pitch;
    for (int c = lane; c < run; c += kWave) dst[c] = src[c];
  }
  if (counts) {
    constexpr int SPW = kWave / SS > 0 ? kWave / SS : 1;   // samples per wave instruction (SS is a power of two <= 64)
    const int c = lane % SS, so = lane / SS;
    for (int smp = wave * SPW + so; smp < nsamp; smp += kMsWaves * SPW)
      if (c < ns_tile && so < SPW) counts[(b0 + smp) * S + s0 + c] = cnt[smp * SS + c];
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Lane-per-id variant for narrow rows (D = 8..11 in 16-byte aligned records, tables below 2^32 rows, no lazy creation).
//
// The kernel above spends its time ISSUING VALU instructions (profiles/r02f_multislot_short_run_pmc.txt: 11.4 VALU
// wave-instructions per id): with LANES lanes per row it gathers 64 / LANES ids per wave instruction, so the list reads,
// the address arithmetic, the segmented scan and the flush run 64 / G times per 64 live ids, and every 64-id step of a
// slot rounds its live ids up to a multiple of G on its own.  Here
//   * a lane owns ONE id and holds the whole row in registers (D floats: ceil(D / 4) wide loads);
//   * phase 1 (per 64 ids: hash -> row, segment id, side outputs, counts) runs for ALL ids of the wave's slots first
//     and appends the live ones to a per-wave list (row as u32, offset in the block's output tile as u16);
//   * phase 2 gathers the list 64 entries at a time, the loads of two gathers in flight before the first is reduced:
//     a segmented scan over the 16-lane DPP rows (runs are contiguous in the list; shifts 1, 2, 4, 8, skipped once no
//     run reaches that far), then the tail of each piece of a run puts its sum into the LDS tile — the FIRST piece of
//     a run with plain stores (the tile is zero; one lane per address), a piece that continues a run from the DPP row
//     or the drain before it with ds_add_f32, one DPP row per instruction (two such pieces can belong to one run),
//     skipped when there is none.  Pieces go in ascending order, in program order: the summation order is a function
//     of the input alone;
//   * the tile goes out as float4 non-temporal stores (the output, the counts and the side outputs are written once and
//     read by a later kernel; the table rows are what should stay in the caches).
// Measured (profiles/r03_multislot_lane.txt): 0.77-0.83 of the row-group kernel's time on the slot_dnn shape; persistent
// blocks with the next tile's offsets / ids requested a tile ahead were built and were SLOWER (the prefetched state
// costs the registers of an occupancy step), as were plain read-add-write flushes and a scalar conflict-free write-out.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kLaneCap = 160;   // list entries per wave

__device__ __forceinline__ int64_t uniform64(int64_t x) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)x);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(x >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

template <int DT, int NSW, int kMsWaves>
__global__ __launch_bounds__(kMsWaves * kWave, 6) void multislot_sumpool_lane_kernel(
    int64_t B, int S, int stride, int key_mode, int64_t N, int64_t pad, int64_t lod_stride, int64_t out_stride, int nsg,
    int vec_out, FastMod fm, const int64_t* __restrict__ values, const int64_t* __restrict__ lod,
    const int64_t* __restrict__ slot_base, const float* __restrict__ W, float* __restrict__ out,
    int32_t* __restrict__ counts, int32_t* __restrict__ seg_of_value, int64_t* __restrict__ rows_out,
    int32_t* __restrict__ status) {
  constexpr int kMsBlock = kMsWaves * kWave;
  constexpr int SS = kMsWaves * NSW;
  constexpr int D = DT;
  constexpr int pitch = (SS * D) | 1;
  constexpr int NV4 = D / 4, REM = D % 4;                 // full float4 loads, leftover floats
  constexpr int NE = REM == 1 ? D : (D + 3) / 4 * 4;      // floats held per lane (a 1-float tail is a dword load)
  constexpr int TILE4 = (kMsTS * pitch + 3) / 4;          // the tile in float4 (16-byte aligned, padded)
  static_assert(kMsTS * pitch < 65536, "tile offsets are 16-bit");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* tile = reinterpret_cast<float*>(smem_raw);                                   // [TS][pitch]
  int* cnt = reinterpret_cast<int*>(tile + 4 * TILE4);                                // [TS][SS]
  unsigned* lrow_all = reinterpret_cast<unsigned*>(cnt + kMsTS * SS);                 // [waves][kLaneCap]
  unsigned short* ltoff_all = reinterpret_cast<unsigned short*>(lrow_all + kMsWaves * kLaneCap);   // [waves][kLaneCap + 32]
  unsigned char* segtab = reinterpret_cast<unsigned char*>(ltoff_all + kMsWaves * (kLaneCap + 32));  // [waves][kSegCap]
  const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
  const int rpos = lane & 15, rowi = lane >> 4;
  unsigned* lrow = lrow_all + wave * kLaneCap;
  unsigned short* ltoff = ltoff_all + wave * (kLaneCap + 32);
  unsigned char* wtab = segtab + wave * kSegCap;
  // tile -> (64-sample piece, slot group): the slot groups of one piece are consecutive blocks
  const int64_t b0 = (int64_t)(blockIdx.x / (unsigned)nsg) * kMsTS;
  const int s0 = (int)(blockIdx.x % (unsigned)nsg) * SS;
  const int ns_tile = min(SS, S - s0);

  for (int i = threadIdx.x; i < TILE4 + kMsTS * SS / 4; i += kMsBlock)     // tile and counts are adjacent
    reinterpret_cast<float4*>(tile)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

  // ---- offsets of the wave's slots, then the first 128 ids of each (all in flight together)
  int64_t gk0[NSW];
  int n[NSW], len[NSW], so_l[NSW];
  int64_t idv[NSW], idv2[NSW];
#pragma unroll
  for (int i = 0; i < NSW; ++i) {
    const int sl = wave * NSW + i;
    n[i] = 0; gk0[i] = 0; len[i] = 0; so_l[i] = 0;
    if (sl < ns_tile) {
      const int64_t* l = lod + (int64_t)(s0 + sl) * lod_stride;
      const int64_t lo = __builtin_nontemporal_load(l + min(b0 + lane, B));
      const int64_t hi = __builtin_nontemporal_load(l + min(b0 + lane + 1, B));
      const int64_t k0 = uniform64(lo);
      const int64_t kend = uniform64(__shfl(hi, kWave - 1, kWave));
      n[i] = (int)(kend - k0);
      gk0[i] = slot_base[s0 + sl] + k0;
      so_l[i] = (int)(lo - k0);
      len[i] = (int)(hi - lo);
    }
  }
#pragma unroll
  for (int i = 0; i < NSW; ++i) {
    const int64_t* vp = values + gk0[i];
    idv[i] = lane < n[i] ? __builtin_nontemporal_load(vp + lane) : 0;
    idv2[i] = kWave + lane < n[i] ? __builtin_nontemporal_load(vp + kWave + lane) : 0;
  }
  __syncthreads();       // the zeroed tile

  // phase 2 pieces: the rows of list entries [j0, j0 + 64) into registers / their segmented sums into the tile
  auto load_rows = [&](int j0, int nl, float (&e)[NE]) {
#pragma unroll
    for (int v = 0; v < NE; ++v) e[v] = 0.f;
    if (j0 + lane < nl) {
      const float* p = W + (int64_t)lrow[j0 + lane] * stride;
#pragma unroll
      for (int c = 0; c < NV4; ++c) {
        const float4 q = *reinterpret_cast<const float4*>(p + 4 * c);
        e[4 * c] = q.x; e[4 * c + 1] = q.y; e[4 * c + 2] = q.z; e[4 * c + 3] = q.w;
      }
      if constexpr (REM == 1) {
        e[4 * NV4] = p[4 * NV4];
      } else if constexpr (REM > 1) {
        const float4 q = *reinterpret_cast<const float4*>(p + 4 * NV4);
        e[4 * NV4] = q.x; e[4 * NV4 + 1] = q.y; e[4 * NV4 + 2] = q.z; e[4 * NV4 + 3] = q.w;
      }
    }
  };
  // tile offset of the last entry of an overflow drain: its run may go on in the next list.  Never reset: a tile offset
  // names one (sample, slot) cell and the ids of a cell are one run, so an entry that matches it IS the continuation
  int carry = -1;
  auto reduce_rows = [&](int j0, int nl, float (&e)[NE]) {
    const int q = j0 + lane;
    const bool ok = q < nl;
    const int t = ok ? (int)ltoff[q] : -1 - lane;        // distinct negatives: never equal to a neighbour
    const int tn = (q + 1 < nl) ? (int)ltoff[q + 1] : -1;
    // a piece that starts a DPP row may continue a run of the row (or the drain) before it
    const int tp = q > 0 ? (int)ltoff[q - 1] : carry;
    const unsigned long long cm = __ballot(ok && rpos == 0 && tp == t);
    int plen = 1;       // length of the piece that ends here (inclusive scan of ones)
#define REC_LSCAN_STEP(O)                                                                   \
  if (more) {                                                                               \
    const int so_ = row_shr_i<O>(-3, t);                                                    \
    const bool tk = rpos >= O && so_ == t;                                                  \
    more = __ballot(tk) != 0;                                                               \
    const int pl = row_shr_i<O>(0, plen);                                                   \
    plen += tk ? pl : 0;                                                                    \
    _Pragma("unroll") for (int v = 0; v < D; ++v) {                                         \
      const float x = row_shr_f<O>(e[v]);                                                   \
      e[v] += tk ? x : 0.f;                                                                 \
    }                                                                                       \
  }
    bool more = true;
    REC_LSCAN_STEP(1) REC_LSCAN_STEP(2) REC_LSCAN_STEP(4) REC_LSCAN_STEP(8)
#undef REC_LSCAN_STEP
    const bool tail = ok && (rpos == 15 || tn != t);
    const bool cont = plen == rpos + 1 && ((cm >> (lane & 48)) & 1ull) != 0;
    float* dst = tile + (t < 0 ? 0 : t);
    if (tail && !cont) {
#pragma unroll
      for (int v = 0; v < D; ++v) dst[v] = e[v];
    }
    if (cm != 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
      for (int ph = 0; ph < 4; ++ph) {
        if (((cm >> (16 * ph)) & 1ull) != 0) {        // wave-uniform
          if (tail && cont && rowi == ph) {
#pragma unroll
            for (int v = 0; v < D; ++v) atomicAdd(dst + v, e[v]);        // ds_add_f32, one lane per address
          }
        }
      }
    }
  };
  auto drain = [&](int nl) {
    for (int j0 = 0; j0 < nl; j0 += 2 * kWave) {
      float ea[NE], eb[NE];
      load_rows(j0, nl, ea);
      const bool two = j0 + kWave < nl;
      if (two) load_rows(j0 + kWave, nl, eb);
      reduce_rows(j0, nl, ea);
      if (two) reduce_rows(j0 + kWave, nl, eb);
    }
  };

  int oob = 0;
  int nl = 0;     // live ids waiting in the list (wave-uniform)
  // ---- phase 1
#pragma unroll
  for (int i = 0; i < NSW; ++i) {
    const int sl = wave * NSW + i;
    const bool tab = n[i] <= kSegCap;
    if (tab) {
      for (int j = 0; __ballot(j < len[i]) != 0; ++j)
        if (j < len[i]) wtab[so_l[i] + j] = (unsigned char)lane;
    }
    wave_fence();
    const int64_t* vp = values + gk0[i];
    int32_t* svp = seg_of_value ? seg_of_value + gk0[i] : nullptr;
    int64_t* rop = rows_out ? rows_out + gk0[i] : nullptr;
    const int segbase = (int)(b0 * S) + s0 + sl;
    for (int c0 = 0; c0 < n[i]; c0 += kWave) {
      if (nl + kWave > kLaneCap) {      // the list could overflow: reduce what it holds (long pieces only)
        wave_fence();
        drain(nl);
        carry = __builtin_amdgcn_readfirstlane((int)ltoff[nl - 1]);
        nl = 0;
        wave_fence();
      }
      const int kk = c0 + lane;
      const bool in = kk < n[i];
      const int64_t id = c0 == 0 ? idv[i] : c0 == kWave ? idv2[i] : (in ? __builtin_nontemporal_load(vp + kk) : 0);
      const bool live = in && (id != pad || pad < 0);
      const int64_t r = key_mode ? (id == 0 ? 0 : (int64_t)(1 + fast_mod(mix64((uint64_t)id), fm))) : id;
      const bool inr = (uint64_t)r < (uint64_t)N;
      oob |= (live && !inr) ? 1 : 0;
      const bool hit = live && inr;
      int seg = 0;
      if (tab) {
        if (in) seg = wtab[kk];
      } else {   // oversized run table: binary search in the lanes' offsets (the last t with so[t] <= kk)
        int lo = 0, hi = kMsTS;
#pragma unroll
        for (int it = 0; it < 6; ++it) {
          const int mid = (lo + hi) >> 1;
          const bool le = __shfl(so_l[i], mid, kWave) <= kk;
          lo = le ? mid : lo;
          hi = le ? hi : mid;
        }
        seg = lo;
      }
      if (in) {
        if (svp) __builtin_nontemporal_store(segbase + seg * S, svp + kk);
        if (rop) __builtin_nontemporal_store(hit ? r : (pad >= 0 ? (key_mode ? 0 : pad) : r), rop + kk);
      }
      if (hit) atomicAdd(&cnt[seg * SS + sl], 1);   // LDS integer add: exact, order-free
      const unsigned long long hm = __ballot(hit);
      const int cpos = nl + __builtin_amdgcn_mbcnt_hi((unsigned)(hm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hm, 0));
      if (hit) { lrow[cpos] = (unsigned)r; ltoff[cpos] = (unsigned short)(seg * pitch + sl * D); }
      nl += __popcll(hm);
    }
    wave_fence();      // wtab is restamped by the next slot
  }
  // ---- phase 2
  drain(nl);
  if (oob) atomicOr(status, REC_FLAG_INDEX_OOB);
  __syncthreads();

  // ---- write-out
  const int nsamp = (int)min((int64_t)kMsTS, B - b0);
  if (vec_out && ns_tile == SS) {       // whole tile, 16-byte aligned output rows: float4 stores
    constexpr int RV = SS * D / 4;      // float4 per sample row (SS * D is a multiple of 4: SS is)
    constexpr int TOT = kMsTS * RV;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int it = 0; it < (TOT + kMsBlock - 1) / kMsBlock; ++it) {
      const int f = it * kMsBlock + threadIdx.x;
      const int smp = f / RV, c4 = f - smp * RV;
      if (f < TOT && smp < nsamp) {
        const float* src = tile + smp * pitch + 4 * c4;
        const f32x4 q = {src[0], src[1], src[2], src[3]};
        __builtin_nontemporal_store(q, reinterpret_cast<f32x4*>(out + (b0 + smp) * out_stride + (int64_t)s0 * D + 4 * c4));
      }
    }
    if (counts && threadIdx.x < kMsTS * SS / 4) {
      const int smp = threadIdx.x / (SS / 4), c4 = threadIdx.x % (SS / 4);
      const int4 q = *reinterpret_cast<const int4*>(cnt + smp * SS + 4 * c4);
      const i32x4 qq = {q.x, q.y, q.z, q.w};
      if (smp < nsamp) __builtin_nontemporal_store(qq, reinterpret_cast<i32x4*>(counts + (b0 + smp) * S + s0 + 4 * c4));
    }
  } else {
    const int run = ns_tile * D;
    for (int smp = wave; smp < nsamp; smp += kMsWaves) {
      float* dst = out + (b0 + smp) * out_stride + (int64_t)s0 * D;
      const float* src = tile + smp * pitch;
      for (int c = lane; c < run; c += kWave) dst[c] = src[c];
    }
    if (counts) {
      for (int i = threadIdx.x; i < kMsTS * SS; i += kMsBlock) {
        const int smp = i / SS, c = i % SS;
        if (c < ns_tile && smp < nsamp) counts[(b0 + smp) * S + s0 + c] = cnt[i];
      }
    }
  }
}

// the same multiply-high modulo as the pool kernel's fused hash (one implementation, one test surface)
__global__ void feasign_rows_kernel(int64_t n, FastMod fm, const int64_t* __restrict__ keys,
                                    int64_t* __restrict__ rows) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const uint64_t k = (uint64_t)keys[i];
    rows[i] = k == 0 ? 0 : (int64_t)(1 + fast_mod(mix64(k), fm));
  }
}

// magic = floor(2^64 / d); d == 1 (a two-row table): 2^64 - 1 makes the estimate x - 1 and the correction step lands on 0
inline FastMod make_fastmod(uint64_t d) {
  FastMod f;
  f.d = d;
  f.magic = d > 1 ? (uint64_t)((((unsigned __int128)1) << 64) / d) : ~0ull;
  return f;
}

// Rows form of the sum-pool's gradient (the SelectedRows VALUE of slot_dnn/net.py:63-75's backward): value k of the CSR
// belongs to segment seg = b * S + s and receives that segment's gradient row d_out[b, s*D:(s+1)*D].  One thread per
// float of row_grad: consecutive threads write consecutive floats, the reads repeat a segment's row (cache hits).
__global__ __launch_bounds__(kBlock) void multislot_sumpool_bwd_kernel(
    int64_t total, int S, int D, int64_t out_stride, const int32_t* __restrict__ seg_of_value,
    const float* __restrict__ d_out, float* __restrict__ row_grad) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= total) return;
  const int64_t k = i / D;
  const int c = (int)(i - k * D);
  const int32_t seg = seg_of_value[k];
  const int64_t b = seg / S;
  const int s = seg - (int32_t)b * S;
  row_grad[i] = d_out[b * out_stride + (int64_t)s * D + c];
}

}  // namespace rec

using namespace rec;

extern "C" int rec_feasign_rows_host(int64_t n, int64_t num_rows, const uint64_t* keys, int64_t* rows) {
  REC_REQUIRE(n >= 0 && num_rows >= 2 && (n == 0 || (keys && rows)), REC_EINVAL, "bad arguments");
  for (int64_t i = 0; i < n; ++i) rows[i] = feasign_row(keys[i], num_rows);
  return REC_OK;
}

extern "C" int rec_feasign_rows(int64_t n, int64_t num_rows, const int64_t* keys, int64_t* rows,
                                void* stream) {
  REC_REQUIRE(n >= 0 && num_rows >= 2 && (n == 0 || (keys && rows)), REC_EINVAL, "bad arguments");
  if (n == 0) return REC_OK;
  const int64_t grid = (n + kBlock - 1) / kBlock;
  REC_REQUIRE(grid < (1ll << 31), REC_ESHAPE, "n too large");
  hipLaunchKernelGGL(feasign_rows_kernel, dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream, n,
                     make_fastmod((uint64_t)(num_rows - 1)), keys, rows);
  return check_launch("rec_feasign_rows");
}

extern "C" int rec_multislot_sumpool_fwd(const rec_multislot_desc* d, const int64_t* values,
                                         const int64_t* lod, const int64_t* slot_base, const float* W,
                                         float* out, int32_t* counts, int32_t* seg_of_value,
                                         int64_t* rows_out, int32_t* status, void* stream) {
  REC_REQUIRE(d, REC_EINVAL, "desc is NULL");
  REC_REQUIRE(d->batch >= 0 && d->num_slots > 0 && d->emb_dim > 0 && d->row_stride >= d->emb_dim &&
                  d->num_rows > 0,
              REC_EINVAL, "bad sizes B=%lld S=%d D=%d stride=%d", (long long)d->batch, d->num_slots,
              d->emb_dim, d->row_stride);
  REC_REQUIRE(d->key_mode == 0 || (d->key_mode == 1 && d->num_rows >= 2), REC_EINVAL,
              "key_mode must be 0 (rows) or 1 (uint64 feasigns, num_rows >= 2)");
  if (d->batch == 0) return REC_OK;
  const int S = d->num_slots, D = d->emb_dim;
  const int64_t lod_stride = d->lod_stride > 0 ? d->lod_stride : d->batch + 1;
  const int64_t out_stride = d->out_stride > 0 ? d->out_stride : (int64_t)S * D;
  REC_REQUIRE(lod_stride >= d->batch + 1 && out_stride >= (int64_t)S * D, REC_EINVAL,
              "lod_stride / out_stride too small");
  REC_REQUIRE(values && lod && slot_base && W && out && status, REC_EINVAL, "null pointer argument");
  REC_REQUIRE(D <= 64, REC_ESHAPE, "emb_dim %d > 64 unsupported by the multi-slot pool", D);
  // float4 row loads: 16-B aligned rows that are padded to a multiple of 4 floats (D itself may be odd:
  // D = 9 in a 16-float record reads 3 x float4 and masks the tail)
  const int dp = (D + 3) / 4 * 4;
  const bool v4 = d->row_stride % 4 == 0 && d->row_stride >= dp && ((uintptr_t)W) % 16 == 0;
  const int lanes = pow2_ceil(v4 ? dp / 4 : D);
  const int nbt = (int)((d->batch + kMsTS - 1) / kMsTS);
  const int state_off = d->init_range > 0.f ? d->state_offset : -1;
  REC_REQUIRE(d->init_range <= 0.f || (d->state_offset >= D && d->state_offset < d->row_stride &&
                                        d->init_dims >= 0 && d->init_dims <= D),
              REC_EINVAL, "lazy creation needs the state float inside the row, behind the weights");
  FastMod fmod = make_fastmod(1);
  if (d->key_mode == 1) fmod = make_fastmod((uint64_t)(d->num_rows - 1));
  hipStream_t st = (hipStream_t)stream;
  // narrow rows in 16-byte aligned records of a table below 2^32 rows: the lane-per-id kernel (REC_MS_LANE=0: off)
  const char* lane_env = getenv("REC_MS_LANE");            // read per call: tests compare both kernels in one process
  const bool lane_on = !(lane_env && *lane_env == '0');
  if (lane_on && v4 && state_off < 0 && d->num_rows < (1ll << 32) && S > 8 && D >= 8 && D <= 11) {
#define REC_MSL_LAUNCH(DT_)                                                                              \
  case DT_: {                                                                                            \
    constexpr int W_ = 8, NSW_ = 2, SS = W_ * NSW_;                                                      \
    constexpr int pitch = (SS * DT_) | 1;                                                                \
    constexpr size_t shmem = ((size_t)((kMsTS * pitch + 3) / 4 * 4) + (size_t)kMsTS * SS) * 4 +          \
                             (size_t)W_ * kLaneCap * 4 + (size_t)W_ * (kLaneCap + 32) * 2 +              \
                             (size_t)W_ * kSegCap;                                                       \
    static_assert(shmem <= 64 * 1024, "LDS of the lane-per-id kernel");                                  \
    const int nsg = (S + SS - 1) / SS;                                                                   \
    const int64_t grid = (int64_t)nbt * nsg;                                                             \
    REC_REQUIRE(grid < (1ll << 31), REC_ESHAPE, "too many tiles");                                       \
    const int vec_out = out_stride % 4 == 0 && ((uintptr_t)out) % 16 == 0 && S % 4 == 0 &&               \
                        (!counts || ((uintptr_t)counts) % 16 == 0);                                      \
    hipLaunchKernelGGL((multislot_sumpool_lane_kernel<DT_, NSW_, W_>), dim3((unsigned)grid),             \
                       dim3(W_ * kWave), shmem, st, d->batch, S, d->row_stride, d->key_mode,             \
                       d->num_rows, d->padding_idx, lod_stride, out_stride, nsg, vec_out, fmod, values,  \
                       lod, slot_base, W, out, counts, seg_of_value, rows_out, status);                  \
    return check_launch("rec_multislot_sumpool_fwd (lane)");                                             \
  }
    switch (D) {
      REC_MSL_LAUNCH(8) REC_MSL_LAUNCH(9) REC_MSL_LAUNCH(10) REC_MSL_LAUNCH(11)
      default: break;
    }
#undef REC_MSL_LAUNCH
  }
  static const int ms_cfg = [] { const char* v = getenv("REC_MS_CFG"); return v && *v ? atoi(v) : 0; }();
#define REC_MS_LAUNCH(V, L, NSW_, W_)                                                                   \
  {                                                                                                     \
    constexpr int kMsWaves = W_;                                                                        \
    constexpr int kMsBlock = W_ * kWave;                                                                \
    constexpr int SS = kMsWaves * NSW_;                                                                 \
    const int pitch = (SS * D) | 1;                                                                     \
    const size_t shmem = ((size_t)kMsTS * pitch + (size_t)kMsTS * SS) * 4 +                             \
                         (size_t)kMsWaves * kWave * 12 + (size_t)kMsWaves * kSegCap;                    \
    const int64_t grid = (int64_t)nbt * ((S + SS - 1) / SS);                                            \
    REC_REQUIRE(grid < (1ll << 31), REC_ESHAPE, "too many tiles");                                      \
    REC_REQUIRE(shmem <= 160 * 1024, REC_ESHAPE, "LDS tile %zu B too large", shmem);                    \
    auto kern = multislot_sumpool_kernel<V, L, NSW_, W_>;                                               \
    if (shmem > 64 * 1024)                                                                              \
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kMsBlock),                                      \
                       shmem, st, d->batch, S, D, d->row_stride, d->key_mode, d->num_rows,              \
                       d->padding_idx, lod_stride, out_stride, nbt, pitch, state_off, d->init_range,    \
                       d->init_dims, d->init_seed, fmod, values, lod, slot_base, W,                     \
                       out, counts, seg_of_value, rows_out, status);                                    \
    return check_launch("rec_multislot_sumpool_fwd");                                                   \
  }
#define REC_MS_NSW(V, L)                                                                                \
  {                                                                                                     \
    if (D <= 10 && S > 8 && ms_cfg == 1) REC_MS_LAUNCH(V, L, 2, 4)                                      \
    else if (D <= 10 && S > 8 && ms_cfg == 2) REC_MS_LAUNCH(V, L, 4, 4)                                 \
    else if (D <= 10 && S > 8 && ms_cfg == 3) REC_MS_LAUNCH(V, L, 1, 8)                                 \
    else if (D <= 10 && S > 8 && ms_cfg == 4) REC_MS_LAUNCH(V, L, 4, 8)                                 \
    else if (D <= 10 && S > 8) REC_MS_LAUNCH(V, L, 2, 8)                                                \
    else if (D <= 20 && S > 4) REC_MS_LAUNCH(V, L, 1, 8)                                                \
    else REC_MS_LAUNCH(V, L, 1, 4)                                                                      \
  }
  if (v4) {
    if (lanes == 1) REC_MS_NSW(4, 1)
    if (lanes == 2) REC_MS_NSW(4, 2)
    if (lanes == 4) REC_MS_NSW(4, 4)
    if (lanes == 8) REC_MS_LAUNCH(4, 8, 1, 4)
    if (lanes == 16) REC_MS_LAUNCH(4, 16, 1, 4)
  } else {
    if (lanes == 1) REC_MS_NSW(1, 1)
    if (lanes == 2) REC_MS_NSW(1, 2)
    if (lanes == 4) REC_MS_NSW(1, 4)
    if (lanes == 8) REC_MS_NSW(1, 8)
    if (lanes == 16) REC_MS_NSW(1, 16)
    if (lanes == 32) REC_MS_LAUNCH(1, 32, 1, 4)
    if (lanes == 64) REC_MS_LAUNCH(1, 64, 1, 4)
  }
#undef REC_MS_NSW
#undef REC_MS_LAUNCH
  set_error("emb_dim %d unsupported", D);
  return REC_ESHAPE;
}

extern "C" int rec_multislot_sumpool_bwd(const rec_multislot_desc* d, int64_t nnz, const int32_t* seg_of_value,
                                         const float* d_out, float* row_grad, void* stream) {
  REC_REQUIRE(d, REC_EINVAL, "desc is NULL");
  REC_REQUIRE(d->batch >= 0 && d->num_slots > 0 && d->emb_dim > 0 && nnz >= 0, REC_EINVAL, "bad sizes");
  if (nnz == 0 || d->batch == 0) return REC_OK;
  REC_REQUIRE(seg_of_value && d_out && row_grad, REC_EINVAL, "null pointer argument");
  const int64_t out_stride = d->out_stride > 0 ? d->out_stride : (int64_t)d->num_slots * d->emb_dim;
  REC_REQUIRE(out_stride >= (int64_t)d->num_slots * d->emb_dim, REC_EINVAL, "out_stride too small");
  const int64_t total = nnz * d->emb_dim;
  const int64_t grid = (total + kBlock - 1) / kBlock;
  REC_REQUIRE(grid < (1ll << 31), REC_ESHAPE, "nnz too large");
  hipLaunchKernelGGL(multislot_sumpool_bwd_kernel, dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream, total,
                     d->num_slots, d->emb_dim, out_stride, seg_of_value, d_out, row_grad);
  return check_launch("rec_multislot_sumpool_bwd");
}
