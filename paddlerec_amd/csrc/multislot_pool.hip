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
//   block = 4 waves = tile of TS = 64 samples x SS = 4*NSW slots; wave w owns NSW slots of the tile.
//   per slot:  offsets -> per-wave LDS (the segment table of the tile)
//              ids, 64 per step, one per lane (coalesced); lane -> segment by binary search in the LDS table
//              rows gathered G = 64/LANES at a time (row group = LANES lanes x float4), all sub-steps of a
//              64-id step issued back to back (16 x LANES/4 rows in flight per wave)
//              WAVE-LEVEL SEGMENTED REDUCTION: the G row groups of a sub-step hold G consecutive ids; a
//              log2(G)-step segmented scan with shuffles (ids of one sample are adjacent) leaves each run's sum
//              in its last group, which adds it to the block's LDS output tile (one lane per address: no
//              conflicts between waves - they own different slots - and program order inside a wave)
//   the LDS tile [TS][SS*D] is written out once, in SS*D*4-byte runs per sample (concat(axis=1) layout), and
//   the pooled-id counts [TS][SS] likewise — a per-slot launch would write D*4 = 36-byte pieces.
// Work is proportional to the number of ids, not to B x S x longest segment: load balance does not depend on
// the segment-length distribution.  Fixed summation order (tree inside a sub-step, ascending across): results
// do not depend on the launch geometry.
#include "rec_common.h"

namespace rec {

constexpr int kMsTS = 64;     // samples per tile (one offset per lane)
constexpr int kMsWaves = kBlock / kWave;

__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int VEC, int LANES, int NSW>
__global__ __launch_bounds__(kBlock) void multislot_sumpool_kernel(
    int64_t B, int S, int D, int stride, int key_mode, int64_t N, int64_t pad, int64_t lod_stride,
    int64_t out_stride, int nbt, int pitch, int state_off, float init_range, int init_dims, uint64_t seed,
    const int64_t* __restrict__ values,
    const int64_t* __restrict__ lod, const int64_t* __restrict__ slot_base,
    const float* __restrict__ W, float* __restrict__ out, int32_t* __restrict__ counts,
    int32_t* __restrict__ seg_of_value, int64_t* __restrict__ rows_out,
    int32_t* __restrict__ status) {
  constexpr int G = kWave / LANES;        // row groups per wave instruction
  constexpr int SS = kMsWaves * NSW;      // slots per tile
  constexpr int U = (kWave / G) < 4 ? (kWave / G) : 4;  // sub-steps gathered back to back (<= a 64-id step)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* tile = reinterpret_cast<float*>(smem_raw);                 // [TS][pitch]
  int* cnt = reinterpret_cast<int*>(tile + kMsTS * pitch);          // [TS][SS]
  int* offs = cnt + kMsTS * SS;                                     // [waves][NSW][TS+1]
  const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
  const int64_t b0 = (int64_t)(blockIdx.x % nbt) * kMsTS;
  const int s0 = (int)(blockIdx.x / nbt) * SS;
  const int ns_tile = min(SS, S - s0);

  for (int i = threadIdx.x; i < kMsTS * pitch; i += kBlock) tile[i] = 0.f;
  for (int i = threadIdx.x; i < kMsTS * SS; i += kBlock) cnt[i] = 0;
  __syncthreads();

  // ---- phase A: the tile's offsets of every slot this wave owns (2 coalesced loads per lane and slot)
  int64_t k0[NSW], base[NSW];
  int n[NSW];
  int64_t idv[NSW];
  int* woffs = offs + wave * NSW * (kMsTS + 1);
#pragma unroll
  for (int i = 0; i < NSW; ++i) {
    const int sl = wave * NSW + i;
    n[i] = 0; k0[i] = 0; base[i] = 0;
    if (sl < ns_tile) {
      const int64_t* l = lod + (int64_t)(s0 + sl) * lod_stride;
      const int64_t ba = min(b0 + lane, B), bb = min(b0 + lane + 1, B);
      const int64_t lo = l[ba], hi = l[bb];
      k0[i] = __shfl(lo, 0, kWave);
      const int64_t kend = __shfl(hi, kWave - 1, kWave);
      n[i] = (int)(kend - k0[i]);
      base[i] = slot_base[s0 + sl];
      woffs[i * (kMsTS + 1) + lane] = (int)(lo - k0[i]);
      if (lane == kWave - 1) woffs[i * (kMsTS + 1) + kMsTS] = n[i];
    }
  }
  // ---- phase B: first 64 ids of every slot (in flight together)
#pragma unroll
  for (int i = 0; i < NSW; ++i) idv[i] = lane < n[i] ? values[base[i] + k0[i] + lane] : 0;
  wave_fence();

  const int lg = lane % LANES, g = lane / LANES;
  const int d0 = lg * VEC;
  int oob = 0;
#pragma unroll
  for (int i = 0; i < NSW; ++i) {
    const int sl = wave * NSW + i;
    const int* so = woffs + i * (kMsTS + 1);
    for (int c0 = 0; c0 < n[i]; c0 += kWave) {
      const int kk = c0 + lane;
      const bool in = kk < n[i];
      const int64_t id = c0 == 0 ? idv[i] : (in ? values[base[i] + k0[i] + kk] : 0);
      const bool live = in && (id != pad || pad < 0);
      const int64_t r = key_mode ? feasign_row((uint64_t)id, N) : id;
      const bool inr = r >= 0 && r < N;
      oob |= (live && !inr) ? 1 : 0;
      const bool hit = live && inr;
      // segment (sample of the tile) of id kk: the last t with so[t] <= kk
      int lo = 0, hi = kMsTS;   // invariant so[lo] <= kk < so[hi] (so[0] = 0, so[TS] = n)
#pragma unroll
      for (int it = 0; it < 6; ++it) {
        const int mid = (lo + hi) >> 1;
        const bool le = so[mid] <= kk;
        lo = le ? mid : lo;
        hi = le ? hi : mid;
      }
      const int seg = in ? lo : -1;
      if (in) {
        const int64_t gk = base[i] + k0[i] + kk;
        if (seg_of_value) seg_of_value[gk] = (int32_t)((b0 + seg) * S + s0 + sl);
        if (rows_out) rows_out[gk] = hit ? r : (pad >= 0 ? (key_mode ? 0 : pad) : r);
      }
      if (hit) atomicAdd(&cnt[seg * SS + sl], 1);   // LDS integer add: exact, order-free
      const int64_t row = hit ? r : 0;
      const int nsub = (min(kWave, n[i] - c0) + G - 1) / G;
      for (int j0 = 0; j0 < nsub; j0 += U) {
        float e[U][VEC];
        int sg[U], sgn[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int src = (j0 + u) * G + g;       // lane that holds this group's id
          const int64_t rr = __shfl(row, src & 63, kWave);
          const int hh = __shfl((int)hit, src & 63, kWave);
          sg[u] = __shfl(seg, src & 63, kWave);
          sgn[u] = __shfl(seg, (src + 1) & 63, kWave);
          const bool ok = (j0 + u) < nsub && hh && d0 < D;
          if (g == G - 1) sgn[u] = -2;            // last group of the sub-step always flushes
          if ((j0 + u) >= nsub) sg[u] = -1;
#pragma unroll
          for (int v = 0; v < VEC; ++v) e[u][v] = 0.f;
          if (ok) {
            vload<VEC>(e[u], W + rr * stride + d0);
            // PS rows are born lazily: an unborn row (state 0, zero memory) reads as its creation values
            if (state_off >= 0 && W[rr * stride + state_off] == 0.f) {
#pragma unroll
              for (int v = 0; v < VEC; ++v)
                e[u][v] = d0 + v < init_dims ? ps_init_value(seed, rr, d0 + v, init_range) : 0.f;
            }
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if ((j0 + u) < nsub) {                  // wave-uniform
            // segmented inclusive scan over the G groups (runs of equal seg are contiguous)
#pragma unroll
            for (int o = 1; o < G; o <<= 1) {
              const int so_ = __shfl_up(sg[u], o * LANES, kWave);
              const bool take = g >= o && so_ == sg[u];
#pragma unroll
              for (int v = 0; v < VEC; ++v) {
                const float t = __shfl_up(e[u][v], o * LANES, kWave);
                e[u][v] += take ? t : 0.f;
              }
            }
            if (sg[u] >= 0 && sgn[u] != sg[u]) {  // tail of a run: its sum goes to the output tile
              float* dst = tile + sg[u] * pitch + sl * D + d0;
#pragma unroll
              for (int v = 0; v < VEC; ++v)
                if (d0 + v < D) atomicAdd(dst + v, e[u][v]);   // ds_add_f32, one lane per address
            }
          }
        }
      }
    }
  }
  if (oob) atomicOr(status, REC_FLAG_INDEX_OOB);
  __syncthreads();

  // ---- the tile goes out in ns_tile*D-float runs per sample (concat(axis=1) layout)
  const int run = ns_tile * D;
  const int nsamp = (int)min((int64_t)kMsTS, B - b0);
  for (int i = threadIdx.x; i < nsamp * run; i += kBlock) {
    const int smp = i / run, c = i - smp * run;
    out[(b0 + smp) * out_stride + (int64_t)s0 * D + c] = tile[smp * pitch + c];
  }
  if (counts) {
    for (int i = threadIdx.x; i < nsamp * ns_tile; i += kBlock) {
      const int smp = i / ns_tile, c = i - smp * ns_tile;
      counts[(b0 + smp) * S + s0 + c] = cnt[smp * SS + c];
    }
  }
}

__global__ void feasign_rows_kernel(int64_t n, int64_t N, const int64_t* __restrict__ keys,
                                    int64_t* __restrict__ rows) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) rows[i] = feasign_row((uint64_t)keys[i], N);
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
                     num_rows, keys, rows);
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
  hipStream_t st = (hipStream_t)stream;
#define REC_MS_LAUNCH(V, L, NSW_)                                                                       \
  {                                                                                                     \
    constexpr int SS = kMsWaves * NSW_;                                                                 \
    const int pitch = (SS * D) | 1;                                                                     \
    const size_t shmem = ((size_t)kMsTS * pitch + (size_t)kMsTS * SS +                                  \
                          (size_t)kMsWaves * NSW_ * (kMsTS + 1)) * 4;                                   \
    const int64_t grid = (int64_t)nbt * ((S + SS - 1) / SS);                                            \
    REC_REQUIRE(grid < (1ll << 31), REC_ESHAPE, "too many tiles");                                      \
    hipLaunchKernelGGL((multislot_sumpool_kernel<V, L, NSW_>), dim3((unsigned)grid), dim3(kBlock),      \
                       shmem, st, d->batch, S, D, d->row_stride, d->key_mode, d->num_rows,              \
                       d->padding_idx, lod_stride, out_stride, nbt, pitch, state_off, d->init_range,    \
                       d->init_dims, d->init_seed, values, lod, slot_base, W,                           \
                       out, counts, seg_of_value, rows_out, status);                                    \
    return check_launch("rec_multislot_sumpool_fwd");                                                   \
  }
#define REC_MS_NSW(V, L)                                                                                \
  {                                                                                                     \
    if (D <= 10 && S > 8) REC_MS_LAUNCH(V, L, 4)                                                        \
    else if (D <= 20 && S > 4) REC_MS_LAUNCH(V, L, 2)                                                   \
    else REC_MS_LAUNCH(V, L, 1)                                                                         \
  }
  if (v4) {
    if (lanes == 1) REC_MS_NSW(4, 1)
    if (lanes == 2) REC_MS_NSW(4, 2)
    if (lanes == 4) REC_MS_NSW(4, 4)
    if (lanes == 8) REC_MS_LAUNCH(4, 8, 1)
    if (lanes == 16) REC_MS_LAUNCH(4, 16, 1)
  } else {
    if (lanes == 1) REC_MS_NSW(1, 1)
    if (lanes == 2) REC_MS_NSW(1, 2)
    if (lanes == 4) REC_MS_NSW(1, 4)
    if (lanes == 8) REC_MS_NSW(1, 8)
    if (lanes == 16) REC_MS_NSW(1, 16)
    if (lanes == 32) REC_MS_LAUNCH(1, 32, 1)
    if (lanes == 64) REC_MS_LAUNCH(1, 64, 1)
  }
#undef REC_MS_NSW
#undef REC_MS_LAUNCH
  set_error("emb_dim %d unsupported", D);
  return REC_ESHAPE;
}
