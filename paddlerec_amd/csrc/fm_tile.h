// DeepFM FM block for NARROW rows (emb_dim not a multiple of 4, <= 12: the reference's own D 9 / D 10 tables,
// models/rank/deepfm/config.yaml:48-50, benchmark.yaml:21) — forward and backward as block-tile kernels.
//
// The row-group kernels of deepfm_fm.hip give every field pow2_ceil(D) lanes of one float: at D 9 a wave instruction
// gathers FOUR rows with 9 of 16 lanes live, a sample's 26 lookups are 7 dependent-issue instructions of one wave, and
// feat / row_grad are written as 36-byte pieces at 36-byte offsets.  Here a block of 256 threads owns a tile of 16
// samples (6-7 blocks per CU):
//   fwd: LANE PER LOOKUP — the 416 lookups of a tile are 1.6 per thread, each thread fetches the whole row of its
//        lookup (three 16-byte loads from one 64-byte record when the table is kept at a multiple-of-4 stride, all of a
//        thread's lookups issued before any is consumed), the tile's feat rows are assembled in LDS, the FM sums run
//        over LDS columns, and feat leaves as float4 runs: the whole tile when feat is dense ([B, F*D]: 16 samples are
//        one contiguous 64-B-aligned run), whole rows when it sits at a padded sample stride.
//   bwd: ONE float4 pass over the contiguous per-sample runs of feat and d_feat: the sparse fields' gradient rows are
//        staged in LDS and leave as one contiguous run per tile, the dense fields' values are parked in LDS and their
//        batch reductions are owned by FIXED threads ((j, d) -> thread) that walk the tile's samples in order —
//        deterministic, then folded by fold_partials_kernel like the wide kernel's.
// Same results as the row-group kernels up to the fp32 summation order of the FM sums (fields ascending here).
#pragma once

#include "rec_common.h"

namespace rec {

constexpr int kFmTileS = 16;       // samples per tile (r05: 32 -> 16 with 2 lookups per thread doubles the blocks per CU)
constexpr int kFmTileMaxD = 12;
constexpr int kFmTileLook = 2;     // lookups a thread holds in flight (kFmTileS * S <= 256 * 2: S <= 32)

struct FmTileLds {                 // float offsets into the dynamic LDS block (host and device compute the same)
  int tile, first, dw, dw1, part, off_bytes, total_bytes;
};
__host__ __device__ inline FmTileLds fm_tile_fwd_lds(int S, int Dn, int D, int pitch) {
  FmTileLds l;
  const int F = S + Dn;
  l.tile = 0;
  l.first = l.tile + kFmTileS * pitch;
  l.dw = l.first + kFmTileS * F;
  l.dw1 = l.dw + Dn * D;
  l.part = l.dw1 + Dn;
  int fl = l.part + kFmTileS * D;
  fl = (fl + 3) & ~3;
  l.off_bytes = fl * 4;                                   // int64 slot offsets behind the floats (16-B aligned)
  l.total_bytes = l.off_bytes + S * 8;
  return l;
}

// feat pitch inside LDS: the dense layout keeps the tile contiguous; a padded sample stride gets whole float4 rows
__host__ __device__ inline int fm_tile_pitch(int F, int D, int64_t feat_ld) {
  return feat_ld == (int64_t)F * D ? F * D : ((F * D + 3) & ~3);
}

template <bool VEC4, bool NT>
__global__ __launch_bounds__(kBlock, 6) void fm_fwd_tile_kernel(
    int64_t B, int S, int Dn, int D, int64_t feat_ld, int stride, int w1_stride, int64_t N, int64_t pad,
    const int64_t* __restrict__ ids, const float* __restrict__ dense, const float* __restrict__ W,
    const float* __restrict__ W1, const float* __restrict__ dense_w, const float* __restrict__ dense_w_one,
    const int64_t* __restrict__ slot_off, float* __restrict__ y1, float* __restrict__ y2, float* __restrict__ feat,
    float* __restrict__ sum_emb, int32_t* __restrict__ status, int zero_to) {
  // zero_to > pitch (padded sample stride only): the columns [pitch, zero_to) of every sample are written as zeros too —
  // a caller whose feat buffer is scratch (rec_deepfm_train_step at launch-bound sizes) then needs no memset launch
  extern __shared__ __attribute__((aligned(16))) unsigned char fm_tile_smem[];
  const int F = S + Dn, FD = F * D;
  const int P = fm_tile_pitch(F, D, feat_ld);
  const FmTileLds L = fm_tile_fwd_lds(S, Dn, D, P);
  float* sm = reinterpret_cast<float*>(fm_tile_smem);
  float* t_tile = sm + L.tile;
  float* t_first = sm + L.first;
  float* s_dw = sm + L.dw;
  float* s_dw1 = sm + L.dw1;
  float* t_part = sm + L.part;
  int64_t* s_off = reinterpret_cast<int64_t*>(fm_tile_smem + L.off_bytes);
  const int tid = threadIdx.x;
  for (int i = tid; i < Dn * D; i += kBlock) s_dw[i] = dense_w[i];
  for (int i = tid; i < Dn; i += kBlock) s_dw1[i] = dense_w_one[i];
  for (int i = tid; i < S; i += kBlock) s_off[i] = slot_off ? slot_off[i] : 0;
  if (P > FD)                                             // the float4 rows' padding columns: zero once, never rewritten
    for (int i = tid; i < kFmTileS * (P - FD); i += kBlock) t_tile[(i / (P - FD)) * P + FD + i % (P - FD)] = 0.f;
  __syncthreads();

  const int64_t ntiles = (B + kFmTileS - 1) / kFmTileS;
  const int nlook = kFmTileS * S;
  int oob = 0;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t b0 = tile * kFmTileS;
    // ---- phase A: one thread per lookup; every row of the thread is requested before the first is used
    float e[kFmTileLook][kFmTileMaxD];
    float one[kFmTileLook];
    int where[kFmTileLook];                                 // LDS offset of the lookup's field, < 0: not a lookup
    bool hit[kFmTileLook];
#pragma unroll
    for (int u = 0; u < kFmTileLook; ++u) {
      const int idx = u * kBlock + tid;
      const int sample = idx / S, slot = idx - sample * S;
      const int64_t b = b0 + sample;
      const bool active = idx < nlook && b < B;
      const int64_t id = active ? ids[b * S + slot] : 0;
      const int64_t r = id + s_off[active ? slot : 0];
      const bool live = active && (id != pad || pad < 0);
      const bool inr = r >= 0 && r < N;
      oob |= (live && !inr) ? 1 : 0;
      hit[u] = live && inr;
      where[u] = active ? sample * P + slot * D : -1;
      const float* row = W + (hit[u] ? r : 0) * (int64_t)stride;
      if (idx < nlook) {                                    // block-uniform per u except in the last round
        if (VEC4) {
#pragma unroll
          for (int c = 0; c < kFmTileMaxD / 4; ++c) {
            if (c * 4 < D) {
              const float4 v = *reinterpret_cast<const float4*>(row + c * 4);
              e[u][c * 4 + 0] = v.x; e[u][c * 4 + 1] = v.y; e[u][c * 4 + 2] = v.z; e[u][c * 4 + 3] = v.w;
            }
          }
        } else {
#pragma unroll
          for (int d = 0; d < kFmTileMaxD; ++d)
            if (d < D) e[u][d] = row[d];
        }
        one[u] = W1[(hit[u] ? r : 0) * (int64_t)w1_stride];
      }
    }
#pragma unroll
    for (int u = 0; u < kFmTileLook; ++u) {
      if (where[u] >= 0) {
        const int idx = u * kBlock + tid;
        const int sample = idx / S, slot = idx - sample * S;
#pragma unroll
        for (int d = 0; d < kFmTileMaxD; ++d)
          if (d < D) t_tile[where[u] + d] = hit[u] ? e[u][d] : 0.f;
        t_first[sample * F + slot] = hit[u] ? one[u] : 0.f;
      }
    }
    // ---- phase A2: the dense "embeddings" x_j * dense_w[j, :] (net.py:110-119)
    for (int idx = tid; idx < kFmTileS * Dn; idx += kBlock) {
      const int sample = idx / Dn, j = idx - sample * Dn;
      const int64_t b = b0 + sample;
      const float x = b < B ? dense[b * Dn + j] : 0.f;
      float* dst = t_tile + sample * P + (S + j) * D;
      for (int d = 0; d < D; ++d) dst[d] = x * s_dw[j * D + d];
      t_first[sample * F + S + j] = x * s_dw1[j];
    }
    __syncthreads();
    // ---- phase B: FM sums over the fields of a sample, per embedding column (net.py:124-136)
    for (int idx = tid; idx < kFmTileS * D; idx += kBlock) {
      const int sample = idx / D, d = idx - sample * D;
      const float* col = t_tile + sample * P + d;
      float s = 0.f, q = 0.f;
      for (int f = 0; f < F; ++f) {
        const float v = col[f * D];
        s += v;
        q += v * v;
      }
      t_part[idx] = s * s - q;
      const int64_t b = b0 + sample;
      if (b < B && sum_emb) sum_emb[b * D + d] = s;
    }
    __syncthreads();
    if (tid < kFmTileS && b0 + tid < B) {
      float t2 = 0.f, t1 = 0.f;
      for (int d = 0; d < D; ++d) t2 += t_part[tid * D + d];
      for (int f = 0; f < F; ++f) t1 += t_first[tid * F + f];
      y1[b0 + tid] = t1;                                    // net.py:113-114
      y2[b0 + tid] = 0.5f * t2;                             // net.py:135
    }
    // ---- phase C: feat leaves as float4 runs
    const int nvalid = (int)((B - b0) < kFmTileS ? (B - b0) : kFmTileS);
    if (feat_ld == (int64_t)FD) {                           // dense layout: the tile is one contiguous run
      float* dst = feat + b0 * FD;
      const int total = nvalid * FD, n4 = total / 4;
      for (int i = tid; i < n4; i += kBlock) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(t_tile + i * 4);
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst + i * 4));
        else *reinterpret_cast<f32x4*>(dst + i * 4) = v;
      }
      for (int i = n4 * 4 + tid; i < total; i += kBlock) dst[i] = t_tile[i];
    } else {                                                // padded sample stride: whole float4 rows
      const int r4 = P / 4;
      for (int i = tid; i < nvalid * r4; i += kBlock) {
        const int sample = i / r4, c4 = i - sample * r4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(t_tile + sample * P + c4 * 4);
        float* dst = feat + (b0 + sample) * feat_ld + c4 * 4;
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst));
        else *reinterpret_cast<f32x4*>(dst) = v;
      }
      if (zero_to > P) {
        const int z4 = (zero_to - P) / 4;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        for (int i = tid; i < nvalid * z4; i += kBlock) {
          const int sample = i / z4, c4 = i - sample * z4;
          *reinterpret_cast<f32x4*>(feat + (b0 + sample) * feat_ld + P + c4 * 4) = zero;
        }
      }
    }
    __syncthreads();                                        // the tile is rewritten by the next round's phase A
  }
  if (oob) atomicOr(status, REC_FLAG_INDEX_OOB);
}

// ------------------------------------------------------------------------------------------------------------ backward
struct FmTileBwdLds {
  int rg, ssum, g1, g2, x, dw, dg, de, total_floats;
};
__host__ __device__ inline FmTileBwdLds fm_tile_bwd_lds(int S, int Dn, int D) {
  FmTileBwdLds l;
  l.rg = 0;
  l.ssum = l.rg + ((kFmTileS * S * D + 3) & ~3);
  l.g1 = l.ssum + kFmTileS * D;
  l.g2 = l.g1 + kFmTileS;
  l.x = l.g2 + kFmTileS;
  l.dw = l.x + kFmTileS * Dn;
  l.dg = l.dw + Dn * D;                 // d_feat of the dense fields [tile][Dn*D]
  l.de = l.dg + kFmTileS * Dn * D;      // feat of the dense fields (read only when dense_w is not given)
  l.total_floats = l.de + kFmTileS * Dn * D;
  return l;
}

template <bool VEC4, bool NT>
__global__ __launch_bounds__(kBlock) void fm_bwd_tile_kernel(
    int64_t B, int S, int Dn, int D, int64_t feat_ld, const float* __restrict__ dense, const float* __restrict__ feat,
    const float* __restrict__ sum_emb, const float* __restrict__ dfeat, const float* __restrict__ dy1,
    const float* __restrict__ dy2, const float* __restrict__ dense_w, float* __restrict__ row_grad,
    float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float fm_tileb_smem[];
  const FmTileBwdLds L = fm_tile_bwd_lds(S, Dn, D);
  float* t_rg = fm_tileb_smem + L.rg;
  float* t_sum = fm_tileb_smem + L.ssum;
  float* t_g1 = fm_tileb_smem + L.g1;
  float* t_g2 = fm_tileb_smem + L.g2;
  float* t_x = fm_tileb_smem + L.x;
  float* s_dw = fm_tileb_smem + L.dw;
  float* t_dg = fm_tileb_smem + L.dg;
  float* t_de = fm_tileb_smem + L.de;
  const int tid = threadIdx.x;
  const int SD = S * D, KD = Dn * D, FD = SD + KD;
  if (dense_w)
    for (int i = tid; i < KD; i += kBlock) s_dw[i] = dense_w[i];
  // this thread's reduction: (j, d) of d_dense_w for tid < Dn*D, j of d_dense_w_one for the next Dn threads
  const bool own_w = tid < KD, own_1 = tid >= KD && tid < KD + Dn;
  const int oj = own_w ? tid / D : (own_1 ? tid - KD : 0), od = own_w ? tid - oj * D : 0;
  float acc = 0.f;
  const int64_t ntiles = (B + kFmTileS - 1) / kFmTileS;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t b0 = tile * kFmTileS;
    const int nvalid = (int)((B - b0) < kFmTileS ? (B - b0) : kFmTileS);
    __syncthreads();                                        // the previous round's readers of the staged values are done
    for (int i = tid; i < nvalid * D; i += kBlock) t_sum[i] = sum_emb[b0 * D + i];
    for (int i = tid; i < nvalid * Dn; i += kBlock) t_x[i] = dense[b0 * Dn + i];
    if (tid < nvalid) {
      t_g1[tid] = dy1[b0 + tid];
      t_g2[tid] = dy2[b0 + tid];
    }
    __syncthreads();
    // ---- one pass over the sample's contiguous F*D run of feat / d_feat: the sparse fields' gradient rows
    //      d row = d_dnn + dy2 * (sum_emb - feat) go to the LDS tile, the dense fields' values are parked in LDS for the
    //      batch reductions below (no thread walks global memory sample by sample)
    if (VEC4) {                                             // feat rows 16-B aligned (feat_ld % 4 == 0)
      const int c4n = (FD + 3) / 4;
      for (int i = tid; i < nvalid * c4n; i += kBlock) {
        const int sample = i / c4n, c0 = (i - sample * c4n) * 4;
        const int64_t p = (b0 + sample) * feat_ld + c0;
        f32x4 ev, gv;
        if (NT) {
          ev = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(feat + p));
          gv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(dfeat + p));
        } else {
          ev = *reinterpret_cast<const f32x4*>(feat + p);
          gv = *reinterpret_cast<const f32x4*>(dfeat + p);
        }
        const float g2 = t_g2[sample];
        const float* sb = t_sum + sample * D;
        int d = c0 % D;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int c = c0 + k;
          if (c < SD) t_rg[sample * SD + c] = gv[k] + g2 * (sb[d] - ev[k]);
          else if (c < FD) {
            t_dg[sample * KD + c - SD] = gv[k];
            t_de[sample * KD + c - SD] = ev[k];
          }
          d = d + 1 == D ? 0 : d + 1;
        }
      }
    } else {
      for (int i = tid; i < nvalid * FD; i += kBlock) {
        const int sample = i / FD, c = i - sample * FD;
        const int64_t p = (b0 + sample) * feat_ld + c;
        const float ev = feat[p], gv = dfeat[p];
        if (c < SD) t_rg[sample * SD + c] = gv + t_g2[sample] * (t_sum[sample * D + c % D] - ev);
        else {
          t_dg[sample * KD + c - SD] = gv;
          t_de[sample * KD + c - SD] = ev;
        }
      }
    }
    __syncthreads();
    // ---- dense fields: batch sums owned by fixed threads, samples in ascending order (deterministic)
    if (own_w) {
      for (int sample = 0; sample < nvalid; ++sample) {
        const float x = t_x[sample * Dn + oj];
        const float ev = dense_w ? x * s_dw[tid] : t_de[sample * KD + tid];
        const float de = t_dg[sample * KD + tid] + t_g2[sample] * (t_sum[sample * D + od] - ev);
        acc += x * de;
      }
    } else if (own_1) {
      for (int sample = 0; sample < nvalid; ++sample) acc += t_g1[sample] * t_x[sample * Dn + oj];
    }
    __syncthreads();
    // ---- the tile's row gradients: one contiguous run [nvalid * S * D] starting 128-B aligned
    {
      float* dst = row_grad + b0 * SD;
      const int total = nvalid * SD, n4 = total / 4;
      for (int i = tid; i < n4; i += kBlock)
        *reinterpret_cast<f32x4*>(dst + i * 4) = *reinterpret_cast<const f32x4*>(t_rg + i * 4);
      for (int i = n4 * 4 + tid; i < total; i += kBlock) dst[i] = t_rg[i];
    }
  }
  if (own_w || own_1) partial[(int64_t)tid * gridDim.x + blockIdx.x] = acc;   // [K][nblk], K = Dn*D + Dn: k == tid
}

}  // namespace rec
