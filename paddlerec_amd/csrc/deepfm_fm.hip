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
// log2(FS) xor-shuffles at the end.
//
// Both kernels are persistent (grid <= 8 blocks per CU, waves stride over tiles of SPW samples) and
// free of data-dependent branches in the load path, so every gather of a tile is issued back to back:
//   fwd: the ids / dense values of the NEXT tile are fetched (coalesced, one load per lane) while the
//        current tile's rows are gathered; they reach the lanes that need them through a per-wave LDS
//        staging area.  Invalid lookups (padding, out of range, dense slots) read row 0 and are
//        zeroed with selects; the out-of-range flag is raised once per wave at the end.
//        (A software-pipelined variant holding two tiles in registers was measured and dropped: 128 VGPRs with
//        spills, 78.8 us vs 72.4 us for this kernel on the Criteo shape.)
//   bwd: the dense-field index of a lane is fixed across samples, so the batch reductions
//        d_dense_w / d_dense_w_one accumulate in registers over the persistent loop and are folded
//        in a fixed order (deterministic); the dense part of feat is recomputed as x * dense_w
//        instead of re-read when dense_w is supplied.
#include <stdlib.h>

#include "rec_common.h"
#include "tail_roles.h"
#include "fm_tile.h"

namespace rec {

constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kMaxBlocks = kNumCU * 8;

template <int LANES>
constexpr int fs_for() {  // fields per sample per wave instruction
  return (kWave / LANES) < 8 ? (kWave / LANES) : 8;
}

// acquire/release at wavefront scope: orders a wave's own LDS writes and reads for the compiler
// (the LDS unit executes one wave's instructions in order); no cross-wave traffic goes through it.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------------------------------ fwd
constexpr int kFwdUnroll = 4;
constexpr int kDnMax = 16;
constexpr int kDenseCh = 2;  // ceil(SPW_max * kDnMax / 64) = 8*16/64

template <int VEC, int LANES, int IDCH, bool NT>
__global__ __launch_bounds__(kBlock) void fm_fwd_kernel(
    int64_t B, int S, int Dn, int D, int FP, int64_t feat_ld, int stride, int w1_stride, int64_t N, int64_t pad,
    const int64_t* __restrict__ ids, const float* __restrict__ dense, const float* __restrict__ W,
    const float* __restrict__ W1, const float* __restrict__ dense_w,
    const float* __restrict__ dense_w_one, const int64_t* __restrict__ slot_off,
    float* __restrict__ y1, float* __restrict__ y2, float* __restrict__ feat,
    float* __restrict__ sum_emb, int32_t* __restrict__ status, int main_blocks, FoldFwd fold) {
  // blocks behind the lookup's own: layer 0's weight fold of a launch-bound step (tail_roles.h, FoldFwd)
  if ((int)blockIdx.x >= main_blocks) {
    dense_fold_fwd_role((int)blockIdx.x - main_blocks, threadIdx.x, fold);
    return;
  }
  constexpr int FS = fs_for<LANES>();
  constexpr int SPW = kWave / (LANES * FS);  // samples per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // [Dn*D] dense_w | [Dn] dense_w_one | [S] slot offsets | per wave: [SPW*S] ids, [SPW*Dn] dense
  float* s_dw = reinterpret_cast<float*>(smem_raw);
  float* s_dw1 = s_dw + Dn * D;
  const int hdr = ((Dn * D + Dn) * 4 + 15) & ~15;
  int64_t* s_off = reinterpret_cast<int64_t*>(smem_raw + hdr);
  const int ids_per_tile = SPW * S, dense_per_tile = SPW * Dn;
  const int wave_bytes = (ids_per_tile * 8 + dense_per_tile * 4 + 15) & ~15;
  const int lane = threadIdx.x % kWave;
  const int wave = threadIdx.x / kWave;
  unsigned char* wbase = smem_raw + hdr + S * 8 + wave * wave_bytes;
  int64_t* w_ids = reinterpret_cast<int64_t*>(wbase);
  float* w_dense = reinterpret_cast<float*>(wbase + ids_per_tile * 8);

  for (int i = threadIdx.x; i < Dn * D; i += kBlock) s_dw[i] = dense_w[i];
  for (int i = threadIdx.x; i < Dn; i += kBlock) s_dw1[i] = dense_w_one[i];
  for (int i = threadIdx.x; i < S; i += kBlock) s_off[i] = slot_off ? slot_off[i] : 0;
  __syncthreads();

  const int lg = lane % LANES;
  const int fs = (lane / LANES) % FS;
  const int sp = lane / (LANES * FS);
  const int d0 = lg * VEC;
  const int F = S + Dn;
  const int NIT = (F + FS - 1) / FS;
  const int64_t ntiles = (B + SPW - 1) / SPW;
  const int64_t tstride = (int64_t)main_blocks * kWavesPerBlock;
  const int64_t n_ids = B * S, n_dense = B * Dn;
  int oob = 0;

  int64_t tile = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (tile >= ntiles) return;

  // prefetch registers: the wave's slice of ids / dense for the tile it will work on next
  int64_t pre_ids[IDCH];
  float pre_dense[kDenseCh];
  auto prefetch = [&](int64_t t) {
#pragma unroll
    for (int c = 0; c < IDCH; ++c) {
      int64_t gi = t * ids_per_tile + c * kWave + lane;
      gi = gi < n_ids ? gi : n_ids - 1;  // clamped: value unused where out of range
      pre_ids[c] = ids[gi];
    }
    if (Dn > 0) {
#pragma unroll
      for (int c = 0; c < kDenseCh; ++c) {
        int64_t gi = t * dense_per_tile + c * kWave + lane;
        gi = gi < n_dense ? gi : n_dense - 1;
        pre_dense[c] = dense[gi];
      }
    }
  };
  prefetch(tile);

  for (; tile < ntiles; tile += tstride) {
    // stage this tile's ids / dense values for the lanes that consume them
#pragma unroll
    for (int c = 0; c < IDCH; ++c) {
      const int i = c * kWave + lane;
      if (i < ids_per_tile) w_ids[i] = pre_ids[c];
    }
    if (Dn > 0) {
#pragma unroll
      for (int c = 0; c < kDenseCh; ++c) {
        const int i = c * kWave + lane;
        if (i < dense_per_tile) w_dense[i] = pre_dense[c];
      }
    }
    wave_lds_fence();
    {  // next tile's ids are in flight while this tile gathers (clamped: the last one re-reads)
      const int64_t nxt = tile + tstride;
      prefetch(nxt < ntiles ? nxt : ntiles - 1);
    }

    const int64_t b = tile * SPW + sp;
    const bool active = b < B;
    const bool dvalid = active && d0 < D;
    float s[VEC], q[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) s[v] = q[v] = 0.f;
    float first = 0.f;
    float* fb = feat + b * feat_ld + d0;                         // feat_ld >= FP * D floats between samples
    const bool compact = FP != F;   // feat = S embedding rows + ONE row of raw dense values (see header)

    for (int it0 = 0; it0 < NIT; it0 += kFwdUnroll) {
      int f[kFwdUnroll];
      int64_t row[kFwdUnroll];
      bool hit[kFwdUnroll];
#pragma unroll
      for (int u = 0; u < kFwdUnroll; ++u) {
        f[u] = (it0 + u) * FS + fs;
        const bool sparse = f[u] < S;
        const int fi = sparse ? f[u] : 0;
        const int64_t id = w_ids[sp * S + fi];
        const int64_t r = id + s_off[fi];
        const bool live = sparse && active && (id != pad || pad < 0);
        const bool inr = r >= 0 && r < N;
        oob |= (live && !inr) ? 1 : 0;
        hit[u] = live && inr;
        row[u] = hit[u] ? r : 0;
      }
      float e[kFwdUnroll][VEC];
      float one[kFwdUnroll];
#pragma unroll
      for (int u = 0; u < kFwdUnroll; ++u) {
        // wave-uniform skip of iterations that hold no sparse field at all
        if ((it0 + u) * FS < S) {
          if (d0 < D) vload<VEC>(e[u], W + row[u] * stride + d0);    // net.py:117
          else {
#pragma unroll
            for (int v = 0; v < VEC; ++v) e[u][v] = 0.f;
          }
          one[u] = W1[row[u] * w1_stride];                                     // net.py:108
        } else {
#pragma unroll
          for (int v = 0; v < VEC; ++v) e[u][v] = 0.f;
          one[u] = 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < kFwdUnroll; ++u) {
        const bool isdense = f[u] >= S && f[u] < F;
        const int j = isdense ? f[u] - S : 0;
        const float x = (isdense && Dn > 0) ? w_dense[sp * Dn + j] : 0.f;   // net.py:110-119
        float o1 = hit[u] ? one[u] : 0.f;
        o1 = isdense ? x * s_dw1[j] : o1;
        first += (lg == 0) ? o1 : 0.f;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          float ev = hit[u] ? e[u][v] : 0.f;
          if (isdense && d0 + v < D) ev = x * s_dw[j * D + d0 + v];
          e[u][v] = ev;
        }
        if (dvalid && f[u] < F) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            s[v] += e[u][v];
            q[v] += e[u][v] * e[u][v];
          }
          if (!compact || f[u] < S) {
            if (NT) vstore_nt<VEC>(fb + (int64_t)f[u] * D, e[u]); else vstore<VEC>(fb + (int64_t)f[u] * D, e[u]);
          }
          if (compact && f[u] == S) {
            float xr[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) xr[v] = (d0 + v < Dn) ? w_dense[sp * Dn + d0 + v] : 0.f;
            vstore<VEC>(fb + (int64_t)S * D, xr);
          }
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
    wave_lds_fence();  // staging area is rewritten at the top of the next iteration
  }
  if (oob) atomicOr(status, REC_FLAG_INDEX_OOB);
}

// ------------------------------------------------------------------------------------------ bwd
constexpr int kBwdUnroll = 4;
constexpr int kMaxDenseIters = 8;   // wave iterations that may contain dense fields

// PADDED: feat / dfeat at a caller-chosen sample stride (rec_deepfm_desc.feat_stride).  A template flag, not a run-time
// branch: with the stride as a plain kernel argument the dense-layout instantiation went from 164 to 180 VGPRs — 3 -> 2
// waves per SIMD, fm_bwd 65 -> 77 us back to back (caught by bench.py's roofline line dropping from 0.656 to 0.597).
template <int VEC, int LANES, int NDI, bool NT, bool PADDED = false>
__global__ __launch_bounds__(kBlock) void fm_bwd_kernel(
    int64_t B, int S, int Dn, int D, int FP, int64_t feat_ld, const float* __restrict__ dense,
    const float* __restrict__ feat, const float* __restrict__ sum_emb,
    const float* __restrict__ dfeat, const float* __restrict__ dy1, const float* __restrict__ dy2,
    const float* __restrict__ dense_w, float* __restrict__ row_grad, float* __restrict__ partial, int rg_nt,
    const int32_t* __restrict__ rank) {
  // rank (or null): the row gradient of lookup (b, f) goes to row rank[b*S+f] of row_grad instead of row b*S+f — the
  // SORTED order of the merge keys (rec_ids_group_slots), < 0 = a dropped lookup, not written
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
  const int nd = NIT - it_d0;      // <= NDI (dispatched on the host)
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;

  float acc[NDI][VEC];
  float acc1[NDI];
  float dwr[NDI][VEC];             // this lane's slice of dense_w for its dense field of iteration k
#pragma unroll
  for (int k = 0; k < NDI; ++k) {
    acc1[k] = 0.f;
    const int f = (it_d0 + k) * FS + fs;
    const bool isd = k < nd && f >= S && f < F;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      acc[k][v] = 0.f;
      dwr[k][v] = (dense_w && isd && d0 + v < D) ? dense_w[(f - S) * D + d0 + v] : 0.f;
    }
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
    const int64_t fo = PADDED ? b * feat_ld + d0 : (b * FP) * (int64_t)D + d0;
    const float* fb = feat + fo;
    const float* gb = dfeat + fo;
    const bool compact = FP != F;   // dense fields have no rows in feat / dfeat: their d_dnn part is handled
                                    // by the caller through the folded layer-0 weights (see header)
    float* rg = row_grad + (b * S) * (int64_t)D + d0;
    const int32_t* rkb = rank ? rank + b * S : nullptr;
    // wave iterations that (may) hold dense fields first: their loads are the irregular ones
    float ed[NDI][VEC], gd[NDI][VEC], xd[NDI];
    int rkd[NDI];
#pragma unroll
    for (int k = 0; k < NDI; ++k) {
      const int f = (it_d0 + k) * FS + fs;
      const bool ok = k < nd && dvalid && f < F;
      xd[k] = 0.f;
      rkd[k] = -1;
#pragma unroll
      for (int v = 0; v < VEC; ++v) ed[k][v] = gd[k][v] = 0.f;
      if (ok) {
        if (rkb && f < S) rkd[k] = rkb[f];
        if (!(compact && f >= S)) vload<VEC>(gd[k], gb + (int64_t)f * D);
        if (f >= S) xd[k] = dense[b * Dn + (f - S)];
        if (f < S || !dense_w) vload<VEC>(ed[k], fb + (int64_t)f * D);
      }
    }
    // wave iterations holding sparse fields only: d row = d_dnn + dy2 * (sum_emb - feat)
    for (int it0 = 0; it0 < it_d0; it0 += kBwdUnroll) {
      float e[kBwdUnroll][VEC], g[kBwdUnroll][VEC];
      int rk[kBwdUnroll];
#pragma unroll
      for (int u = 0; u < kBwdUnroll; ++u) {
        rk[u] = -1;
        if (dvalid && it0 + u < it_d0) {
          const int f = (it0 + u) * FS + fs;
          if (rkb) rk[u] = rkb[f];
          if (NT) {   // read-once streams: do not displace the table lines other kernels will want in L2
            vload_nt<VEC>(e[u], fb + (int64_t)f * D);
            vload_nt<VEC>(g[u], gb + (int64_t)f * D);
          } else {
            vload<VEC>(e[u], fb + (int64_t)f * D);
            vload<VEC>(g[u], gb + (int64_t)f * D);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < kBwdUnroll; ++u) {
        if (dvalid && it0 + u < it_d0) {
          const int f = (it0 + u) * FS + fs;
          float de[VEC];
#pragma unroll
          for (int v = 0; v < VEC; ++v) de[v] = g[u][v] + g2 * (sb[v] - e[u][v]);
          if (rkb) {
            if (rk[u] >= 0) {
              if (rg_nt) vstore_nt<VEC>(row_grad + (int64_t)rk[u] * D + d0, de);
              else vstore<VEC>(row_grad + (int64_t)rk[u] * D + d0, de);
            }
          } else if (rg_nt) vstore_nt<VEC>(rg + (int64_t)f * D, de); else vstore<VEC>(rg + (int64_t)f * D, de);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NDI; ++k) {
      const int f = (it_d0 + k) * FS + fs;
      if (k < nd && dvalid && f < F) {
        float de[VEC];
        const bool isd = f >= S;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const float e = (isd && dense_w) ? xd[k] * dwr[k][v] : ed[k][v];
          de[v] = gd[k][v] + g2 * (sb[v] - e);
        }
        if (!isd) {
          if (rkb) {
            if (rkd[k] >= 0) {
              if (rg_nt) vstore_nt<VEC>(row_grad + (int64_t)rkd[k] * D + d0, de);
              else vstore<VEC>(row_grad + (int64_t)rkd[k] * D + d0, de);
            }
          } else if (rg_nt) vstore_nt<VEC>(rg + (int64_t)f * D, de); else vstore<VEC>(rg + (int64_t)f * D, de);
        } else {
#pragma unroll
          for (int v = 0; v < VEC; ++v) acc[k][v] += xd[k] * de[v];
          if (lg == 0) acc1[k] += g1 * xd[k];
        }
      }
    }
  }

  // fold the SPW samples of the wave, then the waves of the block (fixed order)
#pragma unroll
  for (int k = 0; k < NDI; ++k) {
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
    for (int k = 0; k < NDI; ++k) {
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
  fm_fold_role<false>(blockIdx.x, threadIdx.x, kBlock, partial, nblk, split, out0, out1, FoldedLayer0{}, DenseAdam{}, 0, 0,
                      0);   // tail_roles.h
}

// Launch-geometry knobs, read once from the environment (tools/fm_sweep.py measures them; the defaults below are
// the measured best on MI355X, profiles/r02a_fm_sweep.txt): REC_FM_FWD_BPC / REC_FM_BWD_BPC = cap on resident
// blocks per CU (0 = occupancy limit; 2 and 3 measured slower), REC_FM_NT = 1 streams feat / d_feat with
// non-temporal accesses (fm_bwd 71.3 -> 64.5 us, fm_fwd 73.8 -> 72.9 us).
struct FmTune {
  int fwd_bpc, bwd_bpc, nt, fwd_nt, rg_nt, tile;
};
static const FmTune& tune() {
  static const FmTune t = [] {
    auto geti = [](const char* k, int dflt) {
      const char* v = getenv(k);
      return v && *v ? atoi(v) : dflt;
    };
    const int nt = geti("REC_FM_NT", 1);
    // REC_FM_FWD_NT: feat stores of the forward streamed (1) or cached (0); REC_FM_BWD_RG_NT: the backward's row
    // gradients streamed (1) or cached (0) — measured in-step, profiles/r03_fm_instep.txt
    // REC_FM_TILE=0: narrow rows (D 9 / 10 / 11 ...) on the row-group kernels instead of the block-tile kernels (fm_tile.h)
    return FmTune{geti("REC_FM_FWD_BPC", 0), geti("REC_FM_BWD_BPC", 0), nt, geti("REC_FM_FWD_NT", nt),
                  geti("REC_FM_BWD_RG_NT", 0), geti("REC_FM_TILE", 1)};
  }();
  return t;
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

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// narrow rows -> the block-tile kernels of fm_tile.h (everything they do not cover stays on the row-group kernels)
static bool fm_tile_shape(const rec_deepfm_desc* d) {
  return tune().tile && d->emb_dim <= kFmTileMaxD && d->emb_dim % 4 != 0 && !d->compact_dense &&
         kFmTileS * d->num_slots <= kBlock * kFmTileLook && d->num_dense * d->emb_dim + d->num_dense <= kBlock;
}

}  // namespace rec

using namespace rec;

static int fm_fwd_impl(const rec_deepfm_desc* desc, const int64_t* ids, const float* dense, const float* W, const float* W1,
                       const float* dense_w, const float* dense_w_one, const int64_t* slot_offset, float* y1, float* y2,
                       float* feat, float* sum_emb, int32_t* status, void* stream, const FoldFwd& fold, bool* rode,
                       bool* pad_zeroed);

extern "C" int rec_deepfm_fm_fwd(const rec_deepfm_desc* desc, const int64_t* ids,
                                 const float* dense, const float* W, const float* W1,
                                 const float* dense_w, const float* dense_w_one,
                                 const int64_t* slot_offset, float* y1, float* y2, float* feat,
                                 float* sum_emb, int32_t* status, void* stream) {
  return fm_fwd_impl(desc, ids, dense, W, W1, dense_w, dense_w_one, slot_offset, y1, y2, feat, sum_emb, status, stream,
                     FoldFwd{}, nullptr, nullptr);
}

// *rode = the fold went out with the lookup (the wide-row kernel); false: the caller issues rec_dense_fold_fwd_full
int rec::deepfm_fm_fwd_fold(const rec_deepfm_desc* desc, const int64_t* ids, const float* dense, const float* W,
                            const float* W1, const float* dense_w, const float* dense_w_one, const int64_t* slot_offset,
                            float* y1, float* y2, float* feat, float* sum_emb, int32_t* status, void* stream, FoldFwd fold,
                            bool* rode, bool* pad_zeroed) {
  return fm_fwd_impl(desc, ids, dense, W, W1, dense_w, dense_w_one, slot_offset, y1, y2, feat, sum_emb, status, stream,
                     fold, rode, pad_zeroed);
}

static int fm_fwd_impl(const rec_deepfm_desc* desc, const int64_t* ids, const float* dense, const float* W, const float* W1,
                       const float* dense_w, const float* dense_w_one, const int64_t* slot_offset, float* y1, float* y2,
                       float* feat, float* sum_emb, int32_t* status, void* stream, const FoldFwd& fold, bool* rode,
                       bool* pad_zeroed) {
  if (rode) *rode = false;
  if (pad_zeroed) *pad_zeroed = false;
  if (int rc = check_desc(desc)) return rc;
  if (desc->batch == 0) return REC_OK;
  REC_REQUIRE(ids && W && W1 && y1 && y2 && feat && status, REC_EINVAL, "null pointer argument");
  REC_REQUIRE(desc->num_dense == 0 || (dense && dense_w && dense_w_one), REC_EINVAL,
              "dense inputs missing");
  const int S = desc->num_slots, Dn = desc->num_dense, D = desc->emb_dim;
  hipStream_t st = (hipStream_t)stream;
  const int w1_stride = desc->w1_stride > 0 ? desc->w1_stride : 1;
  REC_REQUIRE(!desc->compact_dense || (Dn > 0 && Dn <= D), REC_ESHAPE,
              "compact_dense needs 0 < num_dense <= emb_dim");
  const int FP = desc->compact_dense ? S + 1 : S + Dn;
  const int64_t feat_ld = desc->feat_stride > 0 ? desc->feat_stride : (int64_t)FP * D;
  REC_REQUIRE(feat_ld >= (int64_t)FP * D && (desc->feat_stride <= 0 || feat_ld % 4 == 0), REC_EINVAL,
              "feat_stride %lld must be a multiple of 4 and >= %d fields x %d", (long long)feat_ld, FP, D);
  if (fm_tile_shape(desc) && aligned16(feat)) {
    const int P = fm_tile_pitch(FP, D, feat_ld);
    const size_t shmem = (size_t)fm_tile_fwd_lds(S, Dn, D, P).total_bytes;
    if (shmem <= 64 * 1024) {
      const bool v4 = desc->row_stride % 4 == 0 && desc->row_stride >= ((D + 3) & ~3) && aligned16(W);
      const int64_t ntiles = (desc->batch + kFmTileS - 1) / kFmTileS;
      const int zero_to = fold.zero_feat_pad && feat_ld > P && feat_ld != (int64_t)FP * D ? (int)feat_ld : 0;
      if (pad_zeroed) *pad_zeroed = zero_to > 0;
#define REC_FWD_TILE(V4_, NT_)                                                                                     \
  {                                                                                                                \
    int64_t grid = resident_blocks(fm_fwd_tile_kernel<V4_, NT_>, kBlock, shmem);                                   \
    if (grid > ntiles) grid = ntiles;                                                                              \
    hipLaunchKernelGGL((fm_fwd_tile_kernel<V4_, NT_>), dim3((unsigned)grid), dim3(kBlock), shmem, st, desc->batch, \
                       S, Dn, D, feat_ld, desc->row_stride, w1_stride, desc->num_rows, desc->padding_idx, ids,     \
                       dense, W, W1, dense_w, dense_w_one, slot_offset, y1, y2, feat, sum_emb, status,             \
                       zero_to);                                                                                   \
  }
      if (v4) { if (tune().fwd_nt) REC_FWD_TILE(true, true) else REC_FWD_TILE(true, false) }
      else { if (tune().fwd_nt) REC_FWD_TILE(false, true) else REC_FWD_TILE(false, false) }
#undef REC_FWD_TILE
      return check_launch("rec_deepfm_fm_fwd (tile)");
    }
  }
  return dispatch_row_shape(D, desc->row_stride, [&](auto vec, auto lanes) -> int {
    constexpr int VEC = decltype(vec)::value, LANES = decltype(lanes)::value;
    constexpr int SPW = kWave / (LANES * fs_for<LANES>());
    const int idch = (SPW * S + kWave - 1) / kWave;  // id loads per lane per tile
    REC_REQUIRE(idch <= 8, REC_ESHAPE, "num_slots %d too large at emb_dim %d (limit %d)", S, D,
                8 * kWave / SPW);
    const size_t hdr = ((size_t)(Dn * D + Dn) * 4 + 15) & ~(size_t)15;
    const size_t wave_bytes = ((size_t)SPW * S * 8 + (size_t)SPW * Dn * 4 + 15) & ~(size_t)15;
    const size_t shmem = hdr + (size_t)S * 8 + kWavesPerBlock * wave_bytes;
    REC_REQUIRE(shmem <= 64 * 1024, REC_ESHAPE, "LDS staging %zu B too large", shmem);
    const int64_t ntiles = (desc->batch + SPW - 1) / SPW;
    const int64_t want = (ntiles + kWavesPerBlock - 1) / kWavesPerBlock;
#define REC_FWD_LAUNCH2(IDCH, NT_)                                                                \
  {                                                                                               \
    int64_t grid = resident_blocks(fm_fwd_kernel<VEC, LANES, IDCH, NT_>, kBlock, shmem);          \
    if (tune().fwd_bpc > 0 && grid > (int64_t)tune().fwd_bpc * kNumCU) grid = (int64_t)tune().fwd_bpc * kNumCU; \
    if (grid > want) grid = want;                                                                 \
    if (grid > kMaxBlocks) grid = kMaxBlocks;                                                     \
    hipLaunchKernelGGL((fm_fwd_kernel<VEC, LANES, IDCH, NT_>), dim3((unsigned)(grid + fold.blocks)), dim3(kBlock), \
                       shmem, st, desc->batch, S, Dn, D, FP, feat_ld, desc->row_stride, w1_stride, \
                       desc->num_rows, desc->padding_idx, ids, dense, W, W1, dense_w, dense_w_one, \
                       slot_offset, y1, y2, feat, sum_emb, status, (int)grid, fold);               \
    if (rode) *rode = fold.blocks > 0;                                                            \
  }
#define REC_FWD_LAUNCH(IDCH) if (tune().fwd_nt) REC_FWD_LAUNCH2(IDCH, true) else REC_FWD_LAUNCH2(IDCH, false)
    if (idch <= 1) { REC_FWD_LAUNCH(1); }
    else if (idch <= 2) { REC_FWD_LAUNCH(2); }
    else if (idch <= 4) { REC_FWD_LAUNCH(4); }
    else { REC_FWD_LAUNCH(8); }
#undef REC_FWD_LAUNCH
#undef REC_FWD_LAUNCH2
    return check_launch("rec_deepfm_fm_fwd");
  });
}

extern "C" int rec_deepfm_fm_bwd_workspace_bytes(const rec_deepfm_desc* desc, size_t* bytes) {
  if (int rc = check_desc(desc)) return rc;
  REC_REQUIRE(bytes, REC_EINVAL, "bytes is NULL");
  const int K = desc->num_dense * desc->emb_dim + desc->num_dense;
  *bytes = align_up((size_t)kMaxBlocks * (K > 0 ? K : 1) * sizeof(float), 256);
  return REC_OK;
}

static int fm_bwd_impl(const rec_deepfm_desc* desc, const float* dense, const float* feat, const float* sum_emb,
                       const float* d_feat_dnn, const float* dy1, const float* dy2, const float* dense_w,
                       const int32_t* row_rank, float* row_grad, float* d_dense_w, float* d_dense_w_one,
                       void* workspace, size_t workspace_bytes, void* stream, int* defer_fold = nullptr);

extern "C" int rec_deepfm_fm_bwd(const rec_deepfm_desc* desc, const float* dense,
                                 const float* feat, const float* sum_emb, const float* d_feat_dnn,
                                 const float* dy1, const float* dy2, const float* dense_w,
                                 float* row_grad, float* d_dense_w, float* d_dense_w_one,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  return fm_bwd_impl(desc, dense, feat, sum_emb, d_feat_dnn, dy1, dy2, dense_w, nullptr, row_grad, d_dense_w,
                     d_dense_w_one, workspace, workspace_bytes, stream);
}

extern "C" int rec_deepfm_fm_bwd_sorted(const rec_deepfm_desc* desc, const float* dense, const float* feat,
                                        const float* sum_emb, const float* d_feat_dnn, const float* dy1,
                                        const float* dy2, const float* dense_w, const int32_t* row_rank,
                                        float* row_grad, float* d_dense_w, float* d_dense_w_one, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  REC_REQUIRE(row_rank || (desc && desc->batch == 0), REC_EINVAL, "row_rank is NULL");
  return fm_bwd_impl(desc, dense, feat, sum_emb, d_feat_dnn, dy1, dy2, dense_w, row_rank, row_grad, d_dense_w,
                     d_dense_w_one, workspace, workspace_bytes, stream);
}

int rec::deepfm_fm_bwd_partial(const rec_deepfm_desc* desc, const float* dense, const float* feat, const float* sum_emb,
                               const float* d_feat_dnn, const float* dy1, const float* dy2, const float* dense_w,
                               float* row_grad, void* workspace, size_t workspace_bytes, void* stream, int* nblk) {
  REC_REQUIRE(nblk && desc && desc->batch > 0, REC_EINVAL, "bad arguments");
  return fm_bwd_impl(desc, dense, feat, sum_emb, d_feat_dnn, dy1, dy2, dense_w, nullptr, row_grad, nullptr, nullptr,
                     workspace, workspace_bytes, stream, nblk);
}

static int fm_bwd_impl(const rec_deepfm_desc* desc, const float* dense, const float* feat, const float* sum_emb,
                       const float* d_feat_dnn, const float* dy1, const float* dy2, const float* dense_w,
                       const int32_t* row_rank, float* row_grad, float* d_dense_w, float* d_dense_w_one,
                       void* workspace, size_t workspace_bytes, void* stream, int* defer_fold) {
  // defer_fold: the fold launch is left to the caller (tail_roles.h); *defer_fold = blocks that wrote partial columns
  if (int rc = check_desc(desc)) return rc;
  const int S = desc->num_slots, Dn = desc->num_dense, D = desc->emb_dim;
  REC_REQUIRE(Dn == 0 || defer_fold || (d_dense_w && d_dense_w_one), REC_EINVAL, "dense args missing");
  REC_REQUIRE(desc->batch == 0 || (feat && sum_emb && d_feat_dnn && dy1 && dy2 && row_grad &&
                                   (Dn == 0 || dense)),
              REC_EINVAL, "null pointer argument");
  size_t need = 0;
  rec_deepfm_fm_bwd_workspace_bytes(desc, &need);
  REC_REQUIRE(workspace && workspace_bytes >= need, REC_EWORKSPACE, "workspace %zu < %zu",
              workspace_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  const int K = Dn * D + Dn;
  REC_REQUIRE(!desc->compact_dense || (Dn > 0 && Dn <= D && dense_w), REC_ESHAPE,
              "compact_dense needs 0 < num_dense <= emb_dim and dense_w");
  const int FP = desc->compact_dense ? S + 1 : S + Dn;
  const int64_t feat_ld = desc->feat_stride > 0 ? desc->feat_stride : (int64_t)FP * D;
  REC_REQUIRE(feat_ld >= (int64_t)FP * D && (desc->feat_stride <= 0 || feat_ld % 4 == 0), REC_EINVAL,
              "feat_stride %lld must be a multiple of 4 and >= %d fields x %d", (long long)feat_ld, FP, D);
  if (desc->batch == 0) {
    if (K) {
      (void)hipMemsetAsync(d_dense_w, 0, (size_t)Dn * D * sizeof(float), st);
      (void)hipMemsetAsync(d_dense_w_one, 0, (size_t)Dn * sizeof(float), st);
    }
    return REC_OK;
  }
  if (fm_tile_shape(desc) && !row_rank && aligned16(row_grad) && K > 0) {
    const size_t shmem = (size_t)fm_tile_bwd_lds(S, Dn, D).total_floats * sizeof(float);
    const bool v4 = feat_ld % 4 == 0 && aligned16(feat) && aligned16(d_feat_dnn);
    const int64_t ntiles = (desc->batch + kFmTileS - 1) / kFmTileS;
    float* partial = (float*)workspace;
    int grid = 1;
#define REC_BWD_TILE(V4_, NT_)                                                                                     \
  {                                                                                                                \
    /* persistent blocks (one block per tile was measured: fwd equal, bwd 59 -> 66 us — 4096 partial columns) */   \
    int64_t g = resident_blocks(fm_bwd_tile_kernel<V4_, NT_>, kBlock, shmem);                                      \
    if (g > ntiles) g = ntiles;                                                                                    \
    if (g > kMaxBlocks) g = kMaxBlocks;                                                                            \
    grid = (int)g;                                                                                                 \
    hipLaunchKernelGGL((fm_bwd_tile_kernel<V4_, NT_>), dim3(grid), dim3(kBlock), shmem, st, desc->batch, S, Dn, D, \
                       feat_ld, dense, feat, sum_emb, d_feat_dnn, dy1, dy2, dense_w, row_grad, partial);           \
  }
    if (v4) { if (tune().nt) REC_BWD_TILE(true, true) else REC_BWD_TILE(true, false) }
    else { REC_BWD_TILE(false, false) }
#undef REC_BWD_TILE
    if (defer_fold) *defer_fold = grid;
    else hipLaunchKernelGGL(fold_partials_kernel, dim3(K), dim3(kBlock), 0, st, partial, grid, Dn * D, d_dense_w, d_dense_w_one);
    return check_launch("rec_deepfm_fm_bwd (tile)");
  }
  return dispatch_row_shape(D, D, [&](auto vec, auto lanes) -> int {
    constexpr int VEC = decltype(vec)::value, LANES = decltype(lanes)::value;
    constexpr int FS = fs_for<LANES>();
    constexpr int SPW = kWave / (LANES * FS);
    const int nit = (S + Dn + FS - 1) / FS;
    const int nd = nit - S / FS;
    REC_REQUIRE(nd <= kMaxDenseIters, REC_ESHAPE,
                "num_dense %d spans more than %d wave iterations at this emb_dim", Dn,
                kMaxDenseIters);
    const int64_t spb = (int64_t)SPW * kWavesPerBlock;
    const int64_t need_blocks = (desc->batch + spb - 1) / spb;
    const size_t shmem = (size_t)kWavesPerBlock * (K > 0 ? K : 1) * sizeof(float);
    float* partial = (float*)workspace;
    int grid = 1;
#define REC_BWD_LAUNCH2(NDI, NT_)                                                                 \
  if (desc->feat_stride > 0) REC_BWD_LAUNCH3(NDI, NT_, true) else REC_BWD_LAUNCH3(NDI, NT_, false)
#define REC_BWD_LAUNCH3(NDI, NT_, PAD_)                                                           \
  {                                                                                               \
    int64_t g = resident_blocks(fm_bwd_kernel<VEC, LANES, NDI, NT_, PAD_>, kBlock, shmem);        \
    if (tune().bwd_bpc > 0 && g > (int64_t)tune().bwd_bpc * kNumCU) g = (int64_t)tune().bwd_bpc * kNumCU; \
    if (g > need_blocks) g = need_blocks;                                                         \
    if (g > kMaxBlocks) g = kMaxBlocks;                                                           \
    grid = (int)g;                                                                                \
    hipLaunchKernelGGL((fm_bwd_kernel<VEC, LANES, NDI, NT_, PAD_>), dim3(grid), dim3(kBlock), shmem, st, \
                       desc->batch, S, Dn, D, FP, feat_ld, dense, feat, sum_emb, d_feat_dnn, dy1, dy2, \
                       dense_w, row_grad, partial, tune().rg_nt, row_rank);                        \
  }
#define REC_BWD_LAUNCH(NDI) if (tune().nt) REC_BWD_LAUNCH2(NDI, true) else REC_BWD_LAUNCH2(NDI, false)
    if (nd <= 1) { REC_BWD_LAUNCH(1); }
    else if (nd <= 2) { REC_BWD_LAUNCH(2); }
    else if (nd <= 4) { REC_BWD_LAUNCH(4); }
    else { REC_BWD_LAUNCH(8); }
#undef REC_BWD_LAUNCH
#undef REC_BWD_LAUNCH2
#undef REC_BWD_LAUNCH3
    if (defer_fold) {
      *defer_fold = grid;
    } else if (K > 0) {
      hipLaunchKernelGGL(fold_partials_kernel, dim3(K), dim3(kBlock), 0, st, partial, grid, Dn * D,
                         d_dense_w, d_dense_w_one);
    }
    return check_launch("rec_deepfm_fm_bwd");
  });
}
