// The dense tail of a launch-bound DeepFM step as ROLES of one launch.
//
// At the reference's own batch size (deepfm/config_bigdata.yaml:23, bs 512) every kernel of the step is a few
// microseconds long and waits for the one before it: the step costs what its launches cost (~4.2 us each, back to back).
// Four of them are tiny parameter-space kernels that depend on nothing but earlier launches of the step:
//   ctr_head_fold_kernel        the last Linear's dW | db | loss from the fused head's per-block partial rows   (head_ops.hip)
//   fold_partials_kernel        d dense_w | d dense_w_one from the FM backward's per-block partial rows         (deepfm_fm.hip)
//   dense_fold_bwd_full_kernel  dW_0' (folded layer 0) -> dW_0 and its share of d dense_w                       (cross_ops.hip)
//   adam_dense_kernel           Adam on every dense parameter                                                   (sparse_update.hip)
// and none of them touches what the merged row update of the same step reads or writes.  Here each is a ROLE: a
// __device__ function over (block-of-the-role, thread) that the kernel of its own launch calls — and that
// small_tail_kernel (sparse_update.hip) calls from the blocks behind its row-bucket blocks, so that ONE launch does the
// row update, the three folds and Adam.  Inside one launch nothing orders two blocks, so the parameters are dealt out by
// OWNER: the block that folds a gradient applies Adam to exactly the parameters that gradient belongs to, and reads the
// old value of a parameter only if it owns it:
//   head-fold block  (16 columns)            -> w_last[j], b_last
//   FM-fold block k  (< Dn D, folded layer)  -> dense_w[k] AND row S D + k of W_0 (its gradient is dense_w[k] x dM[k / D, :],
//                                              the fold's own reduction reads that row's old values)
//   FM-fold block k  (otherwise)             -> dense_w[k] / dense_w_one[k - Dn D]
//   elementwise blocks                       -> the sparse rows of W_0 (gradient: dW_0' as it is) and every other parameter
// Every value is computed by the statements of the role, in the role's order, whoever calls it: the fused launch is
// bit-identical to the four launches (tests/test_deepfm_step_c.py).
#pragma once
#include "rec_common.h"

namespace rec {

struct DenseAdam {          // the flat parameter buffer and its moments (rec_adam_dense), bias-corrected scalars
  float* p;
  float* m;
  float* v;
  float lr_t, eps_t, b1, b2;
};

__device__ __forceinline__ void adam_dense_elem(float& p, float& m, float& v, float gi, float lr_t, float eps_t, float b1,
                                                float b2) {
  const float mi = b1 * m + (1.f - b1) * gi;
  const float vi = b2 * v + (1.f - b2) * gi * gi;
  m = mi;
  v = vi;
  p = p - lr_t * (mi / (sqrtf(vi) + eps_t));
}
__device__ __forceinline__ void adam_dense_at(const DenseAdam& a, int64_t i, float gi) {
  float p = a.p[i], m = a.m[i], v = a.v[i];
  adam_dense_elem(p, m, v, gi, a.lr_t, a.eps_t, a.b1, a.b2);
  a.m[i] = m;
  a.v[i] = v;
  a.p[i] = p;
}

// ---- column j of the [nblk][n2] partial rows of the fused CTR head, blocks in ascending order.  A block owns 16 columns,
// thread (c, q) the rows q, q + 16, ... of column c on four interleaved chains (loads in flight instead of one dependent
// chain per column), then the 16 row groups fold in ascending q.  Columns < n2 - 2 -> dw, n2 - 2 -> db,
// n2 - 1 -> loss = sum of costs * invB.  Threads >= 256 of a larger block only meet the barrier.
template <bool ADAM>
__device__ __forceinline__ void ctr_head_fold_role(int blk, int tid, int nblk, int n2, const float* __restrict__ partial,
                                                   float invB, float* __restrict__ dw, float* __restrict__ db,
                                                   float* __restrict__ loss, const DenseAdam& adam, int64_t w_off,
                                                   int64_t b_off) {
  __shared__ float red[16][17];
  const bool act = tid < kBlock;
  const int c = tid & 15, q = (tid >> 4) & 15;
  const int j = blk * 16 + c;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  if (act && j < n2) {
    int r = q;
    for (; r + 48 < nblk; r += 64) {
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] += partial[(int64_t)(r + 16 * u) * n2 + j];
    }
    for (int u = 0; r < nblk; r += 16, ++u) a[u] += partial[(int64_t)r * n2 + j];
  }
  if (act) red[q][c] = (a[0] + a[1]) + (a[2] + a[3]);
  __syncthreads();
  if (act && q == 0 && j < n2) {
    float t = red[0][c];
#pragma unroll
    for (int k = 1; k < 16; ++k) t += red[k][c];
    if (j < n2 - 2) {
      dw[j] = t;
      if constexpr (ADAM) adam_dense_at(adam, w_off + j, t);
    } else if (j == n2 - 2) {
      db[0] = t;
      if constexpr (ADAM) adam_dense_at(adam, b_off, t);
    } else {
      loss[0] = t * invB;
    }
  }
}

// ---- red[0] = sum of x over the first 256 threads: the fixed-order tree of fold_partials_kernel / dense_fold_bwd_*
__device__ __forceinline__ float tree256(float x, int tid, float* red) {
  if (tid < kBlock) red[tid] = x;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  return red[0];
}

struct FoldedLayer0 {       // layer 0 on folded weights (rec_dense_fold_*_full): null dW0f = no folded layer
  int S, Dn, D, NO;
  const float* dW0f;        // [(S + 1) D, NO]: gradient of the folded weight; its rows S D.. hold dM [Dn, NO]
  float* dW0;               // [(S + Dn) D, NO]: gradient of the real weight
  const float* W0;          // the real weight (old values)
  const float* dense_w;     // [Dn, D] (old values)
};

// ---- out[k] = sum_blk partial[k][blk]: one block per k, strided partial sums + fixed-order tree (fold_partials_kernel).
// ADAM: the block is the OWNER of parameter k (see the header): with a folded layer 0 and k < split it also folds
// d dense_w[k] += sum_n dM[k / D, n] W0[S D + k, n] (dense_fold_bwd_full_kernel's last role), writes row S D + k of dW0
// (= dense_w[k] x dM[k / D, :], that kernel's middle role) and applies Adam to that row and to dense_w[k].
template <bool ADAM>
__device__ __forceinline__ void fm_fold_role(int k, int tid, int nthreads, const float* __restrict__ partial, int nblk,
                                             int split, float* __restrict__ out0, float* __restrict__ out1,
                                             const FoldedLayer0& f, const DenseAdam& adam, int64_t dw_off, int64_t dw1_off,
                                             int64_t w0_off) {
  __shared__ float red[kBlock];
  float t = 0.f;
  if (tid < kBlock)
    for (int i = tid; i < nblk; i += kBlock) t += partial[(int64_t)k * nblk + i];
  const float v0 = tree256(t, tid, red);
  if constexpr (!ADAM) {
    if (tid == 0) {
      if (k < split) out0[k] = v0; else out1[k - split] = v0;
    }
  } else {
    if (k >= split) {
      if (tid == 0) {
        out1[k - split] = v0;
        adam_dense_at(adam, dw1_off + (k - split), v0);
      }
      return;
    }
    if (f.dW0f == nullptr) {
      if (tid == 0) {
        out0[k] = v0;
        adam_dense_at(adam, dw_off + k, v0);
      }
      return;
    }
    const int jd = k, j = jd / f.D, NO = f.NO;
    const float* dM = f.dW0f + (int64_t)f.S * f.D * NO;
    const float dw_old = f.dense_w[jd];
    __syncthreads();                                    // (red is written again)
    float u = 0.f;
    if (tid < kBlock)
      for (int n = tid; n < NO; n += kBlock) u += dM[j * NO + n] * f.W0[(int64_t)(f.S * f.D + jd) * NO + n];
    const float v1 = tree256(u, tid, red);              // (every read of the row's old values is behind this barrier)
    for (int n = tid; n < NO; n += nthreads) {
      const float g = dw_old * dM[j * NO + n];
      const int64_t e = (int64_t)(f.S * f.D + jd) * NO + n;
      f.dW0[e] = g;
      adam_dense_at(adam, w0_off + e, g);
    }
    if (tid == 0) {
      const float g = v0 + v1;
      out0[jd] = g;
      adam_dense_at(adam, dw_off + jd, g);
    }
  }
}

// ---- layer 0's folded weight (rec_dense_fold_fwd_full): W0f[r, :] = W0[r, :] for r < S D (blocks < copy_blocks, elementwise);
// W0f[S D + j, n] = sum_d dense_w[j, d] W0[(S + j) D + d, n] (the blocks behind them).  Its own launch — or blocks behind the
// lookup's (fm_fwd_kernel): the fold reads parameters only, the lookup the table and dense_w, the first GEMM needs both.
struct FoldFwd {
  int blocks;               // copy_blocks + the M blocks; 0 = no fold in this launch
  int S, Dn, D, NO, copy_blocks;
  const float* dw;          // dense_w [Dn, D]
  const float* W0;          // [(S + Dn) D, NO]
  float* W0f;               // [(S + 1) D, NO]
  int zero_feat_pad;        // narrow-row lookup (fm_tile.h) on a padded sample stride: the kernel also zeroes the columns
                            // behind its float4 rows up to feat_stride (a caller's scratch buffer needs no memset launch)
};
__device__ __forceinline__ void dense_fold_fwd_role(int b, int tid, const FoldFwd& r) {
  const int S = r.S, Dn = r.Dn, D = r.D, NO = r.NO;
  if (b < r.copy_blocks) {
    const int64_t total = (int64_t)S * D * NO;
    for (int64_t e = (int64_t)b * kBlock + tid; e < total; e += (int64_t)r.copy_blocks * kBlock) r.W0f[e] = r.W0[e];
    return;
  }
  const int nb = r.blocks - r.copy_blocks;
  float* M = r.W0f + (int64_t)S * D * NO;
  for (int e = (b - r.copy_blocks) * kBlock + tid; e < Dn * NO; e += nb * kBlock) {
    const int j = e / NO, n = e % NO;
    float t = 0.f;
    for (int d = 0; d < D; ++d) t += r.dw[j * D + d] * r.W0[(int64_t)((S + j) * D + d) * NO + n];
    M[e] = t;
  }
}

// ---- what small_tail_kernel's role blocks get (filled by rec_deepfm_train_step; the Adam scalars by the launcher)
struct TailRoles {
  int head_blocks, fm_blocks, w0_blocks, rest_blocks;
  int head_nblk, head_n2;
  const float* head_partial;
  float head_invB;
  float* head_dw;
  float* head_db;
  float* loss;
  int64_t head_w_off, head_b_off;
  int fm_nblk, fm_split;
  const float* fm_partial;
  float* ddw;
  float* ddw1;
  int64_t dw_off, dw1_off, w0_off;
  FoldedLayer0 f;
  int64_t flat_numel;
  const float* flat_grad;
  DenseAdam adam;
  int n_skip;
  int64_t skip_lo[5], skip_hi[5];
};

// host side (internal to the library: the whole-step entry points call them) ------------------------------------------
// true when rec_sparse_adam_record_small runs its row-bucket kernel for these sizes — the launch the roles can ride in
bool small_tail_eligible(int64_t n, int32_t emb_dim, const int64_t* slot_offset, int32_t num_slots);
// rec_sparse_adam_record_small + the roles, ONE launch (small_tail_kernel); roles.adam.{lr_t, eps_t, b1, b2} from hyper
int sparse_adam_record_small_tail(int64_t n, int32_t num_slots, int32_t emb_dim, int32_t rec_stride, int32_t state_stride,
                                  int32_t v_offset, int64_t num_rows, int64_t padding_idx, const int64_t* ids,
                                  const int64_t* slot_offset, const float* grad, const rec_grad_layout* grad_layout,
                                  const float* grad1, const rec_grad_layout* grad1_layout, float* rec, float* MV,
                                  const rec_adam_hyper* hyper, int32_t* status, TailRoles roles, void* stream);
// rec_ctr_head_fwd_bwd without its fold launch: the partial rows [*nblk][n + 2] stay in workspace (head_ops.hip)
int ctr_head_fwd_bwd_partial(int64_t batch, int32_t n, int64_t mean_over, const float* act, int64_t ld_act, const float* w,
                             const float* bias, const float* y1, const float* y2, const int64_t* label, float eps,
                             float clip_lo, float clip_hi, int32_t relu, float* y_dnn, float* pred, float* dz, float* dx,
                             int64_t ld_dx, void* workspace, size_t workspace_bytes, void* stream, int* nblk,
                             float* invB);
// rec_deepfm_fm_fwd with layer 0's weight fold riding behind the lookup's blocks (fold.blocks == 0: the plain lookup)
int deepfm_fm_fwd_fold(const rec_deepfm_desc* desc, const int64_t* ids, const float* dense, const float* W, const float* W1,
                       const float* dense_w, const float* dense_w_one, const int64_t* slot_offset, float* y1, float* y2,
                       float* feat, float* sum_emb, int32_t* status, void* stream, FoldFwd fold, bool* rode,
                       bool* pad_zeroed);
// the FoldFwd of rec_dense_fold_fwd_full's arguments (cross_ops.hip)
FoldFwd dense_fold_fwd_plan(int32_t num_slots, int32_t num_dense, int32_t emb_dim, int32_t n_out, const float* dense_w,
                            const float* W0, float* W0_folded);
// DIN at its shipped batch size (din/config.yaml:20, bs 32) -------------------------------------------------------------
// rec_sparse_sgd_small_multi with the dense parameters' SGD (rec_sgd_dense: p -= lr g) in blocks behind the merges' —
// the tables and the flat dense buffer are disjoint, both only wait for the backward (sparse_update.hip)
int sparse_sgd_small_multi_dense(int32_t count, const rec_small_sgd_job* jobs, float lr, int32_t* status, void* stream,
                                 int64_t dense_n, float* dense_p, const float* dense_g);
// up to 4 rec_emb_gather calls in ONE launch (emb_ops.hip): DIN's target item / target category / item bias rows
struct GatherJob {
  int64_t n;
  int32_t emb_dim, row_stride;
  int64_t num_rows, padding_idx;
  const int64_t* ids;
  const float* W;
  float* out;
  int32_t out_group;
  int64_t out_group_stride;
};
constexpr int kGatherJobsMax = 4;
struct GatherJobs {         // what the kernel gets: the jobs and the first block of each (256 threads, one float per thread)
  int count, blocks;
  int block0[kGatherJobsMax + 1];
  GatherJob j[kGatherJobsMax];
};
// block b (of js.blocks) of the gathers: the values, the padding rule and the out-of-range flag of emb_gather_kernel
__device__ __forceinline__ void gather_role(int b, int tid, const GatherJobs& js, int32_t* __restrict__ status) {
  int k = 0;
#pragma unroll
  for (int i = 1; i < kGatherJobsMax; ++i)
    if (i < js.count && b >= js.block0[i]) k = i;
  const GatherJob& g = js.j[k];
  const int D = g.emb_dim;
  const int64_t e = (int64_t)(b - js.block0[k]) * kBlock + tid;
  const int64_t i = e / D;
  const int d = (int)(e % D);
  if (tid >= kBlock || i >= g.n) return;
  const int64_t id = g.ids[i];
  float v = 0.f;
  if (id != g.padding_idx || g.padding_idx < 0) {
    if (id >= 0 && id < g.num_rows) v = g.W[id * g.row_stride + d];
    else if (d == 0) atomicOr(status, REC_FLAG_INDEX_OOB);
  }
  const int64_t o = g.out_group > 0 ? (i / g.out_group) * g.out_group_stride + (i % g.out_group) * D : i * D;
  g.out[o + d] = v;
}
int gather_jobs_make(int32_t count, const GatherJob* jobs, GatherJobs* out);       // emb_ops.hip (checks + block ranges)
// the gathers riding behind the blocks of the NEXT tile-split rec_din_attention_pool_fwd_ws's combine launch issued by this
// thread (din_attention.hip); din_combine_rider_take() -> true when that launch carried them (else: emb_gather_multi)
void din_combine_rider_set(const GatherJobs* jobs, int32_t* status);
bool din_combine_rider_take();
int emb_gather_multi(int32_t count, const GatherJob* jobs, int32_t* status, void* stream);
// rec_deepfm_fm_bwd without its fold launch: the partial columns [K][*nblk] stay in workspace (deepfm_fm.hip)
int deepfm_fm_bwd_partial(const rec_deepfm_desc* desc, const float* dense, const float* feat, const float* sum_emb,
                          const float* d_feat_dnn, const float* dy1, const float* dy2, const float* dense_w,
                          float* row_grad, void* workspace, size_t workspace_bytes, void* stream, int* nblk);

}  // namespace rec
