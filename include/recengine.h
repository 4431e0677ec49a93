/* recengine — C-ABI of the MI355X-native sparse-embedding + feature-interaction engine that
 * stands in for the Paddle ops on the hot path of PaddleRec's models/rank CTR stack.
 *
 * The reference has no FFI seam of its own: the seam is the Paddle operator API used inside
 * net.py.  Each entry point below names the reference call site whose Paddle
 * ops it replaces; a Paddle custom-op shim (PD_BUILD_OP) or the ctypes/torch
 * adapter in paddlerec_amd/ binds exactly these symbols.
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is a DEVICE pointer unless marked "host".
 *   - the caller owns every buffer (tables, outputs, workspace); nothing is retained.
 *   - every launch is asynchronous on `stream` (a hipStream_t passed as void*); no internal sync.
 *   - return 0 on success or a negative rec_status; rec_last_error() gives thread-local text.
 *   - `status` arguments are device int32 words the kernels OR error bits into
 *     (REC_FLAG_*), so out-of-range indices are reported without a host sync.
 *   - tables are row-major f32 [num_rows, row_stride] with row_stride >= emb_dim (floats).
 */
#ifndef RECENGINE_H_
#define RECENGINE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  REC_OK = 0,
  REC_EINVAL = -1,     /* null pointer / negative size / unsupported combination */
  REC_ESHAPE = -2,     /* dimension outside what the kernels are built for */
  REC_EWORKSPACE = -3, /* workspace too small (query *_workspace_bytes first) */
  REC_EHIP = -4,       /* HIP runtime error at launch */
  REC_ENCCL = -5
} rec_status;

/* bits OR-ed into a device status word */
#define REC_FLAG_INDEX_OOB 1 /* an id was < 0 or >= num_rows (Paddle raises on OOB [EXT]) */
#define REC_FLAG_EXCHANGE_OVERFLOW 2 /* host binders (paddlerec_amd/sharded.py, deduplicated exchange): a rank needed more
                                        distinct rows of one owner than the fixed exchange capacity; the rows behind it
                                        read as zeros and their gradients are dropped — raise the capacity */

const char* rec_last_error(void); /* host; thread-local */
int rec_version(void);            /* host */

/* ------------------------------------------------------------------------------------------
 * DeepFM: embedding lookup + FM first/second order, fused.
 * Replaces concat + Embedding x2 + multiply/unsqueeze/sum/square/concat of
 *   models/rank/deepfm/net.py:105-139 (FM.forward).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int64_t batch;       /* B */
  int32_t num_slots;   /* S  (26) : sparse fields, one id each */
  int32_t num_dense;   /* Dn (13) : dense fields, <= 16 */
  int32_t emb_dim;     /* D  <= 256 */
  int32_t row_stride;  /* floats between consecutive rows of W (>= D) */
  int64_t num_rows;    /* N rows in W / W1 (after slot offsets) */
  int64_t padding_idx; /* id that yields a zero row and no gradient; < 0 = none */
  int32_t w1_stride;   /* floats between consecutive entries of W1; 0 or 1 = dense [N] array.
                          HBM fetches whole 128-B lines: a table kept as 32-float records
                          [W(16) | W1 | ...] (W1 = W + 16, both strides 32) costs ONE line per
                          lookup instead of two (see DESIGN.md "table layout"). */
  int32_t compact_dense; /* 0: feat = [B, S+Dn, D] (= feat_embeddings, net.py:120).
                            1: feat = [B, S+1, D]: the S embedding rows plus ONE row holding the Dn raw dense
                            values (zero padded to D; needs Dn <= D).  The dense "embeddings" x_j*dense_w[j,:]
                            are never materialised: the FM sums still include them (computed in registers),
                            and the top MLP's first layer sees them through folded weights
                            (rec_dense_fold_fwd): feat'[B,(S+1)D] @ [W0_sparse; M; 0] == feat[B,(S+Dn)D] @ W0.
                            In this mode rec_deepfm_fm_bwd takes d_feat_dnn in the same [B,S+1,D] layout and
                            its d_dense_w holds the FM part only (the MLP part comes from rec_dense_fold_bwd). */
  int64_t feat_stride;   /* floats between consecutive samples of feat (and of d_feat_dnn in the backward); 0 = dense
                            (fields x emb_dim).  A caller whose first MLP layer wants an input width that is a multiple
                            of its GEMM tiles (39 fields x 10 = 390 -> 400) keeps feat in a zero-initialised
                            [B, feat_stride] buffer: the kernels never touch the padding columns. */
} rec_deepfm_desc;

/* ids [B,S] i64 (= paddle.concat(sparse_inputs,1), net.py:107); dense [B,Dn] f32;
 * W [N,stride], W1 [N]; dense_w [Dn,D]; dense_w_one [Dn];
 * slot_offset [S] i64 or NULL: row = id + slot_offset[s] (26 tables laid out as one);
 * outputs y1,y2 [B]; feat [B,S+Dn,D] (= feat_embeddings, net.py:120);
 * sum_emb [B,D] (= summed_features_emb, net.py:124; kept for backward). */
int rec_deepfm_fm_fwd(const rec_deepfm_desc* desc, const int64_t* ids, const float* dense,
                      const float* W, const float* W1, const float* dense_w,
                      const float* dense_w_one, const int64_t* slot_offset, float* y1, float* y2,
                      float* feat, float* sum_emb, int32_t* status, void* stream);

/* Backward of the block above (what loss.backward(), tools/trainer.py:151, runs for net.py:105-139).
 * In : dense, feat, sum_emb (saved by fwd), d_feat_dnn [B,S+Dn,D], dy1, dy2 [B];
 *      dense_w [Dn,D] or NULL: when given, the dense part of feat is recomputed as
 *      dense[b,j]*dense_w[j,:] (bit-identical to what fwd stored) instead of re-read.
 * Out: row_grad [B*S, D]  — SelectedRows.value of `embedding` (rows = flattened ids, unmerged);
 *      d_dense_w [Dn,D], d_dense_w_one [Dn] — batch sums, reduced in a fixed order (deterministic).
 *      The SelectedRows.value of `embedding_one` is dy1[b] for every (b,s); it is not
 *      materialised — rec_sparse_adam_rows reads dy1 through rec_grad_layout{S,0,0}. */
int rec_deepfm_fm_bwd_workspace_bytes(const rec_deepfm_desc* desc, size_t* bytes);
int rec_deepfm_fm_bwd(const rec_deepfm_desc* desc, const float* dense, const float* feat,
                      const float* sum_emb, const float* d_feat_dnn, const float* dy1,
                      const float* dy2, const float* dense_w, float* row_grad, float* d_dense_w,
                      float* d_dense_w_one, void* workspace, size_t workspace_bytes, void* stream);
/* The same backward with the SelectedRows value written in SORTED order: the gradient row of lookup (b, s) goes to
 * row_grad + row_rank[b*S+s] * D (row_rank from rec_ids_group_slots / rec_ids_rank; < 0 = padding, not written), so the
 * merge + optimizer kernel reads every row's duplicates as consecutive rows (rec_grad_layout.sorted = 1). */
int rec_deepfm_fm_bwd_sorted(const rec_deepfm_desc* desc, const float* dense, const float* feat,
                             const float* sum_emb, const float* d_feat_dnn, const float* dy1,
                             const float* dy2, const float* dense_w, const int32_t* row_rank, float* row_grad,
                             float* d_dense_w, float* d_dense_w_one, void* workspace, size_t workspace_bytes,
                             void* stream);

/* Folding the dense "embeddings" into the first MLP layer (compact_dense = 1 above).
 *   fwd: M[j,n] = sum_d dense_w[j,d] * W0[(S+j)*D + d, n]                    -> M [Dn, n_out]
 *        so that  sum_j x_j * (dense_w[j,:] @ W0_rows(j))  ==  x @ M   (net.py:110-119,170-171 re-associated)
 *   bwd: given dM [Dn, n_out] (rows S*D .. S*D+Dn-1 of feat'^T dZ0):
 *        dW0[(S+j)*D + d, n] = dense_w[j,d] * dM[j,n];   d_dense_w[j,d] (+)= sum_n dM[j,n] * W0[(S+j)*D+d, n]
 * W0 is the layer's weight [ (S+Dn)*D, n_out ] (Paddle [in,out]); dW0 its gradient buffer (only the dense rows
 * are written). */
int rec_dense_fold_fwd(int32_t num_slots, int32_t num_dense, int32_t emb_dim, int32_t n_out,
                       const float* dense_w, const float* W0, float* M, void* stream);
/* The same folds with the row copies around them, ONE launch each way (what a training step issues):
 *   fwd_full: W0_folded [(S+1)*D, n_out]: rows [0, S*D) = W0's, rows S*D + j = M[j,:] (the rest stays as the caller
 *             left it: zero);
 *   bwd_full: dW0_folded = feat'^T dZ0 [(S+1)*D, n_out] (a scratch of the caller's, NOT the gradient buffer): its sparse
 *             rows are copied into dW0, its rows S*D.. are dM -> the dense rows of dW0 and d_dense_w as above. */
int rec_dense_fold_fwd_full(int32_t num_slots, int32_t num_dense, int32_t emb_dim, int32_t n_out, const float* dense_w,
                            const float* W0, float* W0_folded, void* stream);
int rec_dense_fold_bwd_full(int32_t num_slots, int32_t num_dense, int32_t emb_dim, int32_t n_out, const float* dense_w,
                            const float* W0, const float* dW0_folded, float* dW0, float* d_dense_w,
                            int32_t accumulate_ddw, void* stream);
int rec_dense_fold_bwd(int32_t num_slots, int32_t num_dense, int32_t emb_dim, int32_t n_out,
                       const float* dense_w, const float* W0, const float* dM, float* dW0,
                       float* d_dense_w, int32_t accumulate_ddw, void* stream);

/* ------------------------------------------------------------------------------------------
 * Plain lookup and LoD sum-pool.
 *   rec_emb_gather         : nn.Embedding forward (deepfm/net.py:108,117; dcn_v2/net.py:93-96;
 *                            din/net.py:141-147).  out[i,:] = id==padding_idx ? 0 : W[id,:]
 *   rec_emb_gather_sumpool : sparse_embedding + sequence_pool('sum')
 *                            (models/rank/slot_dnn/net.py:63-75; dnn/static_model_lod.py:70-97).
 *                            out[b,:] = sum_{k in [lod[b],lod[b+1])} W[ids[k],:], padding ids skipped;
 *                            counts[b] = number of non-padding ids pooled (bit-exact).
 * ---------------------------------------------------------------------------------------- */
/* out_group > 0: lookup i is written at out + (i / out_group) * out_group_stride + (i % out_group) * D
 * (e.g. the S sparse fields of a sample into the head of a wider feature row, dcn_v2/net.py:100-108);
 * out_group <= 0: contiguous [n, D]. */
int rec_emb_gather(int64_t n, int32_t emb_dim, int32_t row_stride, int64_t num_rows,
                   int64_t padding_idx, const int64_t* ids, const float* W, float* out,
                   int32_t out_group, int64_t out_group_stride, int32_t* status, void* stream);
int rec_emb_gather_sumpool(int64_t batch, int32_t emb_dim, int32_t row_stride, int64_t num_rows,
                           int64_t padding_idx, const int64_t* ids, const int64_t* lod /*[B+1]*/,
                           const float* W, float* out, int32_t* counts, int32_t* status,
                           void* stream);
/* What an unborn row of a lazily created PS table reads as when the accessor does NOT zero-initialise embed_w
 * (rec_ps_accessor.embed_zero_init = 0; rec_ps_push_rows gives birth with the same values): element 0 (embed_w)
 * always, elements 1.. when init_dims > 1; keyed by (seed, row*row_mul+row_add, d).
 * init_range <= 0 (Paddle's default, zero_init): rows are plain memory — an unborn row reads as zeros. */
typedef struct {
  int32_t state_offset; /* floats from the row start to the state float (0 = unborn) */
  int32_t init_dims;
  float init_range;
  uint64_t seed;
  int64_t row_mul, row_add;
} rec_lazy_init;
/* Owner-side lookup of a row-sharded DeepFM table (tools/static_gpubox_trainer.py:152-160: the pull of the GPU
 * parameter server): out_w[i,:] = rec[rows[i], 0:D], out_w1[i] = rec[rows[i], D] — both embeddings of
 * deepfm/net.py:62-86 from the one record line of a row.  lazy may be NULL. */
int rec_record_gather(int64_t n, int32_t emb_dim, int32_t rec_stride, int64_t num_rows, const int64_t* rows,
                      const float* rec, float* out_w, float* out_w1, const rec_lazy_init* lazy,
                      int32_t* status, void* stream);

/* backward of the sum-pool: SelectedRows.value[k,:] = d_out[sample(k),:]  (rows = ids). */
int rec_emb_sumpool_bwd(int64_t batch, int32_t emb_dim, const int64_t* lod, const float* d_out,
                        float* row_grad, void* stream);

/* ------------------------------------------------------------------------------------------
 * ALL slots of a batch in one launch: the `for s_input in slot_inputs: sparse_embedding(...,
 * padding_idx=0, entry=ShowClickEntry) -> sequence_pool('sum')` loop + concat(axis=1) of
 * models/rank/slot_dnn/net.py:63-77 (408 slots x D 9) and dnn/static_model_lod.py:70-97.
 * Input = the slot-major CSR of rec_parse_feasign_slots, resident on the device:
 *   values [nnz]; ids of (slot s, sample b) = values[slot_base[s] + lod[s*lod_stride + b] ..
 *                                                    slot_base[s] + lod[s*lod_stride + b + 1])
 * key_mode 0: values are table rows (checked against [0,num_rows), REC_FLAG_INDEX_OOB otherwise);
 * key_mode 1: values are uint64 feasign bit patterns (what the reference feeds its PS hash map,
 *             queuedataset_reader.py:56-82); row = rec_feasign_rows' hash, 0 stays the padding row.
 * padding_idx is compared with the VALUE (before hashing); < 0 = no padding id.
 * Outputs: out [B, out_stride] with out[b, s*D:(s+1)*D] = sum of the segment's rows (padding skipped);
 *   counts [B,S] i32 (or NULL) = ids pooled per segment — bit-exact target;
 *   seg_of_value [nnz] i32 (or NULL) = b*S+s of every value: rec_grad_layout.index for the backward;
 *   rows_out [nnz] i64 (or NULL) = table row of every value (dropped ones -> the padding row): the `ids`
 *   of rec_ids_group(n = nnz, num_slots = 1) for the SelectedRows merge of the backward.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int64_t batch;
  int32_t num_slots;
  int32_t emb_dim;
  int32_t row_stride;  /* floats between rows of W */
  int32_t key_mode;    /* 0 rows, 1 uint64 feasigns */
  int64_t num_rows;
  int64_t padding_idx;
  int64_t lod_stride;  /* elements between slot rows of lod; 0 = batch + 1 */
  int64_t out_stride;  /* floats between samples of out; 0 = num_slots * emb_dim */
  /* PS tables create a feature at its first pull (MemorySparseTable + SparseAdaGradSGDRule::InitValue [EXT],
   * slot_dnn/config_online.yaml:66-79 initial_range): init_range > 0 makes a row whose state float
   * W[row*row_stride + state_offset] is 0 read as its creation values — element d < init_dims =
   * uniform(+-init_range) keyed by (init_seed, row, d), the rest 0 — without writing the table
   * (rec_ps_push_rows gives birth with the same values).  init_range <= 0: off. */
  int32_t state_offset;
  int32_t init_dims;
  float init_range;
  uint64_t init_seed;
} rec_multislot_desc;
int rec_multislot_sumpool_fwd(const rec_multislot_desc* desc, const int64_t* values, const int64_t* lod,
                              const int64_t* slot_base, const float* W, float* out, int32_t* counts,
                              int32_t* seg_of_value, int64_t* rows_out, int32_t* status, void* stream);
/* The backward of the pool in ROWS FORM — the SelectedRows value of sequence_pool('sum') o sparse_embedding
 * (slot_dnn/net.py:63-75): row_grad[k, :] = d_out[b, s*D:(s+1)*D] for value k of segment seg_of_value[k] = b*S+s
 * (rows = rows_out of the forward).  For binders that must hand the gradient over as a dense [nnz, D] tensor (the
 * Paddle custom operator rec_multislot_sumpool); the engine's own steps read d_out through rec_grad_layout.index
 * and never materialise it.  desc: batch, num_slots, emb_dim, out_stride are read. */
int rec_multislot_sumpool_bwd(const rec_multislot_desc* desc, int64_t nnz, const int32_t* seg_of_value,
                              const float* d_out, float* row_grad, void* stream);
/* Row of a feasign in a hashed table of num_rows (>= 2) rows: 0 -> 0 (padding row), f -> 1 + mix64(f) %
 * (num_rows - 1) with mix64 = the murmur3 64-bit finaliser (SURVEY §8(d) cfg5 "64-bit mix % N").  Device
 * (keys/rows device pointers) and host variants, bit-identical. */
int rec_feasign_rows(int64_t n, int64_t num_rows, const int64_t* keys, int64_t* rows, void* stream);
int rec_feasign_rows_host(int64_t n, int64_t num_rows, const uint64_t* keys, int64_t* rows);

/* ------------------------------------------------------------------------------------------
 * SelectedRows merge (MergeAdd [EXT]) — integer part: group the n = B*S lookups by row.
 * Replaces the duplicate-row merge every consumer of a sparse=True embedding gradient performs
 * (deepfm/net.py:62-70,80 `sparse=use_sparse`).
 *   sorted_pos [n] i32 : positions (b*S+s) stably sorted by row; padding hits are dropped
 *   uniq_rows  [n] i64 : first n_uniq entries = distinct rows, ascending
 *   seg_offset [n+1] i32: sorted_pos[seg_offset[u] .. seg_offset[u+1]) belong to uniq_rows[u]
 *   n_uniq     [4] i32 : {number of distinct rows, number of non-padding positions, 1 if some row owns >=
 *                         REC_SEG_LONG positions else 0, 0}
 * All four are bit-exact targets.
 * ---------------------------------------------------------------------------------------- */
int rec_ids_group_workspace_bytes(int64_t n, int64_t num_rows, size_t* bytes);
int rec_ids_group(int64_t n, int32_t num_slots, int64_t num_rows, int64_t padding_idx,
                  const int64_t* ids, const int64_t* slot_offset, int32_t* sorted_pos,
                  int64_t* uniq_rows, int32_t* seg_offset, int32_t* n_uniq, int32_t* status,
                  void* workspace, size_t workspace_bytes, void* stream);
/* The same grouping with a caller-chosen 32-bit payload travelling with every lookup instead of its position:
 * sorted_pos[k] = payload[position].  The multi-slot CSR path (slot_dnn/net.py:63-75) passes the (sample, slot)
 * segment of every value (rec_multislot_sumpool_fwd's seg_of_value), so the row-update kernels find a position's
 * gradient row without the random read of rec_grad_layout.index that sorting positions would need (22.7 M
 * 4-byte reads, a memory line each, on the slot_dnn benchmark shape).  Stable in the positions, as above. */
int rec_ids_group_payload(int64_t n, int32_t num_slots, int64_t num_rows, int64_t padding_idx,
                          const int64_t* ids, const int64_t* slot_offset, const int32_t* payload,
                          int32_t* sorted_pos, int64_t* uniq_rows, int32_t* seg_offset, int32_t* n_uniq,
                          int32_t* status, void* workspace, size_t workspace_bytes, void* stream);

/* The same grouping for a [batch, num_slots] id batch whose slot s owns rows [s * slot_rows, (s + 1) * slot_rows) of the
 * table (BASELINE configs[1]: 26 tables x 1 000 000 rows as one table; deepfm/net.py:62-86 with one nn.Embedding per
 * slot) — i.e. rec_ids_group with slot_offset[s] = s * slot_rows and num_rows = num_slots * slot_rows, bit-identical
 * outputs, except that an id outside [0, slot_rows) is flagged REC_FLAG_INDEX_OOB and dropped (with separate tables it
 * IS out of range).  The slot digit of the key is free (column s of the batch is slot s), so the sort is num_slots
 * independent two-digit sorts inside L2-resident 256 KB slices: 7 launches instead of 13 (csrc/ids_group_slots.hip);
 * shapes that path does not cover (batch < 8192, slot_rows > 2^20, num_slots > 60) take the general sort.
 *   rank [batch*num_slots] i32 or NULL: rank[pos] = k with sorted_pos[k] = pos, -1 for dropped lookups — where a
 *   producer writes the gradient row of lookup pos so that the gradient buffer is in sorted order
 *   (rec_grad_layout.sorted). */
int rec_ids_group_slots_workspace_bytes(int64_t batch, int32_t num_slots, int64_t slot_rows, size_t* bytes);
int rec_ids_group_slots(int64_t batch, int32_t num_slots, int64_t slot_rows, int64_t padding_idx, const int64_t* ids,
                        int32_t* sorted_pos, int64_t* uniq_rows, int32_t* seg_offset, int32_t* n_uniq, int32_t* rank,
                        int32_t* status, void* workspace, size_t workspace_bytes, void* stream);
/* rank of any grouping whose sorted_pos holds positions (no payload): rank[0, n) = -1, then rank[sorted_pos[k]] = k for
 * k < n_uniq[1]. */
int rec_ids_rank(int64_t n, const int32_t* n_uniq, const int32_t* sorted_pos, int32_t* rank, void* stream);

/* ------------------------------------------------------------------------------------------
 * Optimizers.  paddle.optimizer.Adam [EXT] (deepfm/dygraph_model.py:61-65, static_model.py:83-84):
 *   lr_t = lr*sqrt(1-b2^t)/(1-b1^t);  m = b1*m+(1-b1)*g;  v = b2*v+(1-b2)*g*g;
 *   p -= lr_t * m / (sqrt(v) + eps*sqrt(1-b2^t))
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  float lr, beta1, beta2, eps;
  int64_t step; /* t, 1-based */
} rec_adam_hyper;

/* Where the gradient row of lookup position `pos` (= b*S+s) lives inside `grad`:
 *   p = index ? index[pos] : pos;  q = p / div;
 *   offset = group > 0 ? (q / group) * group_stride + (q % group) * D : q * D
 * {1,0,0}: contiguous [n,D] (DeepFM row_grad); {S,0,0} with D = 1: dy1 [B] for the first-order table;
 * {1,S,d}: the first S*D columns of a [B,d] feature gradient (DCN-v2);
 * index = seg_of_value of rec_multislot_sumpool_fwd: every id of a pooled (sample, slot) segment reads the ONE
 * gradient row of that segment, d_out[b, s*D:(s+1)*D] — the backward of sequence_pool('sum') without a
 * materialised [nnz, D] row-gradient tensor (slot_dnn/net.py:73). */
typedef struct {
  int32_t div;
  int32_t group;
  int64_t group_stride;
  const float* partials; /* device, from rec_segment_partials for THIS grad + grouping, or NULL */
  const int32_t* index;  /* device [n] or NULL: lookup position -> gradient position */
  int32_t sorted;        /* 1: grad is [n, D] in SORTED order — row k belongs to sorted position k (written through
                          * the rank of rec_ids_group_slots / rec_ids_rank, e.g. by rec_deepfm_fm_bwd_sorted); div,
                          * group, group_stride and index are then ignored.  A row's duplicate gradients are
                          * consecutive rows: the update kernels stream them instead of chasing sorted_pos. */
} rec_grad_layout;

/* Hot rows.  With Zipf-distributed ids one row can own tens of thousands of the n lookups; the per-row
 * duplicate sum of the consumers below would then be a serial chain of that many reads.  rec_segment_partials
 * pre-reduces, per tile of REC_SEG_TILE sorted positions, the pieces of every segment of >= REC_SEG_LONG
 * positions into partials [ceil(n_max/REC_SEG_TILE), 2, emb_dim] (fixed summation tree: deterministic); a
 * consumer given grad_layout.partials adds one partial per tile for such rows.  Optional: with partials =
 * NULL every consumer sums position by position.  grad_layout->partials is ignored on input here. */
#define REC_SEG_TILE 64
#define REC_SEG_LONG 128
int rec_segment_partials_bytes(int64_t n_max, int32_t emb_dim, size_t* bytes);
int rec_segment_partials(int64_t n_max, int32_t emb_dim, const int32_t* n_uniq,
                         const int32_t* seg_offset, const int32_t* sorted_pos, const float* grad,
                         const rec_grad_layout* grad_layout, float* partials, void* stream);

/* lazy_mode=True Adam on the rows of a merged SelectedRows gradient, merge fused in:
 *   g[u,:] = sum_{k in seg(u)} grad_row(sorted_pos[k])            (ascending position order)
 * then Adam on rows uniq_rows[u] of P/M/V ([num_rows,row_stride]).  n_uniq is read on the device.
 * grad_layout NULL = {1,0,0}.  grad_scale (device float[1] or NULL): factor applied to the merged
 * gradient — the global-norm clipping coefficient of rec_clip_scale. */
int rec_sparse_adam_rows(int64_t n_max, int32_t emb_dim, int32_t row_stride,
                         int32_t state_stride /* row stride of M and V; 0 = row_stride */,
                         const int32_t* n_uniq, const int64_t* uniq_rows, const int32_t* seg_offset,
                         const int32_t* sorted_pos, const float* grad,
                         const rec_grad_layout* grad_layout, const float* grad_scale, float* P,
                         float* M, float* V, const rec_adam_hyper* hyper, void* stream);

/* Both embeddings of a DeepFM row (deepfm/net.py:62-86: `embedding` [N,D] and `embedding_one` [N,1]) in one
 * pass over the record layout  rec [N, rec_stride] = W(D) | W1 | m1 | v1 | pad,  MV [N, state_stride] = m(D) at
 * 0 | v(D) at v_offset:  W/m/v from (grad, grad_layout), W1/m1/v1 from (grad1, grad1_layout) — for DeepFM
 * grad1 = dz [B] with layout {S,0,0}.  Same arithmetic as two rec_sparse_adam_rows calls; one record line and
 * one state line are read and written per touched row instead of the record line twice. */
int rec_sparse_adam_record(int64_t n_max, int32_t emb_dim, int32_t rec_stride, int32_t state_stride,
                           int32_t v_offset, const int32_t* n_uniq, const int64_t* uniq_rows,
                           const int32_t* seg_offset, const int32_t* sorted_pos, const float* grad,
                           const rec_grad_layout* grad_layout, const float* grad1,
                           const rec_grad_layout* grad1_layout, const float* grad_scale, float* rec,
                           float* MV, const rec_adam_hyper* hyper, void* stream);

/* PS / gpubox accessor rule — "SparseAdaGradSGDRule" + show/click counters (models/rank/slot_dnn/
 * config_online.yaml:57-79; dnn/net.py:71-79 feature value [show, click, embed_w, embedx(D-1)];
 * formula per SURVEY.md App. B-13 [EXT]: the rule itself lives in the un-vendored Paddle PS code):
 *   scale = sqrt(initial_g2sum / (initial_g2sum + g2sum));  w = clip(w - lr*g*scale, bounds);
 *   g2sum += mean(g^2) per part (embed_w = first weight, embedx = the other D-1);
 *   show += lookups of the row in this batch, click += sum of their labels (label [B] i64 or NULL).
 * rec: record table [num_rows, row_stride], row = [show | click | g2sum_w | g2sum_x | W(D) | pad];
 * the lookup table of rec_emb_gather / rec_emb_gather_sumpool is W = rec + 4 with the same stride
 * (continuous_value_model(use_cvm=False) = "return the embedding without the CVM columns").  num_slots maps a lookup
 * position to its sample (b = pos / num_slots) for the click counter. */
typedef struct {
  float lr, initial_g2sum, min_bound, max_bound;
} rec_adagrad_hyper;
int rec_sparse_adagrad_rows(int64_t n_max, int32_t emb_dim, int32_t row_stride, int32_t num_slots,
                            const int32_t* n_uniq, const int64_t* uniq_rows,
                            const int32_t* seg_offset, const int32_t* sorted_pos, const float* grad,
                            const rec_grad_layout* grad_layout, const int64_t* label, float* rec,
                            const rec_adagrad_hyper* hyper, void* stream);

/* ------------------------------------------------------------------------------------------
 * The full table accessor of the PS / gpubox mode (slot_dnn/config_online.yaml:57-89: SparseAccessor with
 * embedx_threshold, SparseAdaGradSGDRule for embed_w and embedx, ctr_accessor_param; slot_dnn/net.py:61-62
 * ShowClickEntry; tools/static_gpubox_trainer.py:152-160).  The arithmetic follows the published source of
 * PaddlePaddle release/2.4 [EXT] — paddle/fluid/distributed/ps/table/{sparse_sgd_rule.cc, ctr_accessor.cc,
 * memory_sparse_table.cc} and framework/fleet/heter_ps/optimizer.cuh.h — quoted in oracle/ps_ref.py.
 * A feature value lives in one record row; the layout says where its parts are (floats from the row start), so
 * DeepFM (embedx = the 16-dim embedding at 0, embed_w = the first-order weight behind it) and slot_dnn
 * (W = [embed_w, embedx(8)]) use the same kernels:
 *   stat_off: show, click, embed_g2sum, embedx_g2sum, state, delta_score, unseen_days   (7 consecutive floats;
 *             state 0 = no such key (zeroed memory: reads as embed_w = 0, embedx = 0, what PullSparse returns
 *             for a missing key), 1 = value without its embedx part, 2 = embedx exists)
 * rec_ps_push_rows = MemorySparseTable::PushSparse + CtrCommonAccessor::Update on the touched rows (merge of
 *   duplicate gradients fused in, ascending position order): a new key is created without embedx (embed_w = 0 when
 *   embed_zero_init, else uniform(+-initial_range)); show / click / delta_score / unseen_days; then
 *   SparseAdaGradSGDRule::UpdateValueWork on embed_w and (if it exists) embedx:
 *       float pushed = g * grad_scale;  double scaled = pushed / (show_scale ? pushed_show : 1)   [a FLOAT division, as
 *       `double scaled_grad = grad[i] / scale;` over `const float* grad`, `float scale` is];
 *       w = clip(float(w - lr * scaled * sqrtf(g0 / (g0 + g2sum))));  g2sum = float(g2sum + sum(scaled^2) / n)
 *   a value without embedx drops its embedx gradient and is extended (uniform(+-x_initial_range), a pure function
 *   of (seed,row,element); embedx_g2sum = 0) at the end of the push after which
 *   (show-click)*nonclk_coeff + click*click_coeff >= embedx_threshold (NeedExtendMF on the updated counters).
 *   show / click: per-SAMPLE int64 [B] (NULL: 1 per occurrence / 0); num_slots maps a lookup position to its
 *   sample.
 * rec_ps_shrink_rows = CtrCommonAccessor::Shrink over the table: counters decay, values whose score fell below
 *   delete_threshold or with unseen_days > delete_after_unseen_days are zeroed (the key is gone); n_deleted
 *   (device int64 or NULL) counts them.
 * rec_ps_save_select = Save(value, param) + UpdateStatAfterSave(value, param) for every row: selected[row]
 *   (device bytes or NULL) = 1 if a save of kind param writes the value — 0: all; 1 (delta) / 2 (base): score >=
 *   base_threshold && delta_score >= delta_threshold (0 for base) && unseen_days <= delta_keep_days, delta_score of
 *   the selected values reset; 3: all, and unseen_days += 1 (a day passes).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  float lr, initial_g2sum, min_bound, max_bound, initial_range;           /* embed_sgd_param  (embed_w)  */
  float x_lr, x_initial_g2sum, x_min_bound, x_max_bound, x_initial_range; /* embedx_sgd_param (embedx)   */
  float embedx_threshold, nonclk_coeff, click_coeff;
  float grad_scale;        /* pushed gradient = merged gradient * grad_scale (Paddle pushes the gradient of the SUMMED
                              loss: scale_sparse_gradient_with_batch_size / PushCopy's `* bs`); 1 = as given */
  int32_t show_scale;      /* 1 (Paddle's default): the rule divides the pushed gradient by the pushed show */
  int32_t embed_zero_init; /* 1 (Paddle's default): embed_w of a new value is 0; 0: uniform(+-initial_range) */
  uint64_t seed;
  int64_t row_mul, row_add; /* identity of table row r in the creation values: r * row_mul + row_add (a shard
                               passes {world, rank}: a feature is born with the same values however the table
                               is sharded); {1, 0} for an unsharded table */
} rec_ps_accessor;
typedef struct {
  int32_t row_stride, embed_off, embedx_off, embedx_dim, stat_off;
} rec_ps_layout;
typedef struct {
  const float* grad;       /* element (pos, c) at grad + offset(layout, pos, pitch) + col + c */
  rec_grad_layout layout;
  int32_t pitch;           /* the D of rec_grad_layout's offset formula for this source */
  int32_t col;
} rec_grad_src;
int rec_ps_push_rows(int64_t n_max, int32_t num_slots, const rec_ps_layout* layout, const int32_t* n_uniq,
                     const int64_t* uniq_rows, const int32_t* seg_offset, const int32_t* sorted_pos,
                     const rec_grad_src* grad_embedx, const rec_grad_src* grad_embed, const int64_t* show,
                     const int64_t* click, float* rec, const rec_ps_accessor* accessor, void* stream);
/* host: the creation value of element `element` (0 = embed_w, 1+j = embedx[j]) of feature `row` */
float rec_ps_init_value_host(uint64_t seed, int64_t row, int32_t element, float initial_range);
int rec_ps_shrink_rows(int64_t num_rows, const rec_ps_layout* layout, float* rec, float show_click_decay_rate,
                       float delete_threshold, float delete_after_unseen_days, const rec_ps_accessor* accessor,
                       int64_t* n_deleted, void* stream);
int rec_ps_save_select(int64_t num_rows, const rec_ps_layout* layout, float* rec, int32_t param,
                       float base_threshold, float delta_threshold, float delta_keep_days,
                       const rec_ps_accessor* accessor, uint8_t* selected, int64_t* n_selected, void* stream);

/* ------------------------------------------------------------------------------------------
 * Backward of a ONE-logit head Linear(n -> 1) behind a ReLU (deepfm/net.py:169-174 and the sibling CTR MLPs), in one
 * HBM-bound pass over the [batch, n] activation instead of a K = 1 dX GEMM plus a three-launch skinny dW path:
 *   dx[b,j] = (relu == 0 || act[b,j] > 0) ? dz[b] * w[j] : 0;   dw[j] = sum_b act[b,j] * dz[b];   db[0] = sum_b dz[b]
 * Fixed summation order (deterministic).  n % 4 == 0, n <= 512, 16-byte aligned rows.
 * ---------------------------------------------------------------------------------------- */
int rec_mlp_head_bwd_workspace_bytes(int64_t batch, int32_t n, size_t* bytes);
int rec_mlp_head_bwd(int64_t batch, int32_t n, const float* act, int64_t ld_act, const float* dz, const float* w,
                     int32_t relu, float* dx, int64_t ld_dx, float* dw, float* db, void* workspace,
                     size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Train-mode Dropout and L2Decay of the DCN-v2 DNN tower (dcn_v2/net.py:158,164-170,181-183).
 * rec_dropout: out[r,c] = keep ? in[r,c] / (1-p)^nmask : 0 over a [rows, cols] matrix (row strides ld_in / ld_out,
 *   in place allowed).  keep is a pure function of (seed, stream id, r*cols + c) — no stored mask: the backward is the
 *   same call on the gradient.  nmask 2 applies streams a AND b at once (the reference's two dropouts around a ReLU).
 * rec_l2_decay_grad: grad += (coeff / grad_scale[0]) * w — L2Decay appended AFTER global-norm clipping [EXT]; the
 *   Adam kernels multiply the whole gradient by the clipping coefficient, hence the pre-division (grad_scale NULL: 1).
 * ---------------------------------------------------------------------------------------- */
int rec_dropout(int64_t rows, int32_t cols, int64_t ld_in, int64_t ld_out, const float* in, float* out, float p,
                uint64_t seed, uint64_t stream_a, uint64_t stream_b, int32_t nmask, void* stream);
int rec_l2_decay_grad(int64_t n, float* grad, const float* w, float coeff, const float* grad_scale, void* stream);

/* paddle.optimizer.SGD [EXT] (din/dygraph_model.py:64-73): p -= lr * g.  Rows whose gradient is zero do
 * not move, so updating the merged rows of a SelectedRows gradient equals the dense update. */
int rec_sparse_sgd_rows(int64_t n_max, int32_t emb_dim, int32_t row_stride, const int32_t* n_uniq,
                        const int64_t* uniq_rows, const int32_t* seg_offset, const int32_t* sorted_pos,
                        const float* grad, const rec_grad_layout* grad_layout, float* P, float lr,
                        void* stream);
int rec_sgd_dense(int64_t n, float* p, const float* g, float lr, void* stream);
/* The merge and the SGD row update of a SMALL SelectedRows gradient in one launch (n <= 15360 lookups,
 * emb_dim <= 256): at the reference's own batch size (din/config.yaml: batch_size 32, ~5 k lookups per table) the
 * sort-based rec_ids_group + rec_segment_partials + rec_sparse_sgd_rows are 12-13 launches per table whose launch
 * latency, not work, is the step time.  A wave per lookup finds the occurrences of its row with ballots over the id
 * list; the first occurrence's wave adds the gradient rows in ascending position (the SelectedRows merge order)
 * and applies p -= lr * g.  ids [n] i64 rows (padding_idx < 0: none); out-of-range ids are skipped and flagged in
 * status; grad / grad_layout as for rec_sparse_sgd_rows (partials ignored). */
int rec_sparse_sgd_small(int64_t n, int32_t emb_dim, int32_t row_stride, int64_t num_rows, int64_t padding_idx,
                         const int64_t* ids, const float* grad, const rec_grad_layout* grad_layout, float* P,
                         float lr, int32_t* status, void* stream);
/* The same for up to 8 tables in ONE launch (DIN updates seven embedding tables per step, din/dygraph_model.py:64-73:
 * independent of each other, so their merges share the chip instead of running one after the other). */
typedef struct {
  int64_t n;              /* lookups of this table (<= 15360) */
  int32_t emb_dim, row_stride;
  int64_t num_rows, padding_idx;
  const int64_t* ids;     /* [n] rows */
  const float* grad;
  rec_grad_layout grad_layout;
  float* P;
} rec_small_sgd_job;
int rec_sparse_sgd_small_multi(int32_t count, const rec_small_sgd_job* jobs, float lr, int32_t* status, void* stream);
/* The same one-launch merge with the lazy Adam update of BOTH embeddings of a DeepFM record (the arithmetic of
 * rec_sparse_adam_record: W, m, v and W1, m1, v1 of a touched row in one pass) — the reference's bigdata batch size
 * (deepfm/config_bigdata.yaml: 512 x 26 slots = 13312 lookups) replaces grouping sort + two partial passes + record
 * update, 12 launches, by one.  ids [n] (n = B * num_slots, row = id + slot_offset[pos % num_slots] when
 * slot_offset is given); grad: one emb_dim-wide row per position (grad_layout as rec_sparse_adam_record), grad1:
 * the first-order gradient source with its layout (dz with {div = num_slots}).
 * Above 2048 lookups, ONE table (or slot tables of rows <= 16 floats) is merged by ROW BUCKETS: block b owns the rows
 * hashing to bucket b and keeps their (row, position) pairs in ascending position — the same order of additions, the
 * same bits, without comparing every lookup with every other (the reference's one shared table: 98 -> 21 us). */
int rec_sparse_adam_record_small(int64_t n, int32_t num_slots, int32_t emb_dim, int32_t rec_stride,
                                 int32_t state_stride, int32_t v_offset, int64_t num_rows, int64_t padding_idx,
                                 const int64_t* ids, const int64_t* slot_offset, const float* grad,
                                 const rec_grad_layout* grad_layout, const float* grad1,
                                 const rec_grad_layout* grad1_layout, const float* grad_scale, float* rec, float* MV,
                                 const rec_adam_hyper* hyper, int32_t* status, int32_t* scratch, void* stream);
/* scratch: one device int32 of the caller's (may be NULL).  With it ONE block decides for the whole launch whether every
 * id stays inside its slot's span of rows (then a lookup is only compared with the lookups of its own slot); without it
 * every block of the launch reads the whole id list to take that decision itself. */

/* lazy_mode=False Adam on a SelectedRows gradient — the dygraph default (deepfm/dygraph_model.py:61-65,
 * SURVEY.md App. B-3): every one of the num_rows rows is updated, rows absent from the merged gradient with
 * g = 0.  Same arguments as rec_sparse_adam_rows; streams the whole table (6*N*D*4 B per step). */
int rec_adam_rows_all(int64_t num_rows, int32_t emb_dim, int32_t row_stride, int32_t state_stride,
                      const int32_t* n_uniq, const int64_t* uniq_rows, const int32_t* seg_offset,
                      const int32_t* sorted_pos, const float* grad, const rec_grad_layout* grad_layout,
                      const float* grad_scale, float* P, float* M, float* V, const rec_adam_hyper* hyper,
                      void* stream);

/* The same on the record layout of rec_sparse_adam_record (rec = W(D) | W1 | m1 | v1 | pad, MV = m(D) | v(D) at v_offset):
 * both embeddings of every row in ONE pass — 512 B per row where two rec_adam_rows_all passes move 768 (the second
 * one reads and writes the whole record line for W1 / m1 / v1).  Same arithmetic and summation order: bit-identical. */
int rec_adam_record_all(int64_t num_rows, int32_t emb_dim, int32_t rec_stride, int32_t state_stride,
                        int32_t v_offset, const int32_t* n_uniq, const int64_t* uniq_rows,
                        const int32_t* seg_offset, const int32_t* sorted_pos, const float* grad,
                        const rec_grad_layout* grad_layout, const float* grad1,
                        const rec_grad_layout* grad1_layout, const float* grad_scale, float* rec, float* MV,
                        const rec_adam_hyper* hyper, void* stream);

/* dense Adam over a flat buffer (MLP + FM dense weights); grad_scale as above. */
int rec_adam_dense(int64_t n, float* p, float* m, float* v, const float* g, const float* grad_scale,
                   const rec_adam_hyper* hyper, void* stream);

/* paddle.nn.ClipGradByGlobalNorm(clip_norm) [EXT] (dcn_v2/dygraph_model.py:81-88):
 *   global_norm = sqrt(sum over ALL gradients of g^2);  every g *= clip_norm / max(global_norm, clip_norm).
 * rec_sumsq: out[0] (+)= sum x^2 over a dense buffer; rec_sparse_rows_sumsq: the same over the MERGED
 * rows of a SelectedRows gradient (duplicates summed first, as Paddle's merge does); both reduce in a
 * fixed order.  rec_clip_scale turns the total into the device scalar the Adam kernels take. */
int rec_sumsq_workspace_bytes(size_t* bytes);
int rec_sumsq(int64_t n, const float* x, float* out, int32_t accumulate, void* workspace,
              size_t workspace_bytes, void* stream);
int rec_sparse_rows_sumsq(int64_t n_max, int32_t emb_dim, const int32_t* n_uniq,
                          const int32_t* seg_offset, const int32_t* sorted_pos, const float* grad,
                          const rec_grad_layout* grad_layout, float* out, int32_t accumulate,
                          void* workspace, size_t workspace_bytes, void* stream);
int rec_clip_scale(const float* sumsq, float clip_norm, float* scale, void* stream);

/* CrossNetV2 backward glue (dcn_v2/net.py:222-226), one streaming pass:
 *   dU = dX * X0;   dX0_acc = (accumulate ? dX0_acc : 0) + dX * U       (all [m,n] with row strides) */
int rec_cross_bwd_prep(int64_t m, int32_t n, const float* dX, int32_t ld_dx, const float* X0,
                       int32_t ld_x0, const float* U, int32_t ld_u, float* dU, int32_t ld_du,
                       float* dX0_acc, int32_t ld_acc, int32_t accumulate, void* stream);

/* ------------------------------------------------------------------------------------------
 * DIN attention-pool, fused: 4 lookups -> [h, q, h-q, h*q] -> Linear/Sigmoid x2 -> Linear -> + mask ->
 * * E^-0.5 -> softmax over T -> weights @ h.   Replaces din/net.py:141-173; the [B,T,4E] concat and the
 * hidden activations never reach HBM (one block per sample, 32-position tiles, online softmax).
 *   hist_item/hist_cat/tgt_item_seq/tgt_cat_seq [B,T] i64 (no padding_idx: id 0 is a real row, App. B-11)
 *   mask [B,T] i64: 0 valid, -1e9 padding (din/dinReader.py:81-84,99)
 *   tables f32 [rows, stride]; hist and target tables are independent parameters (App. C)
 *   att_w1 [4E,H1] (Paddle [in,out]), att_b1 [H1], att_w2 [H1,H2], att_b2 [H2], att_w3 [H2], att_b3 [1]
 *   out [B,E];  att_weight [B,T] or NULL: the softmax weights (saved for the backward)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int64_t batch;      /* B */
  int32_t max_len;    /* T (padded history length of this batch) */
  int32_t item_dim, cat_dim;   /* E = item_dim + cat_dim <= 256, both multiples of 4 */
  int32_t hidden1, hidden2;    /* 80, 40 */
  int64_t item_rows, cat_rows;
  int32_t item_stride, cat_stride;
} rec_din_desc;

/* act1 [B,T,H1] or NULL: layer-1 activations for the backward.  Written only when rec_din_saves_act1(desc)
 * returns 1 (the reference net's own shape: E 128, attention MLP 80-40-1); otherwise left untouched and the
 * backward must be given NULL for act1_saved. */
int rec_din_saves_act1(const rec_din_desc* desc);
int rec_din_attention_pool_fwd(const rec_din_desc* desc, const int64_t* hist_item,
                               const int64_t* hist_cat, const int64_t* tgt_item_seq,
                               const int64_t* tgt_cat_seq, const int64_t* mask,
                               const float* w_hist_item, const float* w_hist_cat,
                               const float* w_tgt_item_seq, const float* w_tgt_cat_seq,
                               const float* att_w1, const float* att_b1, const float* att_w2,
                               const float* att_b2, const float* att_w3, const float* att_b3,
                               float* out, float* att_weight, float* act1, int32_t* status, void* stream);
/* The same forward with a caller-owned workspace (rec_din_attention_pool_fwd_workspace_bytes; 0 bytes = the shape
 * never needs one).  With it, a batch of FEW samples (the reference's din/config.yaml batch size 32: fewer samples than
 * the chip has block slots) is computed one block per 32-position history tile — each tile softmax-normalised on its
 * own, a second launch rescales the pieces to the softmax over the whole history — instead of one block per sample
 * walking its tiles one after the other (82 -> ~25 us at 32 x 152).  Same results up to fp32 rounding of the
 * softmax's normalisation order. */
int rec_din_attention_pool_fwd_workspace_bytes(const rec_din_desc* desc, size_t* bytes);
int rec_din_attention_pool_fwd_ws(const rec_din_desc* desc, const int64_t* hist_item,
                                  const int64_t* hist_cat, const int64_t* tgt_item_seq,
                                  const int64_t* tgt_cat_seq, const int64_t* mask,
                                  const float* w_hist_item, const float* w_hist_cat,
                                  const float* w_tgt_item_seq, const float* w_tgt_cat_seq,
                                  const float* att_w1, const float* att_b1, const float* att_w2,
                                  const float* att_b2, const float* att_w3, const float* att_b3,
                                  float* out, float* att_weight, float* act1, int32_t* status, void* workspace,
                                  size_t workspace_bytes, void* stream);

/* Backward of the block above w.r.t. the gathered rows (what loss.backward() computes for net.py:141-173):
 *   d_hist [B,T,E] = gradient of [hist_item_emb | hist_cat_emb] per position, d_tgt_seq [B,T,E] likewise
 *   for the target-seq tables; their row-wise merge (rec_ids_group + rec_sparse_sgd_rows) is the
 *   embedding gradient.  att_weight = the forward's softmax weights; att_w1_t = att_w1 transposed
 *   [H1,4E].  out_saved [B,E] / act1_saved [B,T,H1] (both nullable): the forward's output and the layer-1
 *   activations it wrote when its `act1` argument was non-null and rec_din_saves_act1(desc) == 1 (NULL here
 *   otherwise).  With both, the backward runs on the saved activations (dz1 W1^T on the matrix cores, W1^T in
 *   registers); without, hidden activations are recomputed.  The attention MLP's own weight gradients are not
 *   produced (not registered parameters in dygraph mode). */
int rec_din_attention_pool_bwd(const rec_din_desc* desc, const int64_t* hist_item,
                               const int64_t* hist_cat, const int64_t* tgt_item_seq,
                               const int64_t* tgt_cat_seq, const float* w_hist_item,
                               const float* w_hist_cat, const float* w_tgt_item_seq,
                               const float* w_tgt_cat_seq, const float* att_w1,
                               const float* att_w1_t, const float* att_b1, const float* att_w2,
                               const float* att_b2, const float* att_w3, const float* att_weight,
                               const float* out_saved, const float* act1_saved,
                               const float* d_out, float* d_hist, float* d_tgt_seq, void* stream);

/* The same backward with a caller-owned workspace (rec_din_attention_pool_bwd_workspace_bytes; 0 bytes = the shape never
 * needs one): at batches larger than the resident grid the blocks of the saved-activation kernel draw their next SAMPLE
 * from a ticket counter in it instead of walking fixed ranges — samples cost between one and max_len / 32 tiles once
 * tiles whose saved weights are all zero are skipped.  Same results: which block computes a tile changes nothing. */
int rec_din_attention_pool_bwd_workspace_bytes(const rec_din_desc* desc, size_t* bytes);
int rec_din_attention_pool_bwd_ws(const rec_din_desc* desc, const int64_t* hist_item,
                                  const int64_t* hist_cat, const int64_t* tgt_item_seq,
                                  const int64_t* tgt_cat_seq, const float* w_hist_item,
                                  const float* w_hist_cat, const float* w_tgt_item_seq,
                                  const float* w_tgt_cat_seq, const float* att_w1,
                                  const float* att_w1_t, const float* att_b1, const float* att_w2,
                                  const float* att_b2, const float* att_w3, const float* att_weight,
                                  const float* out_saved, const float* act1_saved,
                                  const float* d_out, float* d_hist, float* d_tgt_seq, void* workspace,
                                  size_t workspace_bytes, void* stream);

/* CrossNetMix backward glue for one expert (dcn_v2/net.py:301-317), one pass over [m,n]:
 *   dU = dX*X0*p_e;  dX0_acc (+)= dX*p_e*U;  dp[i] = sum_j dX*X0*U      (p_e = prob[i*prob_stride])
 * and the softmax backward of the gate: dz = p * (dp - sum_e p_e dp_e) over n <= 64 columns. */
int rec_moe_bwd_prep(int64_t m, int32_t n, const float* dX, int32_t ld_dx, const float* X0,
                     int32_t ld_x0, const float* U, int32_t ld_u, const float* prob, int32_t prob_stride,
                     float* dU, int32_t ld_du, float* dX0_acc, int32_t ld_acc, int32_t accumulate,
                     float* dp, int32_t dp_stride, void* stream);
int rec_softmax_rows_bwd(int64_t m, int32_t n, const float* p, int32_t ldp, const float* dp,
                         int32_t lddp, float* dz, int32_t lddz, void* stream);

/* y[i,:] = softmax(x[i,:]) over n <= 64 columns (CrossNetMix expert gate, dcn_v2/net.py:313-316). */
int rec_softmax_rows(int64_t m, int32_t n, const float* x, int32_t ldx, float* y, int32_t ldy,
                     void* stream);

/* ------------------------------------------------------------------------------------------
 * xDeepFM Compressed Interaction Network (models/rank/xdeepfm/net.py:155-202), the passes around the GEMM.
 * Every X_k (k >= 1) is kept d-major, XT[(b,d), c] (row = b*D + d), so the reference's 1x1 Conv2D over the F*S
 * interaction channels (net.py:176-190) is one row-major GEMM  XT_{k+1} = Z @ Wc^T  with M = B*D, K = F*S, N = C
 * (rec_gemm_f32; Wc^T is the conv weight [C, F*S, 1, 1] viewed [C, F*S], passed with trans_b).
 *   rec_cin_view: element (b, j, d) of a feature tensor sits at base + b*stride_b + j*stride_j + d*stride_d
 *                 floats — feat_embeddings [B,F,D] is {F*D, D, 1}; an XT [B*D, ld] is {D*ld, 1, ld}.
 *   rec_cin_outer_fwd : Z[(b,d), f*S + s] = X0[b,f,d] * Xk[b,s,d]                    net.py:163-175
 *   rec_cin_outer_bwd : dX0[b,f,d] (+)= sum_s dZ[(b,d), f*S+s] Xk[b,s,d]
 *                       dXk[b,s,d] (+)= sum_f dZ[(b,d), f*S+s] X0[b,f,d]  (+ dpool[b,s]: the gradient of the
 *                       sum-pooled features of that layer, every d row gets it).  dX0 and dXk may alias
 *                       (layer 1, Xk = X0: pass both accumulate flags).
 *   rec_cin_sumpool(_bwd): pooled[b,c] = sum_d XT[(b,d), c]  (net.py:195-198) and its broadcast backward.
 * Z for a whole batch is B*D*F*S floats (11.8 GB at B 65536, D 9, 39 x 128): callers walk the batch in chunks.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int64_t stride_b;
  int32_t stride_j, stride_d;
} rec_cin_view;

int rec_cin_outer_fwd(int64_t batch, int32_t emb_dim, int32_t F, int32_t S, const float* X0,
                      const rec_cin_view* v0, const float* Xk, const rec_cin_view* vk, float* Z, int64_t ldz,
                      void* stream);
int rec_cin_outer_bwd(int64_t batch, int32_t emb_dim, int32_t F, int32_t S, const float* dZ, int64_t ldz,
                      const float* X0, const rec_cin_view* v0, const float* Xk, const rec_cin_view* vk,
                      float* dX0, const rec_cin_view* dv0, int32_t accumulate_dx0, float* dXk,
                      const rec_cin_view* dvk, int32_t accumulate_dxk, const float* dpool, int64_t ld_dpool,
                      void* stream);
/* The other association of a layer, chosen when C < S (then Y is smaller than Z and its GEMM better shaped):
 *   Y = Xk @ W'^T  with W' = the same conv weight viewed [C*F, S]   (rec_gemm_f32, trans_b; M = B*D, K = S, N = C*F)
 *   rec_cin_contract_fwd : XT_{k+1}[(b,d), c] = sum_f X0[b,f,d] * Y[(b,d), c*F + f]
 *   rec_cin_contract_bwd : dY[(b,d), c*F+f] = dXT[(b,d), c] * X0[b,f,d];  dX0[b,f,d] (+)= sum_c dXT[(b,d),c] Y[(b,d), c*F+f]
 *   (then dW' = dY^T @ Xk and dXk = dY @ W' are plain GEMMs). */
int rec_cin_contract_fwd(int64_t batch, int32_t emb_dim, int32_t F, int32_t C, const float* Y, int64_t ldy,
                         const float* X0, const rec_cin_view* v0, float* XT, int64_t ldx, void* stream);
int rec_cin_contract_bwd(int64_t batch, int32_t emb_dim, int32_t F, int32_t C, const float* Y, int64_t ldy,
                         const float* dXT, int64_t ldx, const float* X0, const rec_cin_view* v0, float* dY,
                         int64_t lddy, float* dX0, const rec_cin_view* dv0, int32_t accumulate_dx0, void* stream);
int rec_cin_sumpool(int64_t batch, int32_t emb_dim, int32_t C, const float* XT, int64_t ldx, float* out,
                    int64_t ldo, void* stream);
int rec_cin_sumpool_bwd(int64_t batch, int32_t emb_dim, int32_t C, const float* dpool, int64_t ldp,
                        float* dXT, int64_t ldx, void* stream);

/* ------------------------------------------------------------------------------------------
 * DLRM (models/rank/dlrm/net.py): BatchNorm1D over the batch, the pairwise dot interaction, accuracy counts.
 *   rec_batchnorm_fwd : nn.BatchNorm1D on X [m, n] (net.py:151-153, after every Linear -> ReLU).  training != 0:
 *                       batch mean and BIASED batch variance, running = momentum*running + (1-momentum)*batch
 *                       (Paddle: momentum 0.9, epsilon 1e-5) [EXT]; training == 0: the running statistics.
 *                       save_mean / save_invstd [n] are kept for the backward.  Column sums in a fixed order.
 *   rec_batchnorm_bwd : dX, dgamma [n], dbeta [n];  relu_mask != 0: X was a ReLU output — dX is zeroed where
 *                       X <= 0 (the ReLU backward folded in), so dX is the gradient of the Linear's output.
 *   rec_dot_interact_fwd : T [batch, F, D] (sample stride ldt) = [emb_1 .. emb_{F-1}, x] ->
 *                       R [batch, D + F(F-1)/2] = [ x | <T_i, T_j> for i < j in row-major order ]   (net.py:96-123:
 *                       bmm + triu + masked_select + concat)
 *   rec_dot_interact_bwd : dT[b,i,:] = sum_{j != i} dZ(i,j) T[b,j,:]  (+ dR[b,:D] for the last field)
 *   rec_accuracy_count : paddle.metric.Accuracy top-1 of softmax(raw) for two classes: counts[0] += #(pred > 0.5
 *                       == label != 0), counts[1] += n   (int64, exact)          dygraph_model.py:58-63,79-81
 * ---------------------------------------------------------------------------------------- */
int rec_batchnorm_workspace_bytes(int64_t m, int32_t n, size_t* bytes);
int rec_batchnorm_fwd(int64_t m, int32_t n, const float* X, int64_t ldx, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, float momentum, float eps, int32_t training,
                      float* Y, int64_t ldy, float* save_mean, float* save_invstd, void* workspace,
                      size_t workspace_bytes, void* stream);
int rec_batchnorm_bwd(int64_t m, int32_t n, const float* X, int64_t ldx, const float* dY, int64_t lddy,
                      const float* gamma, const float* save_mean, const float* save_invstd, int32_t relu_mask,
                      float* dX, int64_t lddx, float* dgamma, float* dbeta, void* workspace,
                      size_t workspace_bytes, void* stream);
int rec_dot_interact_fwd(int64_t batch, int32_t F, int32_t D, const float* T, int64_t ldt, float* R, int64_t ldr,
                         void* stream);
int rec_dot_interact_bwd(int64_t batch, int32_t F, int32_t D, const float* T, int64_t ldt, const float* dR,
                         int64_t ldr, float* dT, int64_t lddt, void* stream);
int rec_accuracy_count(int64_t n, const float* pred, const int64_t* label, int64_t* counts, void* stream);

/* ------------------------------------------------------------------------------------------
 * f32 GEMM on the matrix cores with fused epilogues: C[M,N] = epi(op(A)[M,K] @ op(B)[K,N]).
 * Replaces paddle.nn.Linear / paddle.matmul (+ the elementwise ops around them) on the hot path:
 *   top MLP fwd/bwd  deepfm/net.py:142-174, dcn_v2/net.py:140-184, din/net.py:104-137,175-181
 *   CrossNetV2       dcn_v2/net.py:214-226 : REC_EPI_CROSS computes X_l + X_0*(X_l W + b) in place
 *   CrossNetMix      dcn_v2/net.py:278-320 : REC_EPI_BIAS_TANH for the low-rank projections
 * Row-major everywhere; Paddle's Linear.weight is [in,out] = B[K,N].
 *   trans_a: A is stored [K,M] (dW = X^T G);  trans_b: B is stored [N,K] (dX = G W^T).
 * split_k: 0 = automatic, partial sums are reduced in a fixed order (deterministic).
 *
 * ARITHMETIC — two kernel families, chosen per call:
 *   (a) exact f32: v_mfma_f32_16x16x4_f32 = a k-ordered fmaf chain per output element.  Every call with M < 8192 rows, every
 *       shape (b) does not cover, and EVERY call when the process environment holds REC_GEMM_BF16X3=0.
 *   (b) bf16 x 3 (the DEFAULT for the tall GEMMs of the towers; csrc/gemm_bf16x3.h): each f32 operand is split into three
 *       bf16 terms (8 + 8 + 8 significand bits, round-to-nearest remainders) and the six term products with i + j <= 2
 *       are accumulated in f32 by v_mfma_f32_16x16x32_bf16.  Taken by   forward / dX calls (trans_a = 0): M >= 8192,
 *       N % 4 == 0, N within 15 % of a column-block multiple (336 <= N: 400, 432, 512, 768, 1560 ...), K % 8 == 0, K >= 64,
 *       16-byte aligned rows, split_k <= 1, no b_colsum;   weight-gradient calls (trans_a = 1, no epilogue): K >= 8192,
 *       336 <= M <= 448, 336 <= N <= 416 (REC_GEMM_BF16X3_DW=0: never).  Results are f32-GRADE — the error against
 *       float64 is that of (a), <= 4e-7 of sum |a||b| (tests/test_gemm_gpu.py) — but NOT the bit pattern of (a):
 *         - a step sharded so that a rank holds fewer than 8192 rows runs (a) and does not match the unsharded step's
 *           bits (it matches to the bound above);
 *         - non-finite operands: NaN propagates in both; an operand that is +-Inf, or finite with |x| > 3.39e38 (it
 *           rounds to the bf16 Inf), leaves Inf - Inf in its second term, so (b) returns NaN where (a) returns +-Inf —
 *           also behind REC_EPI_RELU_MASK / BIAS_RELU, where relu(-Inf) = 0 under (a) continues and NaN under (b) does
 *           not; operands whose third term falls below 2^-133 lose it (bf16 subnormals flush): the product of such an
 *           operand carries 16 instead of 24 significand bits, i.e. an absolute error below 2^-126 * |b| — not
 *           observable behind an f32 accumulation of normal numbers (tests/test_gemm_gpu.py::test_gemm_nonfinite_*).
 *       REC_GEMM_BF16X3 (read per call): unset = (b) as above; 1 = (b), but a weight-gradient call with an explicit
 *       split_k stays on (a) (the caller asked for few long blocks beside another kernel); 0 = (a) everywhere.
 *   WORKSPACE: under (b) a forward / dX call WRITES op(B)'s three-plane image into the workspace (unless
 *   rec_gemm_epilogue_args.b_image hands one in) and a weight-gradient call its partial tiles — so, unlike (a) without
 *   split-K, concurrent calls on two streams need one workspace each.  rec_gemm_f32_workspace_bytes covers both families.
 * ---------------------------------------------------------------------------------------- */
typedef enum {
  REC_EPI_NONE = 0,
  REC_EPI_BIAS = 1,         /* acc + bias[j] */
  REC_EPI_BIAS_RELU = 2,    /* max(acc + bias[j], 0) */
  REC_EPI_RELU_MASK = 3,    /* aux0[i,j] > 0 ? acc : 0          (ReLU backward on dX) */
  REC_EPI_CROSS = 4,        /* aux1[i,j] + aux0[i,j] * (acc + bias[j])   (aux0 = X_0, aux1 = X_l) */
  REC_EPI_BIAS_SIGMOID = 5, /* sigmoid(acc + bias[j]) */
  REC_EPI_BIAS_TANH = 6,    /* tanh(acc + bias[j]); bias may be NULL */
  REC_EPI_ADD = 7,          /* acc + aux1[i,j] (+ bias[j], + aux0[i,j] when given) */
  REC_EPI_DTANH = 10,       /* acc * (1 - aux0[i,j]^2)             (tanh backward on dX, CrossNetMix) */
  REC_EPI_DSIGMOID = 9,     /* acc * aux0[i,j] * (1 - aux0[i,j])   (sigmoid backward on dX, din/net.py MLPs) */
  REC_EPI_MOE = 8           /* aux1[i,j] + aux0[i,j] * row_scale[i] * (acc + bias[j])
                               (CrossNetMix, dcn_v2/net.py:301-317: aux0 = x_0, aux1 = running x_{l+1},
                               row_scale = softmax gate of this expert) */
} rec_epilogue;

typedef struct {
  int64_t m;
  int32_t n, k;
  int32_t lda, ldb, ldc; /* leading dimensions in floats, of the arrays as stored */
  int32_t trans_a, trans_b;
  int32_t epilogue;      /* rec_epilogue */
  int32_t split_k;       /* 0 = automatic */
  int32_t num_cus;       /* 0 = the whole chip; > 0: the stream the call is issued on is confined to this many compute
                            units (rec_stream_create_cu_range): split-K and the bf16 x 3 weight-gradient grid are sized
                            for one resident round on THOSE, not on the chip */
} rec_gemm_desc;

typedef struct {
  const float* bias;        /* [N] */
  const float* aux0;        /* [M, ld_aux0] */
  int32_t ld_aux0;
  const float* aux1;        /* [M, ld_aux1] */
  int32_t ld_aux1;
  const float* row_scale;   /* [M] with stride row_scale_stride (REC_EPI_MOE) */
  int32_t row_scale_stride;
  float* out2;              /* [M, ld_out2] or NULL: REC_EPI_CROSS also stores u = acc + bias (for backward) */
  int32_t ld_out2;
  float* b_colsum;          /* [N] or NULL (trans_a calls only): also return the column sums of B over K — the
                               bias gradient when the call computes dW = X^T dY (B = dY), at no extra pass */
  const void* b_image;      /* NULL, or the bf16 x 3 image of THIS call's op(B) made by rec_gemm_b_images (same k, n,
                               trans_b, current values of B): the call then skips its own split launch and does not
                               write the workspace.  Ignored by calls that do not take the bf16 x 3 forward / dX kernel */
  void* relu_bits;          /* NULL, or rec_gemm_relu_bits_bytes(desc) bytes of device memory (8-byte aligned): the ReLU
                               mask of an activation as BITS.  A REC_EPI_BIAS_RELU call WRITES it beside C (one 64-bit word
                               per row, column block and lane group of the bf16 x 3 kernel); a REC_EPI_RELU_MASK call with
                               the same m and n READS it instead of aux0 (which may then be NULL): the dX GEMM of a tower
                               fetches 4 MB of bits where it fetched the 105 MB activation.  Same results bit for bit.
                               Only calls for which rec_gemm_relu_bits_bytes reports eligible = 1 (both sides take the
                               bf16 x 3 forward / dX kernel); any other call given relu_bits fails with REC_EINVAL */
} rec_gemm_epilogue_args;

int rec_gemm_f32_workspace_bytes(const rec_gemm_desc* desc, size_t* bytes);
/* Host query: *eligible = 1 and the size of rec_gemm_epilogue_args.relu_bits for this call (REC_EPI_BIAS_RELU or
 * REC_EPI_RELU_MASK on the bf16 x 3 forward / dX kernel), else 0 / 0.  The layout depends on m and n only: the mask a
 * forward call with (m, n) wrote is the one a dX call with the same (m, n) reads
 * (deepfm/net.py:142-174: Linear -> ReLU, and the ReLU' of its backward). */
int rec_gemm_relu_bits_bytes(const rec_gemm_desc* desc, int32_t* eligible, size_t* bytes);
/* Host query: the split-K factor the planner picks for `desc` (its split_k ignored) when the GEMM may use
 * num_cus compute units (<= 0: the whole chip) — one resident round of blocks.  A caller that runs the GEMM on
 * a CU-restricted stream (rec_stream_create_cu_range) passes the result as desc->split_k. */
int rec_gemm_plan_splits(const rec_gemm_desc* desc, int32_t num_cus, int32_t* splits);
int rec_gemm_f32(const rec_gemm_desc* desc, const float* A, const float* B, float* C,
                 const rec_gemm_epilogue_args* args /* may be NULL */, void* workspace,
                 size_t workspace_bytes, void* stream);
/* Two INDEPENDENT GEMMs — neither reads what the other writes — in one call: the backward of one Linear,
 * dW = X^T G (+ db = colsum(G): desc0, trans_a, no epilogue) and dX = G W^T (+ ReLU' of the layer in front: desc1,
 * trans_b, REC_EPI_NONE / REC_EPI_RELU_MASK / REC_EPI_DSIGMOID), both consume G and nothing of each other (ops.mlp_backward; the
 * reference's Linear backward is two matmul_grad kernels, `paddle.nn.Linear` in deepfm/net.py:142-174).  At the
 * reference's own batch sizes both are launch-bound (M N K < 1.5e8, K <= 1024: csrc/gemm_direct.h) and go out as ONE
 * launch whose workgroups run exactly the code of the single launches — bit-identical to two rec_gemm_f32 calls, one
 * ~4 us launch less; any other pair IS two rec_gemm_f32 calls, desc0 first (workspace: the larger of the two needs).
 * REC_GEMM_PAIR=0: always two calls. */
int rec_gemm_f32_pair(const rec_gemm_desc* desc0, const float* A0, const float* B0, float* C0,
                      const rec_gemm_epilogue_args* args0 /* may be NULL */, const rec_gemm_desc* desc1, const float* A1,
                      const float* B1, float* C1, const rec_gemm_epilogue_args* args1 /* may be NULL */, void* workspace,
                      size_t workspace_bytes, void* stream);
/* Weight images ahead of time.  Under the bf16 x 3 family every forward / dX call splits op(B) [k, n] into its plane image
 * first (a ~5 us launch per call); the weights of a tower change once per step, so a trainer makes ALL images of the step
 * in one launch right after its optimizer (rec_adam_dense) and passes them as rec_gemm_epilogue_args.b_image.
 *   rec_gemm_b_image_bytes : *eligible = 1 and the image size if an op(B) of k x n has an image form, else 0 / 0
 *   rec_gemm_b_images      : items[i].image <- image of op(B_i) (B_i as rec_gemm_f32 takes it: [k,n] row-major with ldb,
 *                            or [n,k] when trans_b); any count (8 images per launch).  items is a HOST array. */
typedef struct {
  const float* B;
  int32_t ldb, k, n, trans_b;
  void* image;     /* device, 16-byte aligned, rec_gemm_b_image_bytes(k, n) bytes */
} rec_gemm_b_image;
int rec_gemm_b_image_bytes(int32_t k, int32_t n, int32_t* eligible, size_t* bytes);
int rec_gemm_b_images(int32_t count, const rec_gemm_b_image* items, void* stream);

/* out[j] = sum_i G[i,j] (bias gradient of a Linear), fixed reduction order. */
int rec_colsum_workspace_bytes(int64_t m, int32_t n, size_t* bytes);
int rec_colsum(int64_t m, int32_t n, int32_t ld, const float* G, float* out, void* workspace,
               size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * CrossNet layers as ONE entry point per layer and direction (SURVEY.md §8(b) `crossnet_v2_layer`,
 * `crossnet_mix_layer`): what a Paddle custom op (PD_BUILD_OP / PD_BUILD_GRAD_OP) binds for
 *   dcn_v2/net.py:214-226  CrossNetV2:  x_{l+1} = x_l + x_0 * (x_l W_l + b_l)
 *   dcn_v2/net.py:278-320  CrossNetMix: x_{l+1} = x_l + sum_e p_e * x_0 * (tanh(tanh(x_l V_e) C_e^T) U_e^T + b_l),
 *                                       p = softmax_e(x_l gate_w + gate_b)
 * They launch the GEMMs (rec_gemm_f32 epilogues) and streaming glue passes declared above in a fixed order; no
 * kernel of their own.  Row strides (ld_*) in floats, 0 = d.  Workspace: query the _workspace_bytes function
 * (GEMM split-K partials + the layer's scratch tensors).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int64_t batch;
  int32_t d;                            /* feature width: num_field * emb_dim */
  int32_t ld_x0, ld_xl, ld_out, ld_u;   /* X_0, X_l, X_{l+1}, saved U_l */
} rec_crossnet_v2_desc;
int rec_crossnet_v2_layer_workspace_bytes(const rec_crossnet_v2_desc* desc, size_t* fwd_bytes, size_t* bwd_bytes);
/* W [d,d] (Paddle [in,out]), bias [d]; U_saved [B, ld_u] (or NULL) receives u = x_l W + b for the backward. */
int rec_crossnet_v2_layer_fwd(const rec_crossnet_v2_desc* desc, const float* X0, const float* Xl, const float* W,
                              const float* bias, float* Xnext, float* U_saved, void* workspace,
                              size_t workspace_bytes, void* stream);
/* Given dXnext = d x_{l+1}:  dW [d,d], db [d], dXl = d x_l (may alias dXnext);  dX0_acc (+)= dXnext * U_l
 * (accumulate_dx0 = 0: overwritten);  fold_dx0 != 0: dXl also receives dX0_acc (the first layer, x_l = x_0). */
int rec_crossnet_v2_layer_bwd(const rec_crossnet_v2_desc* desc, const float* X0, const float* Xl, const float* W,
                              const float* U_saved, const float* dXnext, int32_t ld_dxnext, float* dX0_acc,
                              int32_t ld_acc, int32_t accumulate_dx0, int32_t fold_dx0, float* dXl,
                              int32_t ld_dxl, float* dW, float* db, void* workspace, size_t workspace_bytes,
                              void* stream);

typedef struct {
  int64_t batch;
  int32_t d, rank, experts;             /* experts <= 64 */
  int32_t ld_x0, ld_xl, ld_out;
} rec_crossnet_mix_desc;
int rec_crossnet_mix_layer_workspace_bytes(const rec_crossnet_mix_desc* desc, size_t* fwd_bytes, size_t* bwd_bytes);
/* U, V [E,d,r], C [E,r,r], bias [d], gate_w [d,E] (the E Linear(d,1) stacked), gate_b [E].
 * Saved for the backward: t1 = tanh(x_l V_e), t2 = tanh(t1 C_e^T), both [B, E*r]; prob [B,E]. */
int rec_crossnet_mix_layer_fwd(const rec_crossnet_mix_desc* desc, const float* X0, const float* Xl, const float* U,
                               const float* V, const float* C, const float* bias, const float* gate_w,
                               const float* gate_b, float* Xnext, float* t1, float* t2, float* prob,
                               void* workspace, size_t workspace_bytes, void* stream);
/* gU, gV [E,d,r], gC [E,r,r], gbias [d] are written; the gating layers are shared by all cross layers
 * (net.py:267-268): g_gate_w [d,E] / g_gate_b [E] are overwritten when accumulate_gate == 0, added to otherwise.
 * dXl must not alias dXnext.  dX0_acc / fold_dx0 as above. */
int rec_crossnet_mix_layer_bwd(const rec_crossnet_mix_desc* desc, const float* X0, const float* Xl, const float* U,
                               const float* V, const float* C, const float* bias, const float* gate_w,
                               const float* t1, const float* t2, const float* prob, const float* dXnext,
                               int32_t ld_dxnext, float* dX0_acc, int32_t ld_acc, int32_t accumulate_dx0,
                               int32_t fold_dx0, float* dXl, int32_t ld_dxl, float* gU, float* gV, float* gC,
                               float* gbias, float* g_gate_w, float* g_gate_b, int32_t accumulate_gate,
                               void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Loss head: predict = sigmoid(y1+y2+y_dnn) (deepfm/net.py:47);
 *            cost = log_loss(pred,label,eps=1e-4); avg = mean(cost) (deepfm/dygraph_model.py:53-58)
 * and its gradient dz = d avg / d z.  loss_out[0] = avg (reduced in a fixed order).
 * y2 / y_dnn may be NULL (treated as 0).  workspace >= rec_logloss_workspace_bytes(B).
 * mean_over = denominator of the mean (0 -> batch).  Data-parallel ranks pass the GLOBAL batch so
 * that summing gradients over ranks reproduces one step on the concatenated batch.
 * clip_lo < clip_hi: the logit goes through paddle.clip(z, clip_lo, clip_hi) first (slot_dnn/net.py:84,
 * `F.sigmoid(paddle.clip(y_dnn, min=-15.0, max=15.0))`), dz = 0 outside the open interval; clip_lo >= clip_hi: off.
 * ---------------------------------------------------------------------------------------- */
int rec_logloss_workspace_bytes(int64_t batch, size_t* bytes);
int rec_sigmoid_logloss(int64_t batch, int64_t mean_over, const float* y1, const float* y2,
                        const float* y_dnn,
                        const int64_t* label, float eps, float clip_lo, float clip_hi, float* pred,
                        float* dz, float* loss_out, void* workspace, size_t workspace_bytes, void* stream);

/* binary_cross_entropy_with_logits(reduction='mean') [EXT] (din/dygraph_model.py:58-61): loss_out[0] = mean,
 * pred = sigmoid(logit), dz = (pred - label) / mean_over (0 -> batch).  label is float32 as the DIN reader
 * feeds it.  workspace >= rec_logloss_workspace_bytes(batch). */
int rec_bce_with_logits(int64_t batch, int64_t mean_over, const float* logit, const float* label,
                        float* pred, float* dz, float* loss_out, void* workspace, size_t workspace_bytes,
                        void* stream);

/* paddle.metric.Auc("ROC").update [EXT] (deepfm/dygraph_model.py:69-73,83-84):
 *   bucket = int(pred*num_thresholds); stat_pos[bucket] += label!=0; stat_neg[bucket] += label==0.
 * stat_pos/stat_neg are i64 [num_thresholds+1], accumulated (not cleared). */
int rec_auc_histogram(int64_t batch, const float* pred, const int64_t* label,
                      int32_t num_thresholds, int64_t* stat_pos, int64_t* stat_neg, void* stream);

/* ------------------------------------------------------------------------------------------
 * Row-sharded tables: owner(r) = r mod G, local row = r div G.
 * Stands in for the key-sharded pull/push of core.PSGPU inside exe.train_from_dataset
 * (tools/static_gpubox_trainer.py:152-160,256; models/rank/dnn/net.py:71-79) [HeterPS is EXT].
 * One stable partition of the n = B*S lookups by owner:
 *   send_local_row [n] i64 : local rows grouped by owner (ascending position inside a group) —
 *                            the message of the ids all-to-all; first sum(send_counts[0..G)) valid
 *   send_pos       [n] i64 : position b*S+s of each entry (row-grad gather for the bwd exchange)
 *   send_sample    [n] i64 : b of each entry (dy1 gather for the first-order table)
 *   slot_of_pos    [n] i64 : 1 + index of a position in send order, 0 for padding — the reply of
 *                            the rows all-to-all arrives in send order, so this is the `ids` that
 *                            rec_deepfm_fm_fwd reads the reply buffer with (row 0 = zero row)
 *   send_counts  [G+1] i64 : entries per owner; [G] = dropped (padding / out-of-range) positions
 * All outputs are bit-exact targets.
 * ---------------------------------------------------------------------------------------- */
int rec_shard_route_workspace_bytes(int64_t n, int32_t num_shards, size_t* bytes);
int rec_shard_route(int64_t n, int32_t num_slots, int64_t num_rows, int64_t padding_idx,
                    int32_t num_shards, const int64_t* ids, const int64_t* slot_offset,
                    int64_t* send_local_row, int64_t* send_pos, int64_t* send_sample,
                    int64_t* slot_of_pos, int64_t* send_counts, int32_t* status, void* workspace,
                    size_t workspace_bytes, void* stream);

/* Deduplicated lookup plan: the DISTINCT rows a rank needs of every owner, in fixed-capacity send slots (owner o owns
 * slots [o * cap, (o + 1) * cap)) — what HeterPS does per pass before it builds the per-GPU tables
 * (tools/static_gpubox_trainer.py:237-246 load_into_memory -> PSGPU.begin_pass [EXT]).  rec_ids_group over the shard-major
 * key (row % G) * local_rows + row / G, then four small kernels; no host read, sizes are the capacity:
 *   sorted_pos / uniq_rows / seg_offset / n_uniq : the grouping (uniq_rows = shard-major keys), as rec_ids_group emits it —
 *                  the merge keys of rec_sparse_sgd_rows & co. for the local pre-merge of the row gradients
 *   send_rows    [G * cap] i64 : local row per slot; empty slots hold `local_rows` (the owner groups them away by
 *                  rec_ids_group(num_rows = local_rows + 1, padding_idx = local_rows))
 *   slot_of_pos  [n] i64 : 1 + slot of a position's row, 0 = padding / out of range / behind the capacity — the `ids`
 *                  rec_deepfm_fm_fwd reads the (G * cap + 1)-row reply table with (row 0 = zero row)
 *   slot_of_uniq [n] i64 : slot of distinct row u, G * cap = none;   counts [G] i64 : distinct rows per owner
 * An owner that needs more than cap distinct rows sets REC_FLAG_EXCHANGE_OVERFLOW.  All outputs are bit-exact targets. */
int rec_dedup_plan_workspace_bytes(int64_t n, int32_t num_shards, int64_t local_rows, size_t* bytes);
int rec_dedup_plan(int64_t n, int32_t num_slots, int64_t num_rows, int64_t padding_idx, int32_t num_shards,
                   int64_t local_rows, int32_t cap, const int64_t* ids, const int64_t* slot_offset, int32_t* sorted_pos,
                   int64_t* uniq_rows, int32_t* seg_offset, int32_t* n_uniq, int64_t* send_rows, int64_t* slot_of_pos,
                   int64_t* slot_of_uniq, int64_t* counts, int32_t* status, void* workspace, size_t workspace_bytes,
                   void* stream);

/* ------------------------------------------------------------------------------------------
 * Exchange layer of the row-sharded table (reference: the inter-GPU pull / push of
 * core.PSGPU, tools/static_gpubox_trainer.py:152-160,256 [EXT HeterPS]).  `comm` is an RCCL ncclComm_t — created
 * by rec_comm_init (one per process; rank 0 makes the id with rec_comm_unique_id and distributes its 128 bytes by
 * any means) or owned by the framework.  RCCL is bound at run time; REC_ENCCL when it is missing or fails.
 *   rec_alltoall_exchange: all-to-all(v) of rows of row_bytes bytes — send is grouped by destination rank
 *     (send_counts[d] rows, HOST array of `world` entries), recv by source rank; asynchronous on `stream`.
 *     A sharded step issues it three times: ids to owners, rows back, row-gradients to owners.
 *   rec_allreduce_sum_f32: the flat dense-gradient bucket.
 * ---------------------------------------------------------------------------------------- */
int rec_comm_unique_id(void* id128);
int rec_comm_init(const void* id128, int32_t world, int32_t rank, void** comm);
int rec_comm_destroy(void* comm);
/* ranks of the communicator as RCCL reports them (ncclCommCount) */
int rec_comm_size(void* comm, int32_t* world);
/* 1 when RCCL's entry points resolve in this process (so that every rank can AGREE on the native exchange before
 * any of them enters ncclCommInitRank), else 0; never fails */
int rec_comm_available(void);
int rec_alltoall_exchange(void* comm, const void* send, const int64_t* send_counts, void* recv,
                          const int64_t* recv_counts, int32_t row_bytes, void* stream);
int rec_allreduce_sum_f32(void* comm, float* buf, int64_t n, void* stream);

/* Row H — feature hash, HOST function: xxh32(str(field_idx)+value) % hash_dim
 * (models/rank/dnn/benchmark_reader.py:52).  `bytes` = the concatenated string. */
uint32_t rec_xxh32(const void* bytes, size_t len, uint32_t seed);
int rec_xxh32_hash_mod(const char* const* strings, const int32_t* field_idx, int64_t n,
                       uint32_t hash_dim, int64_t* out);

/* Rows R and H — input pipeline, HOST functions (no device code; outputs are caller-owned, ideally pinned).
 * rec_parse_slot_text : models/rank/deepfm/criteo_reader.py:61-103 (and dcn_v2/reader.py:41-89 with
 *   log1p_dense = 1): "click:L dense_feature:v x13 1:id ... 26:id" per line; missing sparse slot -> id 0,
 *   missing dense slot -> zeros, first value of a repeated slot wins.  label [n], ids [n,n_sparse] i64,
 *   dense [n,n_dense] f32 (parsed as double, then cast — as float() / np.float32 do).
 *   Number fields: "[-]digits[.digits]" with <= 15 significant digits is read as integer / 10^k (correctly rounded, the
 *   value strtod returns), "[-]digits" ids / labels of <= 18 digits by a digit loop; every other form goes to strtod /
 *   strtoll on a NUL-terminated copy of the field.  Only buf[0 .. len) is ever read (an mmap without a final newline is fine).
 * rec_parse_criteo_tsv: models/rank/dnn/benchmark_reader.py:39-54: "label \t 13 ints \t 26 strings";
 *   dense = (x - cont_min[j]) / cont_diff[j] ("" -> 0), ids = xxh32(str(field_idx)+string) % hash_dim with
 *   field_idx = 14..39 (the column index, as the reference hashes it).
 * threads <= 0: one per host core (capped at 64).  *n_lines = lines parsed (<= max_lines). */
int rec_count_lines(const char* buf, size_t len, int32_t threads, int64_t* n_lines); /* to size the outputs */
/* Indices (ascending) of the lines made of blanks only — the parsers above count them as lines; a loader that cuts
 * batches from a whole-file parse skips them like the reference's `for line in f` readers do.  *n_blank = how many
 * there are (may exceed max_out: only the first max_out are written). */
int rec_blank_lines(const char* buf, size_t len, int32_t threads, int64_t max_out, int64_t* idx, int64_t* n_blank);
/* One batch cut out of whole-file parses by line range (the in-memory pass of the gpubox loop): piece p = lines
 * [l0[p], l1[p]) of a rec_parse_feasign_slots result (values[p], lod[p] with row stride lod_stride[p], base[p]); the
 * batch's slot s holds its ids of piece 0, then of piece 1, ... — the CSR a parse of exactly those lines gives.
 * out_lod [num_slots, lines + 1], out_base [num_slots + 1], out_values [max_values >= total ids]. */
int rec_csr_cut(int32_t num_slots, int32_t n_pieces, const int64_t* const* values, const int64_t* const* lod,
                const int64_t* lod_stride, const int64_t* const* base, const int64_t* l0, const int64_t* l1,
                int32_t threads, int64_t* out_values, int64_t max_values, int64_t* out_lod, int64_t* out_base);
/* Occurrences of one byte value, multi-threaded (the ':' count bounds the values of rec_parse_feasign_slots). */
int rec_count_byte(const char* buf, size_t len, int32_t byte, int32_t threads, int64_t* n);
int rec_parse_slot_text(const char* buf, size_t len, int32_t n_sparse, int32_t n_dense,
                        int32_t log1p_dense, int64_t max_lines, int32_t threads, int64_t* label,
                        int64_t* ids, float* dense, int64_t* n_lines);
int rec_parse_criteo_tsv(const char* buf, size_t len, int32_t n_dense, int32_t n_sparse,
                         const float* cont_min, const float* cont_diff, uint32_t hash_dim,
                         int64_t max_lines, int32_t threads, int64_t* label, int64_t* ids, float* dense,
                         int64_t* n_lines);

/* Host parser of multi-value slot lines (models/rank/slot_dnn/queuedataset_reader.py:56-82): a line is
 * "feasign:slot feasign:slot ..." with uint64 feasigns and any number of values per slot.  Slots outside
 * [first_slot, first_slot + num_slots) are dropped; a slot that does not occur in a line gets the single padding
 * id 0 (line_process, :75-80).  Output = the layout rec_emb_gather_sumpool consumes, slot-major CSR:
 *   values[slot_base[s] + lod[s*(max_lines+1) + b] + j] = j-th value of slot s in line b      (token order)
 *   lod [num_slots, max_lines+1] (row s: offsets of slot s over the lines), slot_base [num_slots+1]
 * hash_rows = 0: values are the raw uint64 bit patterns (what the reference feeds its PS hash map);
 * hash_rows >= 2: rows of a hashed table, 0 -> 0 (padding row), f -> 1 + f % (hash_rows-1).
 * REC_EWORKSPACE (with *n_values = the size needed) when max_values is too small.  Multi-threaded, results do
 * not depend on the thread count. */
int rec_parse_feasign_slots(const char* buf, size_t len, int32_t first_slot, int32_t num_slots,
                            uint64_t hash_rows, int64_t max_lines, int64_t max_values, int32_t threads,
                            int64_t* values, int64_t* lod, int64_t* slot_base, int64_t* n_lines,
                            int64_t* n_values);

/* Fills buf[i] = i-th value of a counter-based generator, uniform in [lo,hi) — used to initialise
 * multi-GB tables on the device without a host round trip. */
int rec_fill_uniform(int64_t n, float* buf, float lo, float hi, uint64_t seed, void* stream);

/* Device-to-device copy on `stream` (hipMemcpyAsync).  The host mirrors copy parameter slices with it (the folded
 * layer-0 weights of DeepFM) so that a training step consists of C-ABI calls only and can be replayed from a recorded
 * call list (paddlerec_amd/plan.py). */
int rec_copy_async(void* dst, const void* src, size_t bytes, void* stream);
/* The same for a [rows, width_bytes] window of two pitched device buffers (hipMemcpy2DAsync): a binder that must hand
 * a framework a CONTIGUOUS column block of an engine output (the item / category halves of DIN's d_hist [B,T,E] as the
 * SelectedRows values of two separate embedding parameters, paddlerec_amd/paddle_ops/rec_paddle_ops.cc). */
int rec_copy_2d_async(void* dst, size_t dst_pitch_bytes, const void* src, size_t src_pitch_bytes, size_t width_bytes,
                      size_t rows, void* stream);
/* out [cols, rows] = in [rows, cols]^T (f32, both contiguous).  The DIN backward wants the first attention weight
 * transposed (att_w1_t); a binder without a tensor library of its own materialises it with this. */
int rec_transpose_f32(int64_t rows, int64_t cols, const float* in, float* out, void* stream);
/* dst[i] = (int64) round(src[i * src_stride]).  The gpubox nets carry the click label to the lookup as a float column
 * (show_click = concat([ones, cast(label)]), dnn/static_model.py:86-94); the accessor's push counts clicks in int64 —
 * a binder without a tensor library (the custom operator rec_ps_pull's gradient) converts the column with this. */
int rec_cast_f32_i64(int64_t n, const float* src, int64_t src_stride, int64_t* dst, void* stream);

/* MEASUREMENT AID, no reference counterpart: a stand-in link.  `blocks` (<= 64) workgroups copy `bytes` between two device
 * rings (ring_bytes each, a power of two; the copy wraps) and pace themselves so that the launch lasts bytes / gbytes_per_s + fixed_us
 * — what an RCCL all-to-all over xGMI is to the rest of the chip: a few resident workgroups moving bytes at the links'
 * rate.  paddlerec_amd/sharded.py issues it where the collectives of a world-1 run sit (REC_EMULATE_LINKS=<G>), so that
 * ONE GPU can show whether the exchange of a G-GPU step fits beside the dW GEMMs. */
int rec_link_emulate(size_t bytes, float gbytes_per_s, float fixed_us, int32_t blocks, const void* src, void* dst,
                     size_t ring_bytes, void* stream);

/* One wave busy-waits `micros` microseconds on `stream`.  Host-side stream probe: HIP maps streams onto a
 * few hardware queues and kernels of two streams that share a queue run strictly one after the other, so a
 * caller that wants its HBM-bound side stream to run underneath the GEMMs of its main stream spins both and
 * checks with events that the two spins overlapped (paddlerec_amd/ops.py: concurrent_stream). */
int rec_stream_spin(int32_t micros, void* stream);

/* Streams confined to the compute units [cu_begin, cu_end) of the current device (bit i of the HSA CU mask;
 * gfx950 has 256).  The row-sharded step partitions the chip for its tail: the HBM / xGMI bound chain (gradient
 * all-to-all, sparse Adam, next lookup) on a few CUs, the MFMA-bound dW GEMMs on the rest — two kernels that
 * merely share all CUs do not co-schedule (the GEMM's long-lived blocks hold every wave slot).  The stream is a
 * plain hipStream_t owned by the caller: rec_stream_destroy when done. */
int rec_stream_create_cu_range(int32_t cu_begin, int32_t cu_end, void** stream);
/* ... or to every `stride`-th compute unit starting at `first` (bits first, first + stride, ... < cu_total of the HSA CU
 * mask): a share of the chip that takes the same number of CUs from every XCD whatever the bit -> XCD layout is
 * (workgroup b runs on XCD b % 8, so a contiguous range would starve one XCD's share of every other kernel). */
int rec_stream_create_cu_stride(int32_t first, int32_t stride, int32_t cu_total, void** stream);
int rec_stream_destroy(void* stream);

/* ------------------------------------------------------------------------------------------
 * The tail of a CTR tower and its backward in ONE pass over the last hidden activation h [batch, n] (behind a ReLU):
 *   y_dnn = h @ w + bias;  logit = y1 + y2 + y_dnn (y1 / y2 may be NULL);  pred = sigmoid(clip(logit));
 *   loss = mean(log_loss(pred, label, eps));   dz = d loss / d logit;
 *   dx[b,j] = (relu == 0 || h[b,j] > 0) ? dz[b] * w[j] : 0;   dw[j] = sum_b h[b,j] dz[b];   db = sum_b dz[b]
 * = deepfm/net.py:169-174 (the last Linear(400, 1)) + dygraph_model.py:76-85 (sigmoid, log_loss, mean) forward, and
 * the backward of both, i.e. what rec_gemm_f32 (one output column) + rec_sigmoid_logloss + rec_mlp_head_bwd do in five
 * launches and two passes over h.  Same arithmetic per element as those entry points (eps, clip and mean_over as in
 * rec_sigmoid_logloss); the reductions (loss, dw, db) run in a fixed order of their own.  n % 4 == 0, n <= 512, act / dx /
 * w 16-byte aligned.  y_dnn may be NULL. */
int rec_ctr_head_workspace_bytes(int64_t batch, int32_t n, size_t* bytes);
int rec_ctr_head_fwd_bwd(int64_t batch, int32_t n, int64_t mean_over, const float* act, int64_t ld_act, const float* w,
                         const float* bias, const float* y1, const float* y2, const int64_t* label, float eps,
                         float clip_lo, float clip_hi, int32_t relu, float* y_dnn, float* pred, float* dz,
                         float* loss_out, float* dx, int64_t ld_dx, float* dw, float* db, void* workspace,
                         size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * The WHOLE DeepFM train step behind one entry point: what `dy_model.train_forward` + `loss.backward()` +
 * `optimizer.step()` do per batch in tools/trainer.py:148-152 for models/rank/deepfm (net.py:21-174,
 * dygraph_model.py:76-88) — FM lookup and interactions, the top MLP forward, sigmoid + log-loss (+ the AUC buckets),
 * the MLP backward, the FM backward, the merged lazy-Adam update of both embeddings and Adam on the dense parameters.
 * It issues, on ONE stream, exactly the rec_* calls of this header that the Python mirror (paddlerec_amd/deepfm.py)
 * issues for a step it records into a call list (paddlerec_amd/plan.py): same kernels, same arguments, same order —
 * results are bit-identical to that path (tests/test_deepfm_step_c.py).  A binder that is not Python gets the
 * launch-bound small-batch step (the reference's bigdata batch size 512: ~40 dependent launches) without paying a
 * foreign-function round trip per launch.  At B * num_slots <= 15360 the SelectedRows merge happens inside the
 * record update (rec_sparse_adam_record_small); larger batches group their ids first (rec_ids_group_slots when
 * slot_rows > 0, else rec_ids_group_payload).  At those launch-bound sizes, when the merge runs by row buckets, the
 * entry point goes one step further than the call list: the folds of the head's and the FM backward's partial sums,
 * the folded layer 0's backward and rec_adam_dense run as ROLES of the row update's launch, layer 0's weight fold
 * behind the lookup's blocks (csrc/tail_roles.h: every value by the statements of the stand-alone kernel, the
 * parameters dealt out by owner) — 10 launches instead of 16, the same bits (REC_SMALL_TAIL=0: the call list).  side_stream (may be NULL: everything on `stream`): a second stream of
 * the caller's on which the entry point runs the mirror's large-batch schedule — the id grouping forked in front of the
 * lookup, the sparse update underneath the dW_0 GEMM — ordered against `stream` with events; on return both streams'
 * work is ordered before anything the caller issues on `stream` next.  One step at a time per process.
 *
 * rec_deepfm_net describes the model the way the reference's state_dict does, as pointers into caller-owned device
 * memory: the table as 128-B-line records rec [table_rows, rec_stride] = W(dim) | W1 | m1 | v1 | pad with the
 * second-order moments in mv [table_rows, mv_stride] = m(dim) | v(dim) at v_offset, and the dense parameters as views
 * into ONE flat buffer (flat_param, with flat_grad / flat_m / flat_v of the same length) that rec_adam_dense walks once.
 * Linear i has weight w[i] [in_i, widths[i]] (Paddle layout) and bias b[i]; in_0 = (num_slots + dense_dim) * dim,
 * widths[n_linear - 1] = 1.  With 0 < dense_dim <= dim the step runs layer 0 on folded weights (rec_deepfm_desc
 * .compact_dense = 1): w0_folded [(num_slots + 1) * dim, widths[0]] is scratch that the CALLER zero-initialises once
 * (its last dim - dense_dim rows stay zero). */
#define REC_DEEPFM_MAX_LINEAR 8
typedef struct {
  int32_t num_slots, dim, dense_dim, n_linear;
  int32_t widths[REC_DEEPFM_MAX_LINEAR];
  int64_t table_rows;          /* rows of rec / mv */
  int64_t num_rows;            /* logical rows of the id space after slot offsets (= table_rows unless sharded) */
  int64_t padding_idx;         /* < 0: none */
  int64_t slot_rows;           /* > 0: slot s owns rows [s * slot_rows, (s + 1) * slot_rows) (slot-local grouping) */
  const int64_t* slot_offset;  /* device [num_slots] or NULL */
  float* rec;
  int32_t rec_stride;
  float* mv;
  int32_t mv_stride, v_offset;
  float *dense_w, *dense_w_one;          /* [dense_dim, dim], [dense_dim] */
  float *g_dense_w, *g_dense_w_one;      /* their gradient views */
  float* w[REC_DEEPFM_MAX_LINEAR];
  float* b[REC_DEEPFM_MAX_LINEAR];
  float* gw[REC_DEEPFM_MAX_LINEAR];
  float* gb[REC_DEEPFM_MAX_LINEAR];
  float *flat_param, *flat_grad, *flat_m, *flat_v;
  int64_t flat_numel;
  float* w0_folded;            /* compact mode (see above) and layer0_width > 0 */
  int32_t layer0_width;        /* 0, or a padded input width of layer 0 >= (num_slots + dense_dim) * dim for nets without
                                  the dense fold (dense_dim > dim): the step keeps feat at this sample stride
                                  (rec_deepfm_desc.feat_stride) and runs layer 0 on w0_folded [layer0_width, widths[0]],
                                  a zero-initialised buffer of the caller's whose leading rows it refreshes from w[0]
                                  every step — 39 fields x D 10 = 390 columns become 400, whole GEMM tiles.
                                  w0_folded == w[0] says the caller keeps layer0_width rows behind BOTH w[0] and gw[0]
                                  (the extra rows zero, inside the flat buffers): the step then copies nothing and
                                  writes dW_0 [layer0_width, widths[0]] straight into gw[0] (its extra rows come out
                                  exactly zero, so they stay zero under rec_adam_dense) */
} rec_deepfm_net;
int rec_deepfm_train_step_workspace_bytes(const rec_deepfm_net* net, int64_t batch, size_t* bytes);
/* ids [batch, num_slots] i64, dense [batch, dense_dim] f32, label [batch] i64 -> loss_out [1], pred_out [batch];
 * hyper->step is Adam's 1-based step count.  auc_pos / auc_neg [num_thresholds + 1] i64 or NULL (no metric).
 * status: the sticky out-of-range flag of the lookups (may be NULL). */
int rec_deepfm_train_step(const rec_deepfm_net* net, int64_t batch, const int64_t* ids, const float* dense,
                          const int64_t* label, const rec_adam_hyper* hyper, int64_t* auc_pos, int64_t* auc_neg,
                          int32_t num_thresholds, float* loss_out, float* pred_out, int32_t* status, void* workspace,
                          size_t workspace_bytes, void* stream, void* side_stream);

/* ------------------------------------------------------------------------------------------
 * The whole DIN train step behind one call — the per-batch body of tools/trainer.py:148-152 for models/rank/din:
 * train_forward (din/dygraph_model.py:85-100 = net.py:139-184 + binary_cross_entropy_with_logits), loss.backward(),
 * SGD step (din/dygraph_model.py:64-73; the learning rate of PiecewiseDecay is the caller's: `lr`).  Issues the ~25
 * rec_* calls of paddlerec_amd/din.py:_step on `stream` (csrc/din_step.hip), bit-identical to it; at the reference's batch
 * size (din/config.yaml: 32) the seven embedding updates are one rec_sparse_sgd_small_multi launch, larger batches sort.
 * All pointers are device memory the caller owns: seven contiguous [rows, dim] tables (three item tables, three category
 * tables, item_b [item_rows, 1]: independent parameters), the frozen attention MLP (att_w1 [4E, h1] AND
 * its transpose att_w1_t [h1, 4E] — rec_transpose_f32 makes one —, b1, w2 [h1, h2], b2, w3 [h2], b3: not registered
 * parameters in the reference's dygraph mode, App. B-9), the four registered Linear layers as [in, out] + bias with their
 * gradient buffers, and the flat parameter / gradient buffers holding those eight tensors (rec_sgd_dense walks them).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int32_t item_dim, cat_dim;            /* E = item_dim + cat_dim */
  int64_t item_rows, cat_rows;
  int32_t att_hidden1, att_hidden2;     /* attention MLP 4E -> h1 -> h2 -> 1 (80, 40) */
  int32_t mlp_hidden1, mlp_hidden2;     /* top MLP 2E -> m1 -> m2 -> 1 (80, 40), sigmoid */
  float *w_hist_item, *w_hist_cat, *w_tgt_item_seq, *w_tgt_cat_seq, *w_tgt_item, *w_tgt_cat, *w_item_b;
  const float *att_w1, *att_w1_t, *att_b1, *att_w2, *att_b2, *att_w3, *att_b3;
  float *w_con, *b_con, *w_l0, *b_l0, *w_l1, *b_l1, *w_l2, *b_l2;                  /* linearCon [E,E], linear_0..2 */
  float *g_w_con, *g_b_con, *g_w_l0, *g_b_l0, *g_w_l1, *g_b_l1, *g_w_l2, *g_b_l2;  /* their gradients (written) */
  float *flat_param, *flat_grad;
  int64_t flat_numel;
} rec_din_net;
int rec_din_train_step_workspace_bytes(const rec_din_net* net, int64_t batch, int32_t max_len, size_t* bytes);
/* hist_item / hist_cat / target_item_seq / target_cat_seq / mask [batch, max_len] i64 (mask 0 valid, -1e9 padding),
 * target_item / target_cat [batch] i64, label [batch] f32 -> loss_out [1] (mean BCE), pred_out [batch] = sigmoid(logit).
 * status: the sticky out-of-range flag of the lookups.
 * side_stream (NULL or == stream: everything on `stream`): batches above the one-launch merge group the merge keys of the
 * four [batch, max_len] tables on it from the start of the step and update two of the four tables on it behind the
 * backward (batch 4096: 1.72 -> 1.47 ms at max_len 100); joined into `stream` before the call returns.  Same result. */
int rec_din_train_step(const rec_din_net* net, int64_t batch, int32_t max_len, const int64_t* hist_item,
                       const int64_t* hist_cat, const int64_t* target_item, const int64_t* target_cat, const float* label,
                       const int64_t* mask, const int64_t* target_item_seq, const int64_t* target_cat_seq, float lr,
                       float* loss_out, float* pred_out, int32_t* status, void* workspace, size_t workspace_bytes,
                       void* stream, void* side_stream);

/* ------------------------------------------------------------------------------------------
 * The whole DCN-v2 train step behind one call — the per-batch body of tools/trainer.py:148-152 for models/rank/dcn_v2:
 * train_forward (dcn_v2/dygraph_model.py:103-127 = net.py:89-137: lookup + dense_emb Linear, CrossNetV2 or CrossNetMix,
 * the DNN tower with its train-mode Dropout pairs, the stacked or the parallel head, log_loss), loss.backward(), Adam with
 * ClipGradByGlobalNorm and L2Decay on the DNN weights (dygraph_model.py:73-88, net.py:164-170).  Issues the rec_* calls of
 * paddlerec_amd/dcn_v2.py:train_step on `stream` (csrc/dcn_v2_step.hip), bit-identical to it, for all four structures
 * (is_stacked x low_rank_mix) with and without dropout.  All pointers are device memory the caller owns: the table as
 * [num_rows, emb_stride] with its Adam moments [num_rows, state_stride], every dense parameter as Paddle's [in, out] with
 * its gradient buffer, and the flat buffers holding all dense parameters / gradients / moments in one order
 * (rec_adam_dense and rec_sumsq walk them).
 * ---------------------------------------------------------------------------------------- */
#define REC_DCN_MAX_LAYERS 8
typedef struct {
  int32_t num_slots, dim, dense_dim;    /* S, D, Dn: d = (S + Dn) * D */
  int64_t num_rows, padding_idx;        /* padding_idx < 0: none (net.py:45-54 uses 0) */
  int32_t emb_stride, state_stride;     /* row strides (floats) of emb and of emb_m / emb_v */
  float *emb, *emb_m, *emb_v;
  int32_t cross_num, n_dnn;             /* <= REC_DCN_MAX_LAYERS each */
  int32_t widths[REC_DCN_MAX_LAYERS];   /* layer_sizes of the DNN tower */
  int32_t is_stacked, low_rank_mix;     /* net.py:110-137 stacked / parallel; CrossNetV2 / CrossNetMix */
  int32_t num_experts, low_rank;        /* CrossNetMix only */
  float dropout_rate;                   /* 0: no dropout (eval-mode tower) */
  float l2_dnn, clip_norm;              /* 0: off */
  uint64_t dropout_seed;                /* mask streams of layer i at Adam step t: (t * n_dnn + i) * 2 and + 1 */
  float *dense_emb_w, *dense_emb_b, *g_dense_emb_w, *g_dense_emb_b;   /* Linear(Dn -> D*Dn) */
  float *cross_w[REC_DCN_MAX_LAYERS], *cross_b[REC_DCN_MAX_LAYERS];   /* CrossNetV2 [d,d], [d] */
  float *g_cross_w[REC_DCN_MAX_LAYERS], *g_cross_b[REC_DCN_MAX_LAYERS];
  float *mix_u[REC_DCN_MAX_LAYERS], *mix_v[REC_DCN_MAX_LAYERS], *mix_c[REC_DCN_MAX_LAYERS], *mix_bias[REC_DCN_MAX_LAYERS];
  float *g_mix_u[REC_DCN_MAX_LAYERS], *g_mix_v[REC_DCN_MAX_LAYERS], *g_mix_c[REC_DCN_MAX_LAYERS],
      *g_mix_bias[REC_DCN_MAX_LAYERS];                                /* CrossNetMix [E,d,r], [E,d,r], [E,r,r], [d] */
  float *gate_w, *gate_b, *g_gate_w, *g_gate_b;                       /* the E gating Linear(d,1) stacked: [d,E], [E] */
  float *dnn_w[REC_DCN_MAX_LAYERS], *dnn_b[REC_DCN_MAX_LAYERS], *g_dnn_w[REC_DCN_MAX_LAYERS], *g_dnn_b[REC_DCN_MAX_LAYERS];
  float *fc_w, *fc_b, *g_fc_w, *g_fc_b;                               /* [widths[n-1] (+ d when parallel), 1], [1] */
  float *flat_param, *flat_grad, *flat_m, *flat_v;
  int64_t flat_numel;
} rec_dcn_v2_net;
int rec_dcn_v2_train_step_workspace_bytes(const rec_dcn_v2_net* net, int64_t batch, size_t* bytes);
/* ids [batch, num_slots] i64, dense [batch, dense_dim] f32, label [batch] i64 -> loss_out [1], pred_out [batch];
 * hyper->step = the Adam step count t (1-based; also keys the dropout masks); auc_pos / auc_neg i64
 * [num_thresholds + 1] or NULL. */
int rec_dcn_v2_train_step(const rec_dcn_v2_net* net, int64_t batch, const int64_t* ids, const float* dense,
                          const int64_t* label, const rec_adam_hyper* hyper, int64_t* auc_pos, int64_t* auc_neg,
                          int32_t num_thresholds, float* loss_out, float* pred_out, int32_t* status, void* workspace,
                          size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RECENGINE_H_ */
