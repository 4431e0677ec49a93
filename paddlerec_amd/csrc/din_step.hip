// rec_din_train_step: the whole DIN train step issued from C (include/recengine.h, last section).
//
// Reference call site: tools/trainer.py:148-152 for models/rank/din — dy_model.train_forward
// (din/dygraph_model.py:85-100: net.py:139-184 forward, binary_cross_entropy_with_logits), loss.backward(),
// optimizer.step() (SGD, din/dygraph_model.py:64-73).  The Python mirror (paddlerec_amd/din.py:_step) issues the same
// ~25 rec_* calls through ctypes; at the reference's own batch size (din/config.yaml: 32) every one of them is a few
// microseconds of GPU work and the host's share per foreign call is the step time.  This file states that call list in
// C++ for every binder that is not Python, exactly as csrc/deepfm_step.hip does for DeepFM: no kernel of its own, every
// line a call of an entry point of this library, in the mirror's order with the mirror's arguments — bit-identical by
// construction (tests/test_din_step_c.py holds the two to that).
#include <stdlib.h>

#include "rec_common.h"
#include "tail_roles.h"

using namespace rec;

namespace {

struct Carve {
  char* base;
  size_t off = 0;
  explicit Carve(void* p) : base((char*)p) {}
  template <class T>
  T* take(size_t count) {
    T* p = base ? (T*)(base + off) : nullptr;
    off += align_up(count * sizeof(T), 256);
    return p;
  }
  void* bytes(size_t n) { return take<char>(n); }
};

#define REC_TRY(call)                 \
  do {                                \
    if (int rc_ = (call)) return rc_; \
  } while (0)

constexpr int64_t kSmallMergeMax = 15360;     // ops.SMALL_MERGE_MAX: lookups one rec_sparse_sgd_small* launch merges

struct Shape {
  int Ei, Ec, E, H1, H2, M1, M2;
  int64_t B, T, n;
  bool saves;          // the attention forward saves its layer-1 activations for this shape
  bool small;          // all seven row updates in ONE launch (rec_sparse_sgd_small_multi)
};

rec_din_desc att_desc(const rec_din_net* net, const Shape& s) {
  return rec_din_desc{s.B, (int32_t)s.T, s.Ei, s.Ec, s.H1, s.H2, net->item_rows, net->cat_rows, s.Ei, s.Ec};
}

int shape_of(const rec_din_net* net, int64_t B, int32_t T, Shape* s) {
  REC_REQUIRE(net, REC_EINVAL, "net is NULL");
  REC_REQUIRE(B > 0 && T > 0 && B * (int64_t)T < (1ll << 31), REC_EINVAL, "bad batch %lld x %d", (long long)B, T);
  REC_REQUIRE(net->item_dim > 0 && net->cat_dim > 0 && net->att_hidden1 > 0 && net->att_hidden2 > 0 &&
                  net->mlp_hidden1 > 0 && net->mlp_hidden2 > 0 && net->item_rows > 0 && net->cat_rows > 0,
              REC_EINVAL, "bad net sizes");
  s->Ei = net->item_dim; s->Ec = net->cat_dim; s->E = s->Ei + s->Ec;
  s->H1 = net->att_hidden1; s->H2 = net->att_hidden2; s->M1 = net->mlp_hidden1; s->M2 = net->mlp_hidden2;
  s->B = B; s->T = T; s->n = B * T;
  const rec_din_desc d = att_desc(net, *s);
  s->saves = rec_din_saves_act1(&d) == 1;
  static const bool multi_env = [] { const char* v = getenv("REC_SMALL_MULTI"); return !(v && *v == '0'); }();
  s->small = multi_env && s->n <= kSmallMergeMax && s->Ei <= 256 && s->Ec <= 256;            // din.py:_step
  return REC_OK;
}

struct Buffers {
  float *pooled, *attw, *act1, *emb, *item_b, *x1, *x2, *logit, *dz, *d2, *d1, *de0, *dpooled, *dh, *dq;
  // sort-based merge: one grouping buffer set and one hot-row partials buffer per [B, T] table (din.py: slot i), a fifth
  // for the per-sample tables when the batch itself is above the one-launch merge
  float* pp[5];
  int32_t *sorted_pos[5], *seg_offset[5], *n_uniq[5];
  int64_t* uniq_rows[5];
  void *ws, *ws_group, *ws_group_side, *ws_att, *ws_att_bwd;
  size_t ws_bytes, ws_group_bytes, ws_att_bytes, ws_att_bwd_bytes;
};

int gemm_need(int64_t m, int n, int k, int lda, int ldb, int ldc, int ta, int tb, int epi, size_t* need) {
  rec_gemm_desc d{m, n, k, lda, ldb, ldc, ta, tb, epi, 0};
  size_t b = 0;
  REC_TRY(rec_gemm_f32_workspace_bytes(&d, &b));
  if (b > *need) *need = b;
  return REC_OK;
}

int carve(const rec_din_net* net, const Shape& s, void* workspace, Buffers* bf, size_t* total) {
  Carve c(workspace);
  const size_t B = (size_t)s.B, n = (size_t)s.n, E = (size_t)s.E;
  bf->pooled = c.take<float>(B * E);
  bf->attw = c.take<float>(n);
  bf->act1 = s.saves ? c.take<float>(n * s.H1) : nullptr;
  bf->emb = c.take<float>(B * 2 * E);
  bf->item_b = c.take<float>(B);
  bf->x1 = c.take<float>(B * s.M1);
  bf->x2 = c.take<float>(B * s.M2);
  bf->logit = c.take<float>(B);
  bf->dz = c.take<float>(B);
  bf->d2 = c.take<float>(B * s.M2);
  bf->d1 = c.take<float>(B * s.M1);
  bf->de0 = c.take<float>(B * 2 * E);
  bf->dpooled = c.take<float>(B * E);
  bf->dh = c.take<float>(n * E);
  bf->dq = c.take<float>(n * E);
  for (int i = 0; i < 5; ++i) {
    bf->pp[i] = nullptr;
    bf->sorted_pos[i] = bf->seg_offset[i] = bf->n_uniq[i] = nullptr;
    bf->uniq_rows[i] = nullptr;
  }
  bf->ws_group = bf->ws_group_side = nullptr;
  bf->ws_group_bytes = 0;
  if (!s.small) {                       // sort-based merge of the four [B, T] tables
    size_t pb = 0, gb = 0, b = 0;
    for (int i = 0; i < 5; ++i) {
      REC_TRY(rec_segment_partials_bytes((int64_t)n, i == 4 ? (s.Ei > s.Ec ? s.Ei : s.Ec) : i % 2 ? s.Ec : s.Ei, &pb));
      bf->pp[i] = (float*)c.bytes(pb > 4 ? pb : 4);
      bf->sorted_pos[i] = c.take<int32_t>(n);
      bf->uniq_rows[i] = c.take<int64_t>(n);
      bf->seg_offset[i] = c.take<int32_t>(n + 1);
      bf->n_uniq[i] = c.take<int32_t>(4);
    }
    REC_TRY(rec_ids_group_workspace_bytes((int64_t)n, net->item_rows, &gb));
    REC_TRY(rec_ids_group_workspace_bytes((int64_t)n, net->cat_rows, &b));
    if (b > gb) gb = b;
    bf->ws_group = c.bytes(gb);
    bf->ws_group_side = c.bytes(gb);
    bf->ws_group_bytes = gb;
  }
  const rec_din_desc d = att_desc(net, s);
  size_t ab = 0;
  REC_TRY(rec_din_attention_pool_fwd_workspace_bytes(&d, &ab));
  bf->ws_att = ab ? c.bytes(ab) : nullptr;
  bf->ws_att_bytes = ab;
  REC_TRY(rec_din_attention_pool_bwd_workspace_bytes(&d, &ab));
  bf->ws_att_bwd = ab ? c.bytes(ab) : nullptr;
  bf->ws_att_bwd_bytes = ab;
  size_t need = 0, lb = 0;
  const int E2 = 2 * s.E;
  REC_TRY(gemm_need(s.B, s.E, s.E, s.E, s.E, E2, 0, 0, REC_EPI_BIAS, &need));            // linearCon
  REC_TRY(gemm_need(s.B, s.M1, E2, E2, s.M1, s.M1, 0, 0, REC_EPI_BIAS_SIGMOID, &need));  // linear_0
  REC_TRY(gemm_need(s.B, s.M2, s.M1, s.M1, s.M2, s.M2, 0, 0, REC_EPI_BIAS_SIGMOID, &need));
  REC_TRY(gemm_need(s.B, 1, s.M2, s.M2, 1, 1, 0, 0, REC_EPI_ADD, &need));
  REC_TRY(gemm_need(s.M2, 1, (int)s.B, s.M2, 1, 1, 1, 0, REC_EPI_NONE, &need));          // dW linear_2
  REC_TRY(gemm_need(s.B, s.M2, 1, 1, 1, s.M2, 0, 1, REC_EPI_DSIGMOID, &need));
  REC_TRY(gemm_need(s.M1, s.M2, (int)s.B, s.M1, s.M2, s.M2, 1, 0, REC_EPI_NONE, &need));
  REC_TRY(gemm_need(s.B, s.M1, s.M2, s.M2, s.M2, s.M1, 0, 1, REC_EPI_DSIGMOID, &need));
  REC_TRY(gemm_need(E2, s.M1, (int)s.B, E2, s.M1, s.M1, 1, 0, REC_EPI_NONE, &need));
  REC_TRY(gemm_need(s.B, E2, s.M1, s.M1, s.M1, E2, 0, 1, REC_EPI_NONE, &need));
  REC_TRY(gemm_need(s.E, s.E, (int)s.B, s.E, E2, s.E, 1, 0, REC_EPI_NONE, &need));
  REC_TRY(gemm_need(s.B, s.E, s.E, E2, s.E, s.E, 0, 1, REC_EPI_NONE, &need));
  REC_TRY(rec_logloss_workspace_bytes(s.B, &lb));
  if (lb > need) need = lb;
  need = align_up(need, 256);
  bf->ws = c.bytes(need);
  bf->ws_bytes = need;
  *total = c.off;
  return REC_OK;
}

// C[m,n] (ldc) = epi(op(A) @ op(B)); leading dimensions as the mirror's tensor views have them
int gemm(int64_t m, int n, int k, int lda, int ldb, int ldc, bool ta, bool tb, int epi, const float* A, const float* Bm,
         float* C, const float* bias, const float* aux0, int ld0, const float* aux1, int ld1, float* b_colsum,
         const Buffers& bf, void* st) {
  rec_gemm_desc d{m, n, k, lda, ldb, ldc, ta ? 1 : 0, tb ? 1 : 0, epi, 0};
  rec_gemm_epilogue_args x{};
  x.bias = bias; x.aux0 = aux0; x.ld_aux0 = ld0; x.aux1 = aux1; x.ld_aux1 = ld1; x.b_colsum = b_colsum;
  return rec_gemm_f32(&d, A, Bm, C, &x, bf.ws, bf.ws_bytes, st);
}

// dW (+ db) and dX of one Linear in one call (din.py:lin_bwd -> ops.linear_backward): one launch at the shipped batch size
int lin_bwd(int64_t B, int in, int out, int ldx, int ldg, int ldw, int lddx, const float* X, const float* G, const float* W,
            float* dW, float* db, float* dX, int epi, const float* aux0, int ld0, const Buffers& bf, void* st) {
  rec_gemm_desc d0{in, out, (int)B, ldx, ldg, out, 1, 0, REC_EPI_NONE, 0};
  rec_gemm_desc d1{B, in, out, ldg, ldw, lddx, 0, 1, epi, 0};
  rec_gemm_epilogue_args x0{}, x1{};
  x0.b_colsum = db;
  x1.aux0 = aux0; x1.ld_aux0 = ld0;
  return rec_gemm_f32_pair(&d0, X, G, dW, &x0, &d1, G, W, dX, &x1, bf.ws, bf.ws_bytes, st);
}

}  // namespace

extern "C" int rec_din_train_step_workspace_bytes(const rec_din_net* net, int64_t batch, int32_t max_len, size_t* bytes) {
  REC_REQUIRE(bytes, REC_EINVAL, "bytes is NULL");
  Shape s;
  REC_TRY(shape_of(net, batch, max_len, &s));
  Buffers bf;
  return carve(net, s, nullptr, &bf, bytes);
}

extern "C" int rec_din_train_step(const rec_din_net* net, int64_t batch, int32_t max_len, const int64_t* hist_item,
                                  const int64_t* hist_cat, const int64_t* target_item, const int64_t* target_cat,
                                  const float* label, const int64_t* mask, const int64_t* target_item_seq,
                                  const int64_t* target_cat_seq, float lr, float* loss, float* pred, int32_t* status,
                                  void* workspace, size_t workspace_bytes, void* stream, void* side_stream) {
  Shape s;
  REC_TRY(shape_of(net, batch, max_len, &s));
  REC_REQUIRE(hist_item && hist_cat && target_item && target_cat && label && mask && target_item_seq && target_cat_seq &&
                  loss && pred && status, REC_EINVAL, "null pointer argument");
  REC_REQUIRE(net->w_hist_item && net->w_hist_cat && net->w_tgt_item_seq && net->w_tgt_cat_seq && net->w_tgt_item &&
                  net->w_tgt_cat && net->w_item_b && net->att_w1 && net->att_w1_t && net->att_b1 && net->att_w2 &&
                  net->att_b2 && net->att_w3 && net->att_b3 && net->w_con && net->b_con && net->w_l0 && net->b_l0 &&
                  net->w_l1 && net->b_l1 && net->w_l2 && net->b_l2 && net->g_w_con && net->g_b_con && net->g_w_l0 &&
                  net->g_b_l0 && net->g_w_l1 && net->g_b_l1 && net->g_w_l2 && net->g_b_l2 && net->flat_param &&
                  net->flat_grad && net->flat_numel > 0, REC_EINVAL, "net has a NULL parameter pointer");
  Buffers bf;
  size_t need = 0;
  REC_TRY(carve(net, s, workspace, &bf, &need));
  REC_REQUIRE(workspace && workspace_bytes >= need, REC_EWORKSPACE, "workspace %zu < %zu", workspace_bytes, need);
  const int64_t B = s.B;
  const int E = s.E, Ei = s.Ei, E2 = 2 * s.E, M1 = s.M1, M2 = s.M2;
  const rec_din_desc d = att_desc(net, s);

  // ---- the SelectedRows merge and its two-stream schedule (din.py:_step).  Jobs in din.py's order: the target-seq
  //      tables first (one row per sample collects all its history positions)
  struct Job { int64_t n; const int64_t* ids; const float* grad; float* P; int dim; int64_t rows; int stride; };
  const Job jobs[7] = {
      {s.n, target_item_seq, bf.dq, net->w_tgt_item_seq, Ei, net->item_rows, E},
      {s.n, target_cat_seq, bf.dq + Ei, net->w_tgt_cat_seq, s.Ec, net->cat_rows, E},
      {s.n, hist_item, bf.dh, net->w_hist_item, Ei, net->item_rows, E},
      {s.n, hist_cat, bf.dh + Ei, net->w_hist_cat, s.Ec, net->cat_rows, E},
      {B, target_item, bf.de0 + E, net->w_tgt_item, Ei, net->item_rows, E2},
      {B, target_cat, bf.de0 + E + Ei, net->w_tgt_cat, s.Ec, net->cat_rows, E2},
      {B, target_item, bf.dz, net->w_item_b, 1, net->item_rows, 1}};
  auto sorts = [&](const Job& j) { return !(j.n <= kSmallMergeMax && j.dim <= 256); };   // din.py:_sorts
  // With a side stream and all four [B, T] tables on the sort-based merge: their merge keys (a function of the ids) are
  // computed on the side stream from the start of the step, and behind the backward one item and one category table are
  // updated on each stream.  Independent tables, same calls and arguments: the result does not depend on the schedule
  const bool two = side_stream != nullptr && side_stream != stream && !s.small && sorts(jobs[0]) && sorts(jobs[1]) &&
                   sorts(jobs[2]) && sorts(jobs[3]);
  hipEvent_t* ev = nullptr;
  if (two) REC_TRY(step_events(stream, side_stream, &ev));
  auto order = [&](int k, void* from, void* to) -> int {    // everything issued on `from` so far happens before `to` goes on
    REC_REQUIRE(hipEventRecord(ev[k], (hipStream_t)from) == hipSuccess &&
                    hipStreamWaitEvent((hipStream_t)to, ev[k], 0) == hipSuccess, REC_EHIP, "stream ordering failed");
    return REC_OK;
  };
  auto group = [&](int i, int b, void* ws, size_t ws_bytes, void* st) -> int {       // job i into buffer set b
    const Job& j = jobs[i];
    return rec_ids_group_payload(j.n, 1, j.rows, -1, j.ids, nullptr, nullptr, bf.sorted_pos[b], bf.uniq_rows[b],
                                 bf.seg_offset[b], bf.n_uniq[b], status, ws, ws_bytes, st);
  };
  if (two) {
    REC_TRY(order(0, stream, side_stream));
    for (int i = 0; i < 4; ++i) REC_TRY(group(i, i, bf.ws_group_side, bf.ws_group_bytes, side_stream));
  }

  // ---- forward (din.py:forward = net.py:139-184)
  // launch-bound sizes (din/config.yaml:20, bs 32): the three target gathers ride behind the blocks of the attention's
  // combine launch (or go out as ONE launch of their own), the dense SGD in the merges' launch (tail_roles.h;
  // REC_SMALL_TAIL=0: the mirror's list of launches) — same values, four launches less
  static const bool ride = [] { const char* v = getenv("REC_SMALL_TAIL"); return !(v && *v == '0'); }();
  GatherJobs gjs;
  if (s.small && ride) {
    const GatherJob gj[3] = {
        {B, Ei, Ei, net->item_rows, -1, target_item, net->w_tgt_item, bf.emb + E, 1, E2},             // net.py:143,152
        {B, s.Ec, s.Ec, net->cat_rows, -1, target_cat, net->w_tgt_cat, bf.emb + E + Ei, 1, E2},
        {B, 1, 1, net->item_rows, -1, target_item, net->w_item_b, bf.item_b, 0, 0}};
    REC_TRY(gather_jobs_make(3, gj, &gjs));
    din_combine_rider_set(&gjs, status);
  }
  const int att_rc = rec_din_attention_pool_fwd_ws(&d, hist_item, hist_cat, target_item_seq, target_cat_seq, mask,
                                                   net->w_hist_item, net->w_hist_cat, net->w_tgt_item_seq,
                                                   net->w_tgt_cat_seq, net->att_w1, net->att_b1, net->att_w2, net->att_b2,
                                                   net->att_w3, net->att_b3, bf.pooled, bf.attw, bf.act1, status, bf.ws_att,
                                                   bf.ws_att_bytes, stream);                              // net.py:141-173
  const bool gathers_rode = din_combine_rider_take();    // (always: an unconsumed rider must not wait for a later call)
  if (att_rc) return att_rc;
  REC_TRY(gemm(B, E, E, E, E, E2, false, false, REC_EPI_BIAS, bf.pooled, net->w_con, bf.emb, net->b_con, nullptr, 0,
               nullptr, 0, nullptr, bf, stream));                                                        // net.py:175-176
  if (s.small && ride) {
    if (!gathers_rode) {                                // (the attention did not split by tiles: no combine launch)
      const GatherJob gj[3] = {
          {B, Ei, Ei, net->item_rows, -1, target_item, net->w_tgt_item, bf.emb + E, 1, E2},
          {B, s.Ec, s.Ec, net->cat_rows, -1, target_cat, net->w_tgt_cat, bf.emb + E + Ei, 1, E2},
          {B, 1, 1, net->item_rows, -1, target_item, net->w_item_b, bf.item_b, 0, 0}};
      REC_TRY(emb_gather_multi(3, gj, status, stream));
    }
  } else {
  REC_TRY(rec_emb_gather(B, Ei, Ei, net->item_rows, -1, target_item, net->w_tgt_item, bf.emb + E, 1, E2, status,
                         stream));                                                                        // net.py:143,152
  REC_TRY(rec_emb_gather(B, s.Ec, s.Ec, net->cat_rows, -1, target_cat, net->w_tgt_cat, bf.emb + E + Ei, 1, E2, status,
                         stream));
  REC_TRY(rec_emb_gather(B, 1, 1, net->item_rows, -1, target_item, net->w_item_b, bf.item_b, 0, 0, status, stream));
  }
  REC_TRY(gemm(B, M1, E2, E2, M1, M1, false, false, REC_EPI_BIAS_SIGMOID, bf.emb, net->w_l0, bf.x1, net->b_l0, nullptr, 0,
               nullptr, 0, nullptr, bf, stream));
  REC_TRY(gemm(B, M2, M1, M1, M2, M2, false, false, REC_EPI_BIAS_SIGMOID, bf.x1, net->w_l1, bf.x2, net->b_l1, nullptr, 0,
               nullptr, 0, nullptr, bf, stream));
  REC_TRY(gemm(B, 1, M2, M2, 1, 1, false, false, REC_EPI_ADD, bf.x2, net->w_l2, bf.logit, net->b_l2, nullptr, 0,
               bf.item_b, 1, nullptr, bf, stream));                                                      // net.py:180-183
  REC_TRY(rec_bce_with_logits(B, 0, bf.logit, label, pred, bf.dz, loss, bf.ws, bf.ws_bytes, stream));

  // ---- backward chain: dW (+ db from the same pass), then dX with sigmoid' of the layer's input fused (din.py:lin_bwd)
  REC_TRY(gemm(M2, 1, (int)B, M2, 1, 1, true, false, REC_EPI_NONE, bf.x2, bf.dz, net->g_w_l2, nullptr, nullptr, 0,
               nullptr, 0, net->g_b_l2, bf, stream));
  REC_TRY(gemm(B, M2, 1, 1, 1, M2, false, true, REC_EPI_DSIGMOID, bf.dz, net->w_l2, bf.d2, nullptr, bf.x2, M2, nullptr, 0,
               nullptr, bf, stream));
  REC_TRY(lin_bwd(B, M1, M2, M1, M2, M2, M1, bf.x1, bf.d2, net->w_l1, net->g_w_l1, net->g_b_l1, bf.d1, REC_EPI_DSIGMOID,
                  bf.x1, M1, bf, stream));
  REC_TRY(lin_bwd(B, E2, M1, E2, M1, M1, E2, bf.emb, bf.d1, net->w_l0, net->g_w_l0, net->g_b_l0, bf.de0, REC_EPI_NONE,
                  nullptr, 0, bf, stream));                    // [d linearCon out | d target_concat]
  REC_TRY(lin_bwd(B, E, E, E, E2, E, E, bf.pooled, bf.de0, net->w_con, net->g_w_con, net->g_b_con, bf.dpooled, REC_EPI_NONE,
                  nullptr, 0, bf, stream));
  REC_TRY(rec_din_attention_pool_bwd_ws(&d, hist_item, hist_cat, target_item_seq, target_cat_seq, net->w_hist_item,
                                        net->w_hist_cat, net->w_tgt_item_seq, net->w_tgt_cat_seq, net->att_w1,
                                        net->att_w1_t, net->att_b1, net->att_w2, net->att_b2, net->att_w3, bf.attw,
                                        s.saves ? bf.pooled : nullptr, s.saves ? bf.act1 : nullptr, bf.dpooled, bf.dh,
                                        bf.dq, bf.ws_att_bwd, bf.ws_att_bwd_bytes, stream));

  // ---- SGD (din/dygraph_model.py:64-73): merged rows of the seven tables, then the dense parameters in one launch
  if (s.small) {
    rec_small_sgd_job sj[7];
    for (int i = 0; i < 7; ++i) {
      sj[i].n = jobs[i].n; sj[i].emb_dim = jobs[i].dim; sj[i].row_stride = jobs[i].dim;
      sj[i].num_rows = jobs[i].rows; sj[i].padding_idx = -1; sj[i].ids = jobs[i].ids; sj[i].grad = jobs[i].grad;
      sj[i].grad_layout = rec_grad_layout{1, 1, jobs[i].stride, nullptr, nullptr, 0};
      sj[i].P = jobs[i].P;
    }
    if (ride) {
      REC_TRY(sparse_sgd_small_multi_dense(7, sj, lr, status, stream, net->flat_numel, net->flat_param, net->flat_grad));
      return REC_OK;        // (small: one stream, nothing to join)
    }
    REC_TRY(rec_sparse_sgd_small_multi(7, sj, lr, status, stream));
  } else {
    auto update = [&](int i, bool grouped, void* st) -> int {
      const Job& j = jobs[i];
      const rec_grad_layout gl{1, 1, j.stride, nullptr, nullptr, 0};
      if (!sorts(j))                                      // din.py:_sgd_rows: one launch per small table
        return rec_sparse_sgd_small(j.n, j.dim, j.dim, j.rows, -1, j.ids, j.grad, &gl, j.P, lr, status, st);
      const int b = i < 4 ? i : 4;
      if (!grouped) REC_TRY(group(i, b, bf.ws_group, bf.ws_group_bytes, st));
      REC_TRY(rec_segment_partials(j.n, j.dim, bf.n_uniq[b], bf.seg_offset[b], bf.sorted_pos[b], j.grad, &gl, bf.pp[b], st));
      rec_grad_layout glp = gl;
      glp.partials = bf.pp[b];
      return rec_sparse_sgd_rows(j.n, j.dim, j.dim, bf.n_uniq[b], bf.uniq_rows[b], bf.seg_offset[b], bf.sorted_pos[b], j.grad,
                                 &glp, j.P, lr, st);
    };
    if (two) {
      REC_TRY(order(1, side_stream, stream));             // the merge keys
      REC_TRY(order(2, stream, side_stream));             // the gradients
      REC_TRY(update(0, true, side_stream));
      REC_TRY(update(3, true, side_stream));
      for (int i = 1; i < 7; ++i)
        if (i != 3) REC_TRY(update(i, i < 4, stream));
    } else {
      for (int i = 0; i < 7; ++i) REC_TRY(update(i, false, stream));
    }
  }
  REC_TRY(rec_sgd_dense(net->flat_numel, net->flat_param, net->flat_grad, lr, stream));
  if (two) REC_TRY(order(3, side_stream, stream));
  return REC_OK;
}
