// rec_deepfm_train_step: the whole DeepFM train step issued from C (include/recengine.h, last section).
//
// Reference call site: tools/trainer.py:148-152 (train_forward / backward / optimizer.step per batch) for
// models/rank/deepfm (net.py:21-174, dygraph_model.py:76-88).  The Python mirror (paddlerec_amd/deepfm.py:train_step)
// issues ~40 rec_* calls per step through ctypes; at the reference's own batch size (config_bigdata.yaml: 512) the
// step is a chain of dependent launches of a few microseconds each and the host's share per call sets the step time
// (profiles/r03_small_batch.txt).  paddlerec_amd/plan.py removes that share for Python by replaying a recorded call
// list; this file is the same list stated in C++ for every other binder.  Nothing here launches a kernel of its own:
// every line below is a call of an entry point of this library, in the mirror's order with the mirror's arguments,
// so the two paths are bit-identical by construction (tests/test_deepfm_step_c.py holds them to that).
// One exception, at the launch-bound sizes only (tail_roles.h): the folds of the head's and the FM backward's partial sums,
// the folded layer 0's backward and the dense Adam do not go out as launches of their own but as roles of the launch that
// merges and updates the table rows — the same statements per value, four launches less per step (REC_SMALL_TAIL=0: the
// mirror's list).
#include <stdlib.h>

#include <map>
#include <mutex>
#include <tuple>

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

struct Shape {
  int S, D, Dn, n;
  bool compact;
  int fp;             // fields per sample in feat
  int in0;            // input width of layer 0 as the GEMM sees it (fp * D, or the padded width)
  int in0_real;       // fp * D
  bool pad;           // layer 0 on a zero-padded input width (net->layer0_width)
  int in0_full;       // rows of w[0] as stored ((S + Dn) * D)
  int max_w;          // widest activation / gradient row
  bool small;         // merge inside the record update
  bool head;          // fused one-logit head backward
  bool ctr_head;      // last Linear + loss + their backward in one pass (rec_ctr_head_fwd_bwd)
};

int shape_of(const rec_deepfm_net* net, int64_t B, Shape* s) {
  REC_REQUIRE(net, REC_EINVAL, "net is NULL");
  REC_REQUIRE(net->num_slots > 0 && net->dim > 0 && net->dense_dim >= 0 && net->n_linear >= 1 &&
                  net->n_linear <= REC_DEEPFM_MAX_LINEAR, REC_EINVAL, "bad net sizes");
  REC_REQUIRE(B > 0 && B * net->num_slots < (1ll << 31), REC_EINVAL, "bad batch");
  REC_REQUIRE(net->widths[net->n_linear - 1] == 1, REC_EINVAL, "the last Linear must have one output (net.py:150)");
  s->S = net->num_slots; s->D = net->dim; s->Dn = net->dense_dim; s->n = net->n_linear;
  s->compact = s->Dn > 0 && s->Dn <= s->D;
  s->fp = s->compact ? s->S + 1 : s->S + s->Dn;
  s->in0_real = s->fp * s->D;
  s->pad = !s->compact && net->layer0_width > s->in0_real;
  REC_REQUIRE(net->layer0_width == 0 || s->pad || net->layer0_width == s->in0_real, REC_EINVAL,
              "layer0_width %d: a padded width needs a net without the dense fold and >= %d", net->layer0_width,
              s->in0_real);
  REC_REQUIRE(!s->pad || net->layer0_width % 4 == 0, REC_EINVAL, "layer0_width must be a multiple of 4");
  s->in0 = s->pad ? net->layer0_width : s->in0_real;
  s->in0_full = (s->S + s->Dn) * s->D;
  s->max_w = s->in0;
  for (int i = 0; i < s->n; ++i) {
    REC_REQUIRE(net->widths[i] > 0, REC_EINVAL, "bad layer width");
    if (net->widths[i] > s->max_w) s->max_w = net->widths[i];
  }
  static const bool small_env = [] { const char* v = getenv("REC_SMALL_MERGE"); return !(v && *v == '0'); }();
  s->small = small_env && B * s->S <= 15360;                      // ops.SMALL_MERGE_MAX
  static const bool head_env = [] { const char* v = getenv("REC_MLP_HEAD_FUSED"); return !(v && *v == '0'); }();
  const int nh = s->n > 1 ? net->widths[s->n - 2] : 0;            // input width of the head
  s->head = head_env && s->n > 1 && nh % 4 == 0 && nh <= 512 && B >= 64 &&      // ops._head_ok
            ((uintptr_t)net->w[s->n - 1]) % 16 == 0;
  static const bool ctr_env = [] { const char* v = getenv("REC_CTR_HEAD_FUSED"); return !(v && *v == '0'); }();
  s->ctr_head = ctr_env && s->n > 1 && nh % 4 == 0 && nh <= 512 && ((uintptr_t)net->w[s->n - 1]) % 16 == 0;   // ops.ctr_head_ok
  return REC_OK;
}

struct Buffers {
  float *y1, *y2, *feat, *sum_emb, *act[REC_DEEPFM_MAX_LINEAR], *y_dnn, *dz, *g[2], *row_grad, *dm;
  float *pp, *pp1, *dw0p;
  int32_t* small_scratch;
  int32_t *sorted_pos, *seg_offset, *n_uniq;
  int64_t* uniq_rows;
  void* ws;              // scratch of the individual calls (one at a time: a single region, the largest need)
  size_t ws_bytes;
  void* ws_head;         // the fused head's partial rows when their fold waits for the tail launch (tail_roles.h)
  size_t ws_head_bytes;
  void* ws_side;         // the grouping's scratch: it may run on the side stream beside the main stream's calls
  size_t ws_side_bytes;
  // bf16 x 3 images of the Linear weights, forward (W) and dX (W^T) orientation, made by ONE rec_gemm_b_images launch
  // at the top of the step (deepfm.py:_refresh_weights); NULL where the shape has no image form or the batch is short
  void* img_f[REC_DEEPFM_MAX_LINEAR];
  void* img_t[REC_DEEPFM_MAX_LINEAR];
  // relu_bits[i]: the ReLU mask of act[i] as bits (rec_gemm_epilogue_args.relu_bits: written by the forward GEMM of layer
  // i - 1, read by the dX GEMM of layer i), NULL where either call has no bit form (ops.mlp_forward / mlp_backward)
  void* relu_bits[REC_DEEPFM_MAX_LINEAR];
};

// the largest workspace any single call of the step asks for
int call_workspace(const rec_deepfm_net* net, const Shape& s, int64_t B, size_t* out) {
  size_t need = 0, b = 0;
  auto up = [&](size_t x) { if (x > need) need = x; };
  rec_gemm_desc d{};
  auto gemm_ws = [&](int64_t m, int n, int k, int ta, int tb) {
    d = rec_gemm_desc{m, n, k, 1, 1, n, ta, tb, REC_EPI_NONE, 0};
    d.lda = ta ? (int)m : k; d.ldb = tb ? k : n; d.ldc = n;
    if (rec_gemm_f32_workspace_bytes(&d, &b) == REC_OK) up(b);
  };
  int in = s.in0;
  for (int i = 0; i < s.n; ++i) {
    const int w = net->widths[i];
    gemm_ws(B, w, in, 0, 0);          // forward
    gemm_ws(B, in, w, 0, 1);          // dX
    gemm_ws(in, w, (int)B, 1, 0);     // dW (K = batch)
    in = w;
  }
  if (int rc = rec_logloss_workspace_bytes(B, &b)) return rc;
  up(b);
  rec_deepfm_desc fd{B, s.S, s.Dn, s.D, s.D, 1, -1, 1, s.compact ? 1 : 0};
  if (int rc = rec_deepfm_fm_bwd_workspace_bytes(&fd, &b)) return rc;
  up(b);
  if (s.head) {
    if (int rc = rec_mlp_head_bwd_workspace_bytes(B, net->widths[s.n - 2], &b)) return rc;
    up(b);
  }
  if (s.ctr_head) {
    if (int rc = rec_ctr_head_workspace_bytes(B, net->widths[s.n - 2], &b)) return rc;
    up(b);
  }
  *out = align_up(need, 256);
  return REC_OK;
}

int group_workspace(const rec_deepfm_net* net, const Shape& s, int64_t B, size_t* out) {
  size_t b = 0;
  if (net->slot_rows > 0) {
    if (int rc = rec_ids_group_slots_workspace_bytes(B, s.S, net->slot_rows, &b)) return rc;
  } else {
    if (int rc = rec_ids_group_workspace_bytes(B * s.S, net->num_rows, &b)) return rc;
  }
  *out = align_up(b, 256);
  return REC_OK;
}

int carve(const rec_deepfm_net* net, const Shape& s, int64_t B, void* workspace, Buffers* bf, size_t* total) {
  Carve c(workspace);
  const size_t n = (size_t)B * s.S;
  bf->y1 = c.take<float>(B);
  bf->y2 = c.take<float>(B);
  bf->feat = c.take<float>((size_t)B * s.in0);
  bf->sum_emb = c.take<float>((size_t)B * s.D);
  bf->act[0] = bf->feat;
  for (int i = 1; i < s.n; ++i) bf->act[i] = c.take<float>((size_t)B * net->widths[i - 1]);
  bf->y_dnn = c.take<float>(B);
  bf->dz = c.take<float>(B);
  bf->g[0] = c.take<float>((size_t)B * s.max_w);
  bf->g[1] = c.take<float>((size_t)B * s.max_w);
  bf->row_grad = c.take<float>(n * s.D);
  bf->dm = c.take<float>((size_t)(s.Dn > 0 ? s.Dn : 1) * net->widths[0]);
  bf->dw0p = (s.pad || s.compact) ? c.take<float>((size_t)s.in0 * net->widths[0]) : nullptr;
  bf->small_scratch = c.take<int32_t>(1);
  bf->pp = bf->pp1 = nullptr;
  bf->sorted_pos = bf->seg_offset = bf->n_uniq = nullptr;
  bf->uniq_rows = nullptr;
  bf->ws_side = nullptr;
  bf->ws_side_bytes = 0;
  if (!s.small) {
    size_t pb = 0, pb1 = 0;
    if (int rc = rec_segment_partials_bytes((int64_t)n, s.D, &pb)) return rc;
    if (int rc = rec_segment_partials_bytes((int64_t)n, 1, &pb1)) return rc;
    bf->pp = (float*)c.bytes(pb > 4 ? pb : 4);
    bf->pp1 = (float*)c.bytes(pb1 > 4 ? pb1 : 4);
    bf->sorted_pos = c.take<int32_t>(n);
    bf->uniq_rows = c.take<int64_t>(n);
    bf->seg_offset = c.take<int32_t>(n + 1);
    bf->n_uniq = c.take<int32_t>(4);
    size_t gw = 0;
    if (int rc = group_workspace(net, s, B, &gw)) return rc;
    bf->ws_side = c.bytes(gw);
    bf->ws_side_bytes = gw;
  }
  static const bool img_env = [] { const char* v = getenv("REC_GEMM_IMAGES"); return !(v && *v == '0'); }();
  {
    int in = s.in0;
    for (int i = 0; i < s.n; ++i) {
      bf->img_f[i] = bf->img_t[i] = nullptr;
      const int w = net->widths[i];
      if (img_env && B >= 8192) {                              // deepfm.py: use_images
        int32_t ok = 0;
        size_t ib = 0;
        if (rec_gemm_b_image_bytes(in, w, &ok, &ib) == REC_OK && ok) bf->img_f[i] = c.bytes(ib);
        if (rec_gemm_b_image_bytes(w, in, &ok, &ib) == REC_OK && ok) bf->img_t[i] = c.bytes(ib);
      }
      in = w;
    }
  }
  static const bool bits_env = [] { const char* v = getenv("REC_RELU_BITS"); return !(v && *v == '0'); }();
  for (int i = 0; i < s.n; ++i) {
    bf->relu_bits[i] = nullptr;
    if (i == 0 || !bits_env) continue;
    // act[i] = ReLU output of layer i - 1 ([B, widths[i-1]], K = its input width) and mask of the dX GEMM of layer i
    const int wi = net->widths[i - 1], kin = i == 1 ? s.in0 : net->widths[i - 2];
    rec_gemm_desc df{B, wi, kin, kin, wi, wi, 0, 0, REC_EPI_BIAS_RELU, 0, 0};
    rec_gemm_desc db{B, wi, net->widths[i], net->widths[i], net->widths[i], wi, 0, 1, REC_EPI_RELU_MASK, 0, 0};
    int32_t okf = 0, okb = 0;
    size_t nf = 0, nb = 0;
    if (rec_gemm_relu_bits_bytes(&df, &okf, &nf) == REC_OK && rec_gemm_relu_bits_bytes(&db, &okb, &nb) == REC_OK && okf && okb &&
        nf == nb)
      bf->relu_bits[i] = c.bytes(nf);
  }
  size_t cw = 0;
  if (int rc = call_workspace(net, s, B, &cw)) return rc;
  bf->ws = c.bytes(cw);
  bf->ws_bytes = cw;
  bf->ws_head = nullptr;
  bf->ws_head_bytes = 0;
  if (s.small && s.ctr_head) {
    size_t hb = 0;
    if (int rc = rec_ctr_head_workspace_bytes(B, net->widths[s.n - 2], &hb)) return rc;
    bf->ws_head = c.bytes(hb);
    bf->ws_head_bytes = hb;
  }
  *total = c.off;
  return REC_OK;
}

// C = epi(op(A) @ op(B)) on contiguous operands (the step's own buffers and parameter views)
int gemm(int64_t m, int n, int k, bool ta, bool tb, int epi, const float* A, const float* Bm, float* C,
         const float* bias, const float* aux0, int ld0, float* b_colsum, int split_k, const Buffers& bf, void* st,
         const void* b_image = nullptr, int num_cus = 0, void* relu_bits = nullptr) {
  rec_gemm_desc d{};
  d.num_cus = num_cus;
  d.m = m; d.n = n; d.k = k;
  d.lda = ta ? (int)m : k;
  d.ldb = tb ? k : n;
  d.ldc = n;
  d.trans_a = ta; d.trans_b = tb; d.epilogue = epi; d.split_k = split_k;
  rec_gemm_epilogue_args x{};
  x.bias = bias; x.aux0 = aux0; x.ld_aux0 = ld0; x.b_colsum = b_colsum; x.b_image = b_image; x.relu_bits = relu_bits;
  return rec_gemm_f32(&d, A, Bm, C, &x, bf.ws, bf.ws_bytes, st);
}

// the backward of one Linear in one call (ops.linear_backward): dW = X^T G + db, and dX = G W^T (+ ReLU' by `mask`)
int linear_backward(int64_t B, int in, int w, const float* X, const float* G, const float* W, float* dW, float* db,
                    float* dX, const float* mask, const Buffers& bf, void* st, const void* b_image_t,
                    void* mask_bits = nullptr) {
  rec_gemm_desc d0{}, d1{};
  d0.m = in; d0.n = w; d0.k = (int)B; d0.lda = in; d0.ldb = w; d0.ldc = w; d0.trans_a = 1; d0.epilogue = REC_EPI_NONE;
  d1.m = B; d1.n = in; d1.k = w; d1.lda = w; d1.ldb = w; d1.ldc = in; d1.trans_b = 1;
  d1.epilogue = mask ? REC_EPI_RELU_MASK : REC_EPI_NONE;
  rec_gemm_epilogue_args x0{}, x1{};
  x0.b_colsum = db;
  x1.aux0 = mask; x1.ld_aux0 = mask ? in : 0; x1.b_image = b_image_t; x1.relu_bits = mask ? mask_bits : nullptr;
  return rec_gemm_f32_pair(&d0, X, G, dW, &x0, &d1, G, W, dX, &x1, bf.ws, bf.ws_bytes, st);
}

}  // namespace

extern "C" int rec_deepfm_train_step_workspace_bytes(const rec_deepfm_net* net, int64_t batch, size_t* bytes) {
  REC_REQUIRE(bytes, REC_EINVAL, "bytes is NULL");
  Shape s;
  if (int rc = shape_of(net, batch, &s)) return rc;
  Buffers bf;
  return carve(net, s, batch, nullptr, &bf, bytes);
}

#define REC_TRY(call)            \
  do {                           \
    if (int rc_ = (call)) return rc_; \
  } while (0)

// Fork / join events of the two-stream schedule, one set per (device, stream, side stream): an event belongs to the
// device it was created on, and two host threads stepping two nets (distinct streams, as the contract asks) must not
// re-record each other's events between a record and its wait.  Created on first use under a lock, kept for the life of
// the process (a handful of entries: one per stream pair a caller ever steps on).
int rec::step_events(void* stream, void* side_stream, hipEvent_t** out) {
  struct Set { hipEvent_t ev[kStepEvents]; };
  static std::mutex mu;
  static std::map<std::tuple<int, void*, void*>, Set> cache;
  int dev = 0;
  REC_REQUIRE(hipGetDevice(&dev) == hipSuccess, REC_EHIP, "hipGetDevice failed");
  std::lock_guard<std::mutex> lock(mu);
  auto key = std::make_tuple(dev, stream, side_stream);
  auto it = cache.find(key);
  if (it == cache.end()) {
    Set st{};
    for (auto& e : st.ev)
      REC_REQUIRE(hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess, REC_EHIP, "hipEventCreate failed");
    it = cache.emplace(key, st).first;
  }
  *out = it->second.ev;                                      // std::map nodes do not move
  return REC_OK;
}

extern "C" int rec_deepfm_train_step(const rec_deepfm_net* net, int64_t batch, const int64_t* ids, const float* dense,
                                     const int64_t* label, const rec_adam_hyper* hyper, int64_t* auc_pos,
                                     int64_t* auc_neg, int32_t num_thresholds, float* loss_out, float* pred_out,
                                     int32_t* status, void* workspace, size_t workspace_bytes, void* stream,
                                     void* side_stream) {
  Shape s;
  if (int rc = shape_of(net, batch, &s)) return rc;
  REC_REQUIRE(ids && label && hyper && loss_out && pred_out && (dense || s.Dn == 0), REC_EINVAL,
              "null pointer argument");
  REC_REQUIRE(net->rec && net->mv && net->flat_param && net->flat_grad && net->flat_m && net->flat_v, REC_EINVAL,
              "net: null parameter pointer");
  REC_REQUIRE(!(s.compact || s.pad) || net->w0_folded, REC_EINVAL,
              "net: w0_folded is needed with 0 < dense_dim <= dim and with a padded layer0_width");
  REC_REQUIRE((auc_pos == nullptr) == (auc_neg == nullptr), REC_EINVAL, "auc_pos / auc_neg: both or neither");
  Buffers bf;
  size_t need = 0;
  if (int rc = carve(net, s, batch, workspace, &bf, &need)) return rc;
  REC_REQUIRE(workspace && workspace_bytes >= need, REC_EWORKSPACE, "workspace %zu < %zu", workspace_bytes, need);
  const int64_t B = batch;
  const int S = s.S, D = s.D, Dn = s.Dn, n = s.n;
  const size_t f4 = sizeof(float);

  // -- the mirror's large-batch schedule (deepfm.py:train_step, overlap): with a side stream the id grouping is forked
  //    BEFORE the lookup and runs beside the HBM-bound FM kernels / under the forward GEMMs, and the sparse update runs
  //    on it underneath the MFMA-bound dW_0 GEMM; without one (or when the merge happens inside the record update)
  //    everything is issued on `stream` in the same order
  const bool overlap = side_stream != nullptr && side_stream != stream && !s.small;
  void* sst = overlap ? side_stream : stream;
  hipEvent_t* ev = nullptr;                                  // fork, after fm_bwd, join: re-recorded every step
  if (overlap) REC_TRY(step_events(stream, side_stream, &ev));
  auto order = [&](int k, void* from, void* to) -> int {    // everything issued on `from` so far happens before `to` goes on
    REC_REQUIRE(hipEventRecord(ev[k], (hipStream_t)from) == hipSuccess &&
                    hipStreamWaitEvent((hipStream_t)to, ev[k], 0) == hipSuccess, REC_EHIP, "stream ordering failed");
    return REC_OK;
  };
  const int64_t nlook = B * S;
  if (!s.small) {                                            // SelectedRows merge keys: they only depend on the ids
    if (overlap) REC_TRY(order(0, stream, sst));
    if (net->slot_rows > 0) {
      REC_TRY(rec_ids_group_slots(B, S, net->slot_rows, net->padding_idx, ids, bf.sorted_pos, bf.uniq_rows,
                                  bf.seg_offset, bf.n_uniq, nullptr, status, bf.ws_side, bf.ws_side_bytes, sst));
    } else {
      REC_TRY(rec_ids_group_payload(nlook, S, net->num_rows, net->padding_idx, ids, net->slot_offset, nullptr,
                                    bf.sorted_pos, bf.uniq_rows, bf.seg_offset, bf.n_uniq, status, bf.ws_side,
                                    bf.ws_side_bytes, sst));
    }
  }
  // -- FM: lookup + first / second order + the MLP's input (net.py:104-136)
  rec_deepfm_desc fd{B, S, Dn, D, net->rec_stride, net->table_rows, net->padding_idx, net->rec_stride,
                     s.compact ? 1 : 0, s.pad ? (int64_t)s.in0 : 0};
  // (launch-bound sizes: layer 0's weight fold rides behind the lookup's blocks — tail_roles.h, FoldFwd — and the
  //  narrow-row lookup zeroes the padding columns of its samples itself)
  static const bool fold_ride = [] { const char* v = getenv("REC_SMALL_TAIL"); return !(v && *v == '0'); }();
  bool fold_rode = false, pad_zeroed = false;
  FoldFwd ff{};
  if (s.compact && s.small && fold_ride)
    ff = dense_fold_fwd_plan(S, Dn, D, net->widths[0], net->dense_w, net->w[0], net->w0_folded);
  ff.zero_feat_pad = s.pad && s.small && fold_ride ? 1 : 0;
  REC_TRY(deepfm_fm_fwd_fold(&fd, ids, dense, net->rec, net->rec + D, net->dense_w, net->dense_w_one, net->slot_offset,
                             bf.y1, bf.y2, bf.feat, bf.sum_emb, status, stream, ff, &fold_rode, &pad_zeroed));
  if (s.pad && !pad_zeroed)   // the padding columns of feat (workspace memory) must be zero: they meet zero weight rows, but 0 x NaN
    REC_REQUIRE(hipMemset2DAsync(bf.feat + s.in0_real, (size_t)s.in0 * f4, 0, (size_t)(s.in0 - s.in0_real) * f4,
                                 (size_t)B, (hipStream_t)stream) == hipSuccess, REC_EHIP, "hipMemset2DAsync failed");
  // -- layer 0 on folded weights (deepfm.py:_mlp_weights): W0' = [ W0[:S*D] ; M ; 0 ]
  const float* w0 = net->w[0];
  float* gw0 = net->gw[0];
  if (s.compact) {  // sparse rows + folded dense rows in one launch; dW_0' lands in scratch (bf.dw0p), not in gw[0]
    if (!fold_rode)
      REC_TRY(rec_dense_fold_fwd_full(S, Dn, D, net->widths[0], net->dense_w, net->w[0], net->w0_folded, stream));
    w0 = net->w0_folded;
    gw0 = bf.dw0p;
  }
  // layer 0 on its weight with zero rows behind it (deepfm.py:_mlp_weights, padded): in place when the caller's
  // parameter and gradient buffers hold the rows (w0_folded == w[0]), else on a refreshed copy
  const bool pad_copy = s.pad && net->w0_folded != net->w[0];
  if (pad_copy) {
    REC_TRY(rec_copy_async(net->w0_folded, net->w[0], (size_t)s.in0_real * net->widths[0] * f4, stream));
    w0 = net->w0_folded;
    gw0 = bf.dw0p;
  }
  // -- top MLP forward (net.py:142-174): bias / ReLU in the GEMM epilogue; with the fused head the last Linear, the loss
  //    and the backward of both are ONE pass over the last hidden activation (deepfm.py: fused_head)
  const int n_fwd = s.ctr_head ? n - 1 : n;
  {   // every weight image of the step in one launch (the mirror's _refresh_weights(images)): the GEMMs below skip their
      // own split launches
    rec_gemm_b_image items[2 * REC_DEEPFM_MAX_LINEAR];
    int cnt = 0, in = s.in0;
    for (int i = 0; i < n; ++i) {
      const int w = net->widths[i];
      const float* wi = i == 0 ? w0 : net->w[i];
      if (bf.img_f[i]) items[cnt++] = rec_gemm_b_image{wi, w, in, w, 0, bf.img_f[i]};
      if (bf.img_t[i]) items[cnt++] = rec_gemm_b_image{wi, w, w, in, 1, bf.img_t[i]};
      in = w;
    }
    if (cnt) REC_TRY(rec_gemm_b_images(cnt, items, stream));
  }
  {
    int in = s.in0;
    for (int i = 0; i < n_fwd; ++i) {
      const bool last = i == n - 1;
      float* out = last ? bf.y_dnn : bf.act[i + 1];
      REC_TRY(gemm(B, net->widths[i], in, false, false, last ? REC_EPI_BIAS : REC_EPI_BIAS_RELU, bf.act[i],
                   i == 0 ? w0 : net->w[i], out, net->b[i], nullptr, 0, nullptr, 0, bf, stream, bf.img_f[i], 0,
                   last ? nullptr : bf.relu_bits[i + 1]));
      in = net->widths[i];
    }
  }
  const float* g = bf.dz;       // gradient of the current layer's OUTPUT, [B, widths[i]]
  int gi = 0;                   // next free ping-pong buffer
  int n_run = n;
  // -- launch-bound sizes: the step's folds and its dense Adam as roles of the row update's launch (tail_roles.h).  The
  //    roles address parameters by their offset in the flat buffer: every parameter a role owns must sit at the same
  //    offset in flat_param and flat_grad (the mirror's FlatParams does), else the mirror's list of launches is issued
  const float* P0 = net->flat_param;
  const float* G0 = net->flat_grad;
  auto flat_pair = [&](const float* pp, const float* gg, int64_t cnt) {
    return pp && gg && pp >= P0 && pp + cnt <= P0 + net->flat_numel && pp - P0 == gg - G0;
  };
  const bool pad_copy_ = s.pad && net->w0_folded != net->w[0];
  bool tail = s.small && s.ctr_head && !pad_copy_ && small_tail_eligible(nlook, D, net->slot_offset, S) &&
              flat_pair(net->w[n - 1], net->gw[n - 1], net->widths[n - 2]) && flat_pair(net->b[n - 1], net->gb[n - 1], 1) &&
              flat_pair(net->w[0], net->gw[0], (int64_t)s.in0_full * net->widths[0]) &&
              (Dn == 0 || (flat_pair(net->dense_w, net->g_dense_w, (int64_t)Dn * D) &&
                           flat_pair(net->dense_w_one, net->g_dense_w_one, Dn)));
  int head_nblk = 0;
  float head_inv = 0.f;
  if (s.ctr_head) {
    const int nh = net->widths[n - 2];
    if (tail)
      REC_TRY(ctr_head_fwd_bwd_partial(B, nh, 0, bf.act[n - 1], nh, net->w[n - 1], net->b[n - 1], bf.y1, bf.y2, label, 1e-4f,
                                       0.f, 0.f, 1, nullptr, pred_out, bf.dz, bf.g[gi], nh, bf.ws_head, bf.ws_head_bytes,
                                       stream, &head_nblk, &head_inv));
    else
    REC_TRY(rec_ctr_head_fwd_bwd(B, nh, 0, bf.act[n - 1], nh, net->w[n - 1], net->b[n - 1], bf.y1, bf.y2, label, 1e-4f,
                                 0.f, 0.f, 1, nullptr, pred_out, bf.dz, loss_out, bf.g[gi], nh, net->gw[n - 1],
                                 net->gb[n - 1], bf.ws, bf.ws_bytes, stream));
    g = bf.g[gi];
    gi ^= 1;
    n_run = n - 1;
  } else {
    // -- sigmoid + log_loss (dygraph_model.py:76-85)
    REC_TRY(rec_sigmoid_logloss(B, 0, bf.y1, bf.y2, bf.y_dnn, label, 1e-4f, 0.f, 0.f, pred_out, bf.dz, loss_out, bf.ws,
                                bf.ws_bytes, stream));
  }
  // -- the AUC buckets (create_metrics)
  if (auc_pos) REC_TRY(rec_auc_histogram(B, pred_out, label, num_thresholds, auc_pos, auc_neg, stream));
  // -- MLP backward (ops.mlp_backward, defer_first): dX of layer 0 before its dW
  if (!s.ctr_head && s.head) {
    const int nh = net->widths[n - 2];
    REC_TRY(rec_mlp_head_bwd(B, nh, bf.act[n - 1], nh, bf.dz, net->w[n - 1], 1, bf.g[gi], nh, net->gw[n - 1],
                             net->gb[n - 1], bf.ws, bf.ws_bytes, stream));
    g = bf.g[gi];
    gi ^= 1;
    n_run = n - 1;
  }
  const float* g0 = nullptr;
  float* d_flat = nullptr;
  bool dw0_done = false;
  for (int i = n_run - 1; i >= 0; --i) {
    const int in = i == 0 ? s.in0 : net->widths[i - 1], w = net->widths[i];
    if (i == 0) {
      g0 = g;
      d_flat = bf.g[gi];
      if (s.small) {          // no second stream at these sizes: dW_0 beside dX_0 instead of behind fm_bwd (the mirror's
        dw0_done = true;      // mlp_backward without defer_first)
        REC_TRY(linear_backward(B, in, w, bf.feat, g, w0, gw0, net->gb[0], d_flat, nullptr, bf, stream, bf.img_t[0]));
        break;
      }
      REC_TRY(gemm(B, in, w, false, true, REC_EPI_NONE, g, w0, d_flat, nullptr, nullptr, 0, nullptr, 0, bf, stream,
                   bf.img_t[0]));
      break;
    }
    REC_TRY(linear_backward(B, in, w, bf.act[i], g, net->w[i], net->gw[i], net->gb[i], bf.g[gi], bf.act[i], bf, stream,
                            bf.img_t[i], bf.relu_bits[i]));
    g = bf.g[gi];
    gi ^= 1;
  }
  // -- FM backward: row gradients of the S lookups + the dense FM parameters (net.py:104-136 backward)
  rec_deepfm_desc bd{B, S, Dn, D, D, 1, -1, 1, s.compact ? 1 : 0, s.pad ? (int64_t)s.in0 : 0};   // reads no table
  int fm_nblk = 0;
  if (tail)
    REC_TRY(deepfm_fm_bwd_partial(&bd, dense, bf.feat, bf.sum_emb, d_flat, bf.dz, bf.dz, net->dense_w, bf.row_grad, bf.ws,
                                  bf.ws_bytes, stream, &fm_nblk));
  else
  REC_TRY(rec_deepfm_fm_bwd(&bd, dense, bf.feat, bf.sum_emb, d_flat, bf.dz, bf.dz, net->dense_w, bf.row_grad,
                            net->g_dense_w, net->g_dense_w_one, bf.ws, bf.ws_bytes, stream));
  // -- merged lazy Adam on W / W1 of the touched rows (optimizer.step on the SelectedRows gradients): on the side stream
  //    when there is one, underneath dW_0
  rec_grad_layout gl{1, 0, 0, nullptr, nullptr, 0}, gl1{S, 0, 0, nullptr, nullptr, 0};
  if (tail) {     // ... and, behind its row-bucket blocks, both folds, the folded layer 0's backward and the dense Adam
    TailRoles t{};
    const int nh = net->widths[n - 2], NO = net->widths[0];
    t.head_blocks = (nh + 2 + 15) / 16;
    t.head_nblk = head_nblk; t.head_n2 = nh + 2; t.head_partial = (const float*)bf.ws_head; t.head_invB = head_inv;
    t.head_dw = net->gw[n - 1]; t.head_db = net->gb[n - 1]; t.loss = loss_out;
    t.head_w_off = net->w[n - 1] - P0; t.head_b_off = net->b[n - 1] - P0;
    t.n_skip = 0;
    auto skip = [&](int64_t lo, int64_t cnt) { t.skip_lo[t.n_skip] = lo; t.skip_hi[t.n_skip] = lo + cnt; ++t.n_skip; };
    skip(t.head_w_off, nh);
    skip(t.head_b_off, 1);
    t.fm_blocks = Dn * D + Dn;
    t.fm_nblk = fm_nblk; t.fm_split = Dn * D; t.fm_partial = (const float*)bf.ws;
    t.ddw = net->g_dense_w; t.ddw1 = net->g_dense_w_one;
    if (Dn > 0) {
      t.dw_off = net->dense_w - P0; t.dw1_off = net->dense_w_one - P0;
      skip(t.dw_off, (int64_t)Dn * D);
      skip(t.dw1_off, Dn);
    }
    t.w0_off = net->w[0] - P0;
    t.f = FoldedLayer0{S, Dn, D, NO, nullptr, nullptr, nullptr, nullptr};
    t.w0_blocks = 0;
    if (s.compact) {
      t.f.dW0f = gw0; t.f.dW0 = net->gw[0]; t.f.W0 = net->w[0]; t.f.dense_w = net->dense_w;
      skip(t.w0_off, (int64_t)s.in0_full * NO);
      const int64_t we = (int64_t)S * D * NO;
      t.w0_blocks = (int)((we + 4 * 1024 - 1) / (4 * 1024));
    }
    t.flat_numel = net->flat_numel; t.flat_grad = net->flat_grad;
    t.adam.p = net->flat_param; t.adam.m = net->flat_m; t.adam.v = net->flat_v;
    t.rest_blocks = (int)((net->flat_numel + 4 * 1024 - 1) / (4 * 1024));
    if (t.rest_blocks > 256) t.rest_blocks = 256;
    REC_TRY(sparse_adam_record_small_tail(nlook, S, D, net->rec_stride, net->mv_stride, net->v_offset, net->table_rows,
                                          net->padding_idx, ids, net->slot_offset, bf.row_grad, &gl, bf.dz, &gl1, net->rec,
                                          net->mv, hyper, status, t, stream));
    return REC_OK;
  }
  if (s.small) {
    REC_TRY(rec_sparse_adam_record_small(nlook, S, D, net->rec_stride, net->mv_stride, net->v_offset, net->table_rows,
                                         net->padding_idx, ids, net->slot_offset, bf.row_grad, &gl, bf.dz, &gl1,
                                         nullptr, net->rec, net->mv, hyper, status, bf.small_scratch, stream));
  } else {
    if (overlap) REC_TRY(order(1, stream, sst));
    REC_TRY(rec_segment_partials(nlook, D, bf.n_uniq, bf.seg_offset, bf.sorted_pos, bf.row_grad, &gl, bf.pp, sst));
    REC_TRY(rec_segment_partials(nlook, 1, bf.n_uniq, bf.seg_offset, bf.sorted_pos, bf.dz, &gl1, bf.pp1, sst));
    gl.partials = bf.pp;
    gl1.partials = bf.pp1;
    REC_TRY(rec_sparse_adam_record(nlook, D, net->rec_stride, net->mv_stride, net->v_offset, bf.n_uniq, bf.uniq_rows,
                                   bf.seg_offset, bf.sorted_pos, bf.row_grad, &gl, bf.dz, &gl1, nullptr, net->rec,
                                   net->mv, hyper, sst));
  }
  // -- dW_0 / db_0, then the folded rows back into the real parameters' gradients (deepfm.py:_fold_backward).  Beside the
  //    HBM-bound update the mirror splits K 16 ways instead of the planner's 32: half a resident round of blocks
  static const int dw0_split = [] { const char* v = getenv("REC_DW0_SPLIT"); return v && *v ? atoi(v) : 16; }();
  static const int dw0_cus = [] { const char* v = getenv("REC_DW0_CUS"); return v && *v ? atoi(v) : 192; }();   // deepfm.py
  if (!dw0_done)
    REC_TRY(gemm(s.in0, net->widths[0], (int)B, true, false, REC_EPI_NONE, bf.feat, g0, gw0, nullptr, nullptr, 0,
                 net->gb[0], overlap && B >= 16384 ? dw0_split : 0, bf, stream, nullptr,
                 overlap && B >= 16384 ? dw0_cus : 0));
  if (s.compact)
    REC_TRY(rec_dense_fold_bwd_full(S, Dn, D, net->widths[0], net->dense_w, net->w[0], gw0, net->gw[0],
                                    net->g_dense_w, 1, stream));
  if (pad_copy) REC_TRY(rec_copy_async(net->gw[0], gw0, (size_t)s.in0_real * net->widths[0] * f4, stream));
  // -- Adam on every dense parameter (one pass over the flat buffer)
  REC_TRY(rec_adam_dense(net->flat_numel, net->flat_param, net->flat_m, net->flat_v, net->flat_grad, nullptr, hyper,
                         stream));
  if (overlap) REC_TRY(order(2, sst, stream));
  return REC_OK;
}
