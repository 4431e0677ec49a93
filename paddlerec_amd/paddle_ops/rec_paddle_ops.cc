// rec_paddle_ops.cc — the engine's fused kernels as Paddle custom C++ operators (north_star:
// "exposed as Paddle custom C++ ops through a thin C-ABI").
//
// Written against Paddle's public custom-op API only (`paddle/extension.h`: paddle::Tensor, PD_BUILD_OP / PD_BUILD_GRAD_OP,
// PD_KERNEL, PD_INFER_SHAPE / PD_INFER_DTYPE, PD_CHECK).  With a real PaddlePaddle the file is loaded by
//     paddle.utils.cpp_extension.load(name="rec_ops", sources=["rec_paddle_ops.cc"],
//                                     extra_include_paths=[".../include"], extra_ldflags=["-L...", "-lrecengine"])
// Here (no Paddle installable) it is compiled by paddlerec_amd/build.py against the executable stand-in header
// paddlerec_amd/paddle_ops/mock/paddle/extension.h into paddlerec_amd/paddle_ops/librec_paddle_ops.so, and the compat
// namespace's paddle.utils.cpp_extension.load() drives the registered kernels: the reference's unmodified
// tools/trainer.py, on a net.py patched by integration/*.patch, reaches fm_fwd_kernel / fm_bwd_kernel & co through
// exactly these functions.  Every operator is a translation of tensors to pointers + sizes around ONE C-ABI entry point
// per direction (include/recengine.h); the shim owns no kernel and no device code (built by the host compiler).
//
// Sparse tables.  Paddle's custom-op interface carries dense tensors only — an operator cannot emit the SelectedRows a
// `sparse=True` embedding produces (deepfm/net.py:62-70,80).  The gradient of a table input is therefore returned in
// rows form: Grad("W") is the SelectedRows VALUE, one gradient row per lookup, whose `rows` are the operator's id input
// (named in the REC_SELECTED_ROWS note below).  The compat loader re-attaches the ids and hands the pair to the sparse
// optimizer kernels (rec_ids_group + rec_sparse_adam_rows / rec_adam_rows_all / rec_sparse_sgd_rows); a binder inside
// real Paddle does the same in a 5-line PyLayer or feeds `rec_sparse_*` directly.
#include <cstdint>
#include <vector>

#include "paddle/extension.h"
#include "recengine.h"

#ifdef PD_MOCK_EXTENSION_H
// table input `T` receives a rows-form gradient whose rows are the flattened ids of input `I` (metadata for the loader)
#define REC_SELECTED_ROWS(T, I) .Note("selected_rows=" T ":" I)
#else
#define REC_SELECTED_ROWS(T, I)
#endif

#define REC_CALL(expr) PD_CHECK((expr) == REC_OK, #expr ": ", rec_last_error())

namespace {

using paddle::DataType;
using paddle::Tensor;
using Shape = std::vector<int64_t>;

inline void want(const Tensor& t, DataType dt, size_t rank, const char* name) {
  PD_CHECK(t.dtype() == dt, name, ": wrong dtype");
  PD_CHECK(t.shape().size() == rank, name, ": rank ", t.shape().size(), ", expected ", rank);
}
inline Tensor workspace(size_t bytes, const Tensor& like) {          // caller-owned scratch (the engine never allocates)
  return paddle::empty({static_cast<int64_t>(bytes ? bytes : 1)}, DataType::UINT8, like.place());
}

// =====================================================================================================================
// rec_deepfm_fm — replaces models/rank/deepfm/net.py:105-139 (FM.forward after the concat): both embedding lookups,
// the first-order sum, the dense "embeddings", concat, and the second-order term, in one kernel (fm_fwd_kernel).
//   Ids [B,S] i64 (= paddle.concat(sparse_inputs, 1)), Dense [B,Dn], W [N,D] (fm.embedding.weight),
//   W1 [N,1] (fm.embedding_one.weight), DenseW [1,Dn,D], DenseWOne [Dn]
//   -> Y1 [B,1], Y2 [B,1], FeatEmb [B,S+Dn,D]; SumEmb [B,D] and Status [1] i32 are kept for the gradient / the caller.
// =====================================================================================================================
rec_deepfm_desc fm_desc(const Shape& ids, const Shape& dense, const Shape& w, int64_t padding_idx) {
  rec_deepfm_desc d;
  d.batch = ids[0];
  d.num_slots = static_cast<int32_t>(ids[1]);
  d.num_dense = static_cast<int32_t>(dense[1]);
  d.emb_dim = static_cast<int32_t>(w[1]);
  d.row_stride = d.emb_dim;          // Paddle parameters: W and W1 are separate dense [N,D] / [N,1] tensors
  d.num_rows = w[0];
  d.padding_idx = padding_idx;
  d.w1_stride = 1;
  d.compact_dense = 0;
  d.feat_stride = 0;
  return d;
}

std::vector<Tensor> RecDeepFmFwd(const Tensor& ids, const Tensor& dense, const Tensor& w, const Tensor& w1,
                                 const Tensor& dense_w, const Tensor& dense_w_one, int64_t padding_idx) {
  want(ids, DataType::INT64, 2, "Ids");
  want(dense, DataType::FLOAT32, 2, "Dense");
  want(w, DataType::FLOAT32, 2, "W");
  PD_CHECK(w1.dtype() == DataType::FLOAT32 && w1.numel() == w.shape()[0], "W1 must hold one float per row of W");
  PD_CHECK(dense.shape()[0] == ids.shape()[0], "Ids / Dense batch mismatch");
  const rec_deepfm_desc d = fm_desc(ids.shape(), dense.shape(), w.shape(), padding_idx);
  PD_CHECK(dense_w.numel() == static_cast<int64_t>(d.num_dense) * d.emb_dim && dense_w_one.numel() == d.num_dense,
           "DenseW / DenseWOne shape");
  const int64_t B = d.batch;
  auto y1 = paddle::empty({B, 1}, DataType::FLOAT32, ids.place());
  auto y2 = paddle::empty({B, 1}, DataType::FLOAT32, ids.place());
  auto feat = paddle::empty({B, d.num_slots + d.num_dense, d.emb_dim}, DataType::FLOAT32, ids.place());
  auto sum_emb = paddle::empty({B, d.emb_dim}, DataType::FLOAT32, ids.place());
  auto status = paddle::full({1}, 0, DataType::INT32, ids.place());
  REC_CALL(rec_deepfm_fm_fwd(&d, ids.data<int64_t>(), dense.data<float>(), w.data<float>(), w1.data<float>(),
                             dense_w.data<float>(), dense_w_one.data<float>(), /*slot_offset=*/nullptr, y1.data<float>(),
                             y2.data<float>(), feat.data<float>(), sum_emb.data<float>(), status.data<int32_t>(),
                             ids.stream()));
  return {y1, y2, feat, sum_emb, status};
}

std::vector<Shape> RecDeepFmInferShape(const Shape& ids, const Shape& dense, const Shape& w, const Shape& w1,
                                       const Shape& dense_w, const Shape& dense_w_one, int64_t padding_idx) {
  const int64_t B = ids[0], S = ids[1], Dn = dense[1], D = w[1];
  return {{B, 1}, {B, 1}, {B, S + Dn, D}, {B, D}, {1}};
}
std::vector<DataType> RecDeepFmInferDtype(DataType ids, DataType dense, DataType w, DataType w1, DataType dense_w,
                                          DataType dense_w_one) {
  return {dense, dense, dense, dense, DataType::INT32};
}

// gradient of the block (what loss.backward(), tools/trainer.py:151, runs for net.py:105-139): fm_bwd_kernel.
//   Grad(W)  = SelectedRows value [B*S, D], rows = Ids flattened (padding rows are dropped by the optimizer's merge)
//   Grad(W1) = SelectedRows value of embedding_one: dy1[b] for every (b, s) — returned as [B,1]; one gradient row serves
//              the S lookups of a sample (rec_grad_layout{div = S})
std::vector<Tensor> RecDeepFmBwd(const Tensor& ids, const Tensor& dense, const Tensor& feat, const Tensor& sum_emb,
                                 const Tensor& dense_w, const Tensor& d_feat, const Tensor& dy1, const Tensor& dy2) {
  const Shape fs = feat.shape();
  const int64_t B = fs[0];
  const int32_t S = static_cast<int32_t>(ids.shape()[1]), D = static_cast<int32_t>(fs[2]);
  const int32_t Dn = static_cast<int32_t>(fs[1]) - S;
  PD_CHECK(d_feat.numel() == feat.numel() && dy1.numel() == B && dy2.numel() == B, "gradient shapes");
  rec_deepfm_desc d = fm_desc(ids.shape(), dense.shape(), {1, D}, -1);
  auto row_grad = paddle::empty({B * S, D}, DataType::FLOAT32, feat.place());
  auto row_grad1 = paddle::empty({B, 1}, DataType::FLOAT32, feat.place());
  auto d_dense_w = paddle::empty({1, Dn, D}, DataType::FLOAT32, feat.place());
  auto d_dense_w_one = paddle::empty({Dn}, DataType::FLOAT32, feat.place());
  size_t ws_bytes = 0;
  REC_CALL(rec_deepfm_fm_bwd_workspace_bytes(&d, &ws_bytes));
  auto ws = workspace(ws_bytes, feat);
  REC_CALL(rec_deepfm_fm_bwd(&d, dense.data<float>(), feat.data<float>(), sum_emb.data<float>(), d_feat.data<float>(),
                             dy1.data<float>(), dy2.data<float>(), dense_w.data<float>(), row_grad.data<float>(),
                             d_dense_w.data<float>(), d_dense_w_one.data<float>(), ws.data<uint8_t>(), ws_bytes,
                             feat.stream()));
  REC_CALL(rec_copy_async(row_grad1.data<float>(), dy1.data<float>(), static_cast<size_t>(B) * sizeof(float),
                          feat.stream()));
  return {row_grad, row_grad1, d_dense_w, d_dense_w_one};
}

std::vector<Shape> RecDeepFmBwdInferShape(const Shape& ids, const Shape& dense, const Shape& feat, const Shape& sum_emb,
                                          const Shape& dense_w, const Shape& d_feat, const Shape& dy1, const Shape& dy2) {
  const int64_t B = ids[0], S = ids[1], D = feat[2], Dn = feat[1] - S;
  return {{B * S, D}, {B, 1}, {1, Dn, D}, {Dn}};
}

// =====================================================================================================================
// rec_crossnet_v2_layer — one layer of CrossNetV2.forward (models/rank/dcn_v2/net.py:222-226):
//   X_{l+1} = X_l + X_0 * (X_l W_l + b_l);  W [d,d] (Paddle [in,out]), B [d].  U = X_l W + b is kept for the gradient.
// =====================================================================================================================
std::vector<Tensor> RecCrossV2Fwd(const Tensor& x0, const Tensor& xl, const Tensor& w, const Tensor& b) {
  want(x0, DataType::FLOAT32, 2, "X0");
  want(xl, DataType::FLOAT32, 2, "Xl");
  const int64_t B = xl.shape()[0];
  const int32_t d = static_cast<int32_t>(xl.shape()[1]);
  PD_CHECK(x0.shape() == xl.shape() && w.numel() == static_cast<int64_t>(d) * d && b.numel() == d, "CrossNetV2 shapes");
  auto out = paddle::empty({B, d}, DataType::FLOAT32, xl.place());
  auto u = paddle::empty({B, d}, DataType::FLOAT32, xl.place());
  rec_crossnet_v2_desc desc{B, d, 0, 0, 0, 0};
  size_t fwd = 0, bwd = 0;
  REC_CALL(rec_crossnet_v2_layer_workspace_bytes(&desc, &fwd, &bwd));
  auto ws = workspace(fwd, xl);
  REC_CALL(rec_crossnet_v2_layer_fwd(&desc, x0.data<float>(), xl.data<float>(), w.data<float>(), b.data<float>(),
                                     out.data<float>(), u.data<float>(), ws.data<uint8_t>(), fwd, xl.stream()));
  return {out, u};
}
std::vector<Shape> RecCrossV2InferShape(const Shape& x0, const Shape& xl, const Shape& w, const Shape& b) {
  return {xl, xl};
}
std::vector<DataType> RecCrossV2InferDtype(DataType x0, DataType xl, DataType w, DataType b) { return {xl, xl}; }

std::vector<Tensor> RecCrossV2Bwd(const Tensor& x0, const Tensor& xl, const Tensor& w, const Tensor& u,
                                  const Tensor& d_out) {
  const int64_t B = xl.shape()[0];
  const int32_t d = static_cast<int32_t>(xl.shape()[1]);
  PD_CHECK(d_out.numel() == xl.numel(), "Grad(Out) shape");
  auto dx0 = paddle::empty({B, d}, DataType::FLOAT32, xl.place());
  auto dxl = paddle::empty({B, d}, DataType::FLOAT32, xl.place());
  auto dw = paddle::empty({d, d}, DataType::FLOAT32, xl.place());
  auto db = paddle::empty({d}, DataType::FLOAT32, xl.place());
  rec_crossnet_v2_desc desc{B, d, 0, 0, 0, 0};
  size_t fwd = 0, bwd = 0;
  REC_CALL(rec_crossnet_v2_layer_workspace_bytes(&desc, &fwd, &bwd));
  auto ws = workspace(bwd, xl);
  // the operator returns this layer's share of d X_0; the framework's autograd sums the layers' shares
  REC_CALL(rec_crossnet_v2_layer_bwd(&desc, x0.data<float>(), xl.data<float>(), w.data<float>(), u.data<float>(),
                                     d_out.data<float>(), 0, dx0.data<float>(), 0, /*accumulate_dx0=*/0, /*fold_dx0=*/0,
                                     dxl.data<float>(), 0, dw.data<float>(), db.data<float>(), ws.data<uint8_t>(), bwd,
                                     xl.stream()));
  return {dx0, dxl, dw, db};
}

// =====================================================================================================================
// rec_crossnet_mix_layer — one layer of CrossNetMix.forward (models/rank/dcn_v2/net.py:278-320), all experts:
//   x_{l+1} = x_l + sum_e softmax_e(x_l gate_w + gate_b) * x_0 * (tanh(tanh(x_l V_e) C_e^T) U_e^T + b)
//   U, V [E,d,r], C [E,r,r], Bias [d,1], GateW [d,E] (the E Linear(d,1).weight columns side by side), GateB [E]
// =====================================================================================================================
rec_crossnet_mix_desc mix_desc(const Shape& xl, const Shape& u) {
  rec_crossnet_mix_desc m;
  m.batch = xl[0];
  m.d = static_cast<int32_t>(xl[1]);
  m.experts = static_cast<int32_t>(u[0]);
  m.rank = static_cast<int32_t>(u[2]);
  m.ld_x0 = m.ld_xl = m.ld_out = 0;
  return m;
}
std::vector<Tensor> RecCrossMixFwd(const Tensor& x0, const Tensor& xl, const Tensor& u, const Tensor& v, const Tensor& c,
                                   const Tensor& bias, const Tensor& gate_w, const Tensor& gate_b) {
  want(xl, DataType::FLOAT32, 2, "Xl");
  want(u, DataType::FLOAT32, 3, "U");
  const rec_crossnet_mix_desc m = mix_desc(xl.shape(), u.shape());
  const int64_t B = m.batch, Er = static_cast<int64_t>(m.experts) * m.rank;
  PD_CHECK(x0.shape() == xl.shape() && u.shape()[1] == m.d && v.shape() == u.shape() &&
               c.numel() == Er * m.rank && bias.numel() == m.d && gate_w.numel() == static_cast<int64_t>(m.d) * m.experts &&
               gate_b.numel() == m.experts, "CrossNetMix shapes");
  auto out = paddle::empty({B, m.d}, DataType::FLOAT32, xl.place());
  auto t1 = paddle::empty({B, Er}, DataType::FLOAT32, xl.place());
  auto t2 = paddle::empty({B, Er}, DataType::FLOAT32, xl.place());
  auto prob = paddle::empty({B, m.experts}, DataType::FLOAT32, xl.place());
  size_t fwd = 0, bwd = 0;
  REC_CALL(rec_crossnet_mix_layer_workspace_bytes(&m, &fwd, &bwd));
  auto ws = workspace(fwd, xl);
  REC_CALL(rec_crossnet_mix_layer_fwd(&m, x0.data<float>(), xl.data<float>(), u.data<float>(), v.data<float>(),
                                      c.data<float>(), bias.data<float>(), gate_w.data<float>(), gate_b.data<float>(),
                                      out.data<float>(), t1.data<float>(), t2.data<float>(), prob.data<float>(),
                                      ws.data<uint8_t>(), fwd, xl.stream()));
  return {out, t1, t2, prob};
}
std::vector<Shape> RecCrossMixInferShape(const Shape& x0, const Shape& xl, const Shape& u, const Shape& v, const Shape& c,
                                         const Shape& bias, const Shape& gate_w, const Shape& gate_b) {
  return {xl, {xl[0], u[0] * u[2]}, {xl[0], u[0] * u[2]}, {xl[0], u[0]}};
}
std::vector<DataType> RecCrossMixInferDtype(DataType x0, DataType xl, DataType u, DataType v, DataType c, DataType bias,
                                            DataType gate_w, DataType gate_b) {
  return {xl, xl, xl, xl};
}
std::vector<Tensor> RecCrossMixBwd(const Tensor& x0, const Tensor& xl, const Tensor& u, const Tensor& v, const Tensor& c,
                                   const Tensor& bias, const Tensor& gate_w, const Tensor& t1, const Tensor& t2,
                                   const Tensor& prob, const Tensor& d_out) {
  const rec_crossnet_mix_desc m = mix_desc(xl.shape(), u.shape());
  const int64_t B = m.batch;
  auto f32 = [&](const Shape& s) { return paddle::empty(s, DataType::FLOAT32, xl.place()); };
  auto dx0 = f32({B, m.d}), dxl = f32({B, m.d});
  auto gu = f32(u.shape()), gv = f32(v.shape()), gc = f32(c.shape()), gbias = f32(bias.shape());
  auto ggw = f32(gate_w.shape()), ggb = f32({m.experts});
  size_t fwd = 0, bwd = 0;
  REC_CALL(rec_crossnet_mix_layer_workspace_bytes(&m, &fwd, &bwd));
  auto ws = workspace(bwd, xl);
  REC_CALL(rec_crossnet_mix_layer_bwd(&m, x0.data<float>(), xl.data<float>(), u.data<float>(), v.data<float>(),
                                      c.data<float>(), bias.data<float>(), gate_w.data<float>(), t1.data<float>(),
                                      t2.data<float>(), prob.data<float>(), d_out.data<float>(), 0, dx0.data<float>(), 0,
                                      /*accumulate_dx0=*/0, /*fold_dx0=*/0, dxl.data<float>(), 0, gu.data<float>(),
                                      gv.data<float>(), gc.data<float>(), gbias.data<float>(), ggw.data<float>(),
                                      ggb.data<float>(), /*accumulate_gate=*/0, ws.data<uint8_t>(), bwd, xl.stream()));
  return {dx0, dxl, gu, gv, gc, gbias, ggw, ggb};
}

// =====================================================================================================================
// rec_din_attention_pool — replaces models/rank/din/net.py:141-173: the four history / target-sequence lookups, the
// [h, q, h-q, h*q] concat, the 80-40-1 attention MLP, mask, scale, softmax over T and weights @ h -> Out [B,E].
//   ids / Mask [B,T] i64 (Mask 0 valid, -1e9 padding: din/dinReader.py:81-84,99); tables [rows, dim];
//   AttW1 [4E,H1], AttB1 [H1], AttW2 [H1,H2], AttB2 [H2], AttW3 [H2,1], AttB3 [1]   (Paddle Linear: [in,out])
// =====================================================================================================================
rec_din_desc din_desc(const Shape& ids, const Shape& wi, const Shape& wc, const Shape& w1, const Shape& w2) {
  rec_din_desc d;
  d.batch = ids[0];
  d.max_len = static_cast<int32_t>(ids[1]);
  d.item_dim = static_cast<int32_t>(wi[1]);
  d.cat_dim = static_cast<int32_t>(wc[1]);
  d.hidden1 = static_cast<int32_t>(w1[1]);
  d.hidden2 = static_cast<int32_t>(w2[1]);
  d.item_rows = wi[0];
  d.cat_rows = wc[0];
  d.item_stride = d.item_dim;
  d.cat_stride = d.cat_dim;
  return d;
}
std::vector<Tensor> RecDinAttFwd(const Tensor& hist_item, const Tensor& hist_cat, const Tensor& tgt_item_seq,
                                 const Tensor& tgt_cat_seq, const Tensor& mask, const Tensor& w_hi, const Tensor& w_hc,
                                 const Tensor& w_ti, const Tensor& w_tc, const Tensor& a_w1, const Tensor& a_b1,
                                 const Tensor& a_w2, const Tensor& a_b2, const Tensor& a_w3, const Tensor& a_b3) {
  want(hist_item, DataType::INT64, 2, "HistItem");
  want(mask, DataType::INT64, 2, "Mask");
  PD_CHECK(hist_cat.shape() == hist_item.shape() && tgt_item_seq.shape() == hist_item.shape() &&
               tgt_cat_seq.shape() == hist_item.shape() && mask.shape() == hist_item.shape(), "id / mask shapes");
  PD_CHECK(w_ti.shape() == w_hi.shape() && w_tc.shape() == w_hc.shape(), "target tables must match the history tables");
  const rec_din_desc d = din_desc(hist_item.shape(), w_hi.shape(), w_hc.shape(), a_w1.shape(), a_w2.shape());
  const int64_t B = d.batch, T = d.max_len, E = d.item_dim + d.cat_dim;
  PD_CHECK(a_w1.shape()[0] == 4 * E && a_w2.shape()[0] == d.hidden1 && a_w3.numel() == d.hidden2, "attention MLP shapes");
  auto out = paddle::empty({B, E}, DataType::FLOAT32, mask.place());
  auto att = paddle::empty({B, T}, DataType::FLOAT32, mask.place());
  const bool saves = rec_din_saves_act1(&d) == 1;
  auto act1 = paddle::empty(saves ? Shape{B, T, d.hidden1} : Shape{1}, DataType::FLOAT32, mask.place());
  auto status = paddle::full({1}, 0, DataType::INT32, mask.place());
  size_t ws_bytes = 0;
  REC_CALL(rec_din_attention_pool_fwd_workspace_bytes(&d, &ws_bytes));
  auto ws = workspace(ws_bytes, mask);
  REC_CALL(rec_din_attention_pool_fwd_ws(
      &d, hist_item.data<int64_t>(), hist_cat.data<int64_t>(), tgt_item_seq.data<int64_t>(), tgt_cat_seq.data<int64_t>(),
      mask.data<int64_t>(), w_hi.data<float>(), w_hc.data<float>(), w_ti.data<float>(), w_tc.data<float>(),
      a_w1.data<float>(), a_b1.data<float>(), a_w2.data<float>(), a_b2.data<float>(), a_w3.data<float>(),
      a_b3.data<float>(), out.data<float>(), att.data<float>(), saves ? act1.data<float>() : nullptr,
      status.data<int32_t>(), ws_bytes ? ws.data<uint8_t>() : nullptr, ws_bytes, mask.stream()));
  return {out, att, act1, status};
}
std::vector<Shape> RecDinAttInferShape(const Shape& hist_item, const Shape& hist_cat, const Shape& tgt_item_seq,
                                       const Shape& tgt_cat_seq, const Shape& mask, const Shape& w_hi, const Shape& w_hc,
                                       const Shape& w_ti, const Shape& w_tc, const Shape& a_w1, const Shape& a_b1,
                                       const Shape& a_w2, const Shape& a_b2, const Shape& a_w3, const Shape& a_b3) {
  const rec_din_desc d = din_desc(hist_item, w_hi, w_hc, a_w1, a_w2);
  const int64_t B = d.batch, T = d.max_len;
  return {{B, d.item_dim + d.cat_dim}, {B, T}, rec_din_saves_act1(&d) == 1 ? Shape{B, T, d.hidden1} : Shape{1}, {1}};
}

// gradient w.r.t. the gathered rows: Grad(table) = SelectedRows value [B*T, dim] (rows = that table's id input).  The
// attention MLP gets no gradient: its Linears are not registered parameters in the reference's dygraph mode (duplicate
// add_sublayer names, din/net.py:84-102).
std::vector<Tensor> RecDinAttBwd(const Tensor& hist_item, const Tensor& hist_cat, const Tensor& tgt_item_seq,
                                 const Tensor& tgt_cat_seq, const Tensor& w_hi, const Tensor& w_hc, const Tensor& w_ti,
                                 const Tensor& w_tc, const Tensor& a_w1, const Tensor& a_b1, const Tensor& a_w2,
                                 const Tensor& a_b2, const Tensor& a_w3, const Tensor& out, const Tensor& att,
                                 const Tensor& act1, const Tensor& d_out) {
  const rec_din_desc d = din_desc(hist_item.shape(), w_hi.shape(), w_hc.shape(), a_w1.shape(), a_w2.shape());
  const int64_t B = d.batch, T = d.max_len, E = d.item_dim + d.cat_dim, n = B * T;
  auto f32 = [&](const Shape& s) { return paddle::empty(s, DataType::FLOAT32, att.place()); };
  auto d_hist = f32({B, T, E}), d_tgt = f32({B, T, E});
  auto w1t = f32({d.hidden1, 4 * E});
  void* st = att.stream();
  REC_CALL(rec_transpose_f32(4 * E, d.hidden1, a_w1.data<float>(), w1t.data<float>(), st));
  const bool saved = act1.numel() == n * d.hidden1 && rec_din_saves_act1(&d) == 1;
  size_t ws_bytes = 0;
  REC_CALL(rec_din_attention_pool_bwd_workspace_bytes(&d, &ws_bytes));
  auto ws = workspace(ws_bytes, att);
  REC_CALL(rec_din_attention_pool_bwd_ws(
      &d, hist_item.data<int64_t>(), hist_cat.data<int64_t>(), tgt_item_seq.data<int64_t>(), tgt_cat_seq.data<int64_t>(),
      w_hi.data<float>(), w_hc.data<float>(), w_ti.data<float>(), w_tc.data<float>(), a_w1.data<float>(),
      w1t.data<float>(), a_b1.data<float>(), a_w2.data<float>(), a_b2.data<float>(), a_w3.data<float>(), att.data<float>(),
      saved ? out.data<float>() : nullptr, saved ? act1.data<float>() : nullptr, d_out.data<float>(),
      d_hist.data<float>(), d_tgt.data<float>(), ws_bytes ? ws.data<uint8_t>() : nullptr, ws_bytes, st));
  // the engine writes [item | cat] columns side by side; the four parameters are separate tensors
  auto g_hi = f32({n, d.item_dim}), g_hc = f32({n, d.cat_dim}), g_ti = f32({n, d.item_dim}), g_tc = f32({n, d.cat_dim});
  const size_t f4 = sizeof(float), pitch = static_cast<size_t>(E) * f4;
  REC_CALL(rec_copy_2d_async(g_hi.data<float>(), d.item_dim * f4, d_hist.data<float>(), pitch, d.item_dim * f4, n, st));
  REC_CALL(rec_copy_2d_async(g_hc.data<float>(), d.cat_dim * f4, d_hist.data<float>() + d.item_dim, pitch, d.cat_dim * f4,
                             n, st));
  REC_CALL(rec_copy_2d_async(g_ti.data<float>(), d.item_dim * f4, d_tgt.data<float>(), pitch, d.item_dim * f4, n, st));
  REC_CALL(rec_copy_2d_async(g_tc.data<float>(), d.cat_dim * f4, d_tgt.data<float>() + d.item_dim, pitch, d.cat_dim * f4,
                             n, st));
  return {g_hi, g_hc, g_ti, g_tc};
}

}  // namespace

PD_BUILD_OP(rec_deepfm_fm)
    .Inputs({"Ids", "Dense", "W", "W1", "DenseW", "DenseWOne"})
    .Outputs({"Y1", "Y2", "FeatEmb", "SumEmb", "Status"})
    .Attrs({"padding_idx: int64_t"})
    .SetKernelFn(PD_KERNEL(RecDeepFmFwd))
    .SetInferShapeFn(PD_INFER_SHAPE(RecDeepFmInferShape))
    .SetInferDtypeFn(PD_INFER_DTYPE(RecDeepFmInferDtype));
PD_BUILD_GRAD_OP(rec_deepfm_fm)
    .Inputs({"Ids", "Dense", "FeatEmb", "SumEmb", "DenseW", paddle::Grad("FeatEmb"), paddle::Grad("Y1"), paddle::Grad("Y2")})
    .Outputs({paddle::Grad("W"), paddle::Grad("W1"), paddle::Grad("DenseW"), paddle::Grad("DenseWOne")})
    .SetKernelFn(PD_KERNEL(RecDeepFmBwd))
    .SetInferShapeFn(PD_INFER_SHAPE(RecDeepFmBwdInferShape))
    REC_SELECTED_ROWS("W", "Ids") REC_SELECTED_ROWS("W1", "Ids");

PD_BUILD_OP(rec_crossnet_v2_layer)
    .Inputs({"X0", "Xl", "W", "B"})
    .Outputs({"Out", "U"})
    .SetKernelFn(PD_KERNEL(RecCrossV2Fwd))
    .SetInferShapeFn(PD_INFER_SHAPE(RecCrossV2InferShape))
    .SetInferDtypeFn(PD_INFER_DTYPE(RecCrossV2InferDtype));
PD_BUILD_GRAD_OP(rec_crossnet_v2_layer)
    .Inputs({"X0", "Xl", "W", "U", paddle::Grad("Out")})
    .Outputs({paddle::Grad("X0"), paddle::Grad("Xl"), paddle::Grad("W"), paddle::Grad("B")})
    .SetKernelFn(PD_KERNEL(RecCrossV2Bwd));

PD_BUILD_OP(rec_crossnet_mix_layer)
    .Inputs({"X0", "Xl", "U", "V", "C", "Bias", "GateW", "GateB"})
    .Outputs({"Out", "T1", "T2", "Prob"})
    .SetKernelFn(PD_KERNEL(RecCrossMixFwd))
    .SetInferShapeFn(PD_INFER_SHAPE(RecCrossMixInferShape))
    .SetInferDtypeFn(PD_INFER_DTYPE(RecCrossMixInferDtype));
PD_BUILD_GRAD_OP(rec_crossnet_mix_layer)
    .Inputs({"X0", "Xl", "U", "V", "C", "Bias", "GateW", "T1", "T2", "Prob", paddle::Grad("Out")})
    .Outputs({paddle::Grad("X0"), paddle::Grad("Xl"), paddle::Grad("U"), paddle::Grad("V"), paddle::Grad("C"),
              paddle::Grad("Bias"), paddle::Grad("GateW"), paddle::Grad("GateB")})
    .SetKernelFn(PD_KERNEL(RecCrossMixBwd));

PD_BUILD_OP(rec_din_attention_pool)
    .Inputs({"HistItem", "HistCat", "TgtItemSeq", "TgtCatSeq", "Mask", "WHistItem", "WHistCat", "WTgtItemSeq", "WTgtCatSeq",
             "AttW1", "AttB1", "AttW2", "AttB2", "AttW3", "AttB3"})
    .Outputs({"Out", "AttWeight", "Act1", "Status"})
    .SetKernelFn(PD_KERNEL(RecDinAttFwd))
    .SetInferShapeFn(PD_INFER_SHAPE(RecDinAttInferShape));
PD_BUILD_GRAD_OP(rec_din_attention_pool)
    .Inputs({"HistItem", "HistCat", "TgtItemSeq", "TgtCatSeq", "WHistItem", "WHistCat", "WTgtItemSeq", "WTgtCatSeq", "AttW1",
             "AttB1", "AttW2", "AttB2", "AttW3", "Out", "AttWeight", "Act1", paddle::Grad("Out")})
    .Outputs({paddle::Grad("WHistItem"), paddle::Grad("WHistCat"), paddle::Grad("WTgtItemSeq"), paddle::Grad("WTgtCatSeq")})
    .SetKernelFn(PD_KERNEL(RecDinAttBwd))
    REC_SELECTED_ROWS("WHistItem", "HistItem") REC_SELECTED_ROWS("WHistCat", "HistCat")
    REC_SELECTED_ROWS("WTgtItemSeq", "TgtItemSeq") REC_SELECTED_ROWS("WTgtCatSeq", "TgtCatSeq");
