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
// input `T` is a PS record table the gradient operator updates in place: the loader lists it with the program's sparse
// tables (pass checkpoints, shrink, save) under the variable's name
#define REC_PS_TABLE(T) .Note("ps_table=" T)
#else
#define REC_SELECTED_ROWS(T, I)
#define REC_PS_TABLE(T)
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

// =====================================================================================================================
// rec_multislot_sumpool — replaces the loop of models/rank/slot_dnn/net.py:63-77 (and dnn/static_model_lod.py:70-97):
// for every slot `sparse_embedding(padding_idx=0) -> sequence_pool('sum')`, then concat(axis=1) — ALL slots of the batch
// in one launch (rec_multislot_sumpool_fwd).  The custom-op interface hands a kernel dense tensors without their LoD, so
// the multi-value slots arrive the way every custom sum-pool operator is fed, as two dense tensors:
//   Values  [nnz] (or [nnz,1]) i64 : the ids of slot 0's B segments, then slot 1's, ... (slot-major CSR)
//   Offsets [S, B+1] i64           : segment (s, b) = Values[Offsets[s,b] .. Offsets[s,b+1])   (absolute offsets)
//   W       [N, stride] f32        : the table; a row's embedding = its first emb_dim floats (stride == emb_dim for a
//                                    plain [dict_dim, emb_dim] parameter, wider for a record table)
//   key_mode 0: ids are rows of W;  1: uint64 feasigns hashed to rows on the device (row = 1 + mix64(f) % (N-1))
//   -> Out [B, S*emb_dim], Counts [B,S] i32 (ids pooled per segment, padding skipped), SegOfValue [nnz] i32 = b*S+s,
//      Rows [nnz] i64 (table row of every value; dropped ones -> the padding row), Status [1] i32.
// =====================================================================================================================
rec_multislot_desc ms_desc(int64_t B, int64_t S, int emb_dim, const Shape& w, int64_t padding_idx, int key_mode) {
  rec_multislot_desc d;
  d.batch = B;
  d.num_slots = static_cast<int32_t>(S);
  d.emb_dim = emb_dim;
  d.row_stride = static_cast<int32_t>(w[1]);
  d.key_mode = key_mode;
  d.num_rows = w[0];
  d.padding_idx = padding_idx;
  d.lod_stride = 0;
  d.out_stride = 0;
  d.state_offset = 0;
  d.init_dims = 0;
  d.init_range = 0.f;      // plain memory: a key that was never pushed reads as zeros (Paddle's zero_init default)
  d.init_seed = 0;
  return d;
}
std::vector<Tensor> RecMultislotFwd(const Tensor& values, const Tensor& offsets, const Tensor& w, int emb_dim,
                                    int64_t padding_idx, int key_mode) {
  PD_CHECK(values.dtype() == DataType::INT64 && (values.shape().size() == 1 ||
                                                 (values.shape().size() == 2 && values.shape()[1] == 1)),
           "Values must be [nnz] or [nnz,1] int64");
  want(offsets, DataType::INT64, 2, "Offsets");
  want(w, DataType::FLOAT32, 2, "W");
  const int64_t S = offsets.shape()[0], B = offsets.shape()[1] - 1, nnz = values.numel();
  PD_CHECK(S >= 1 && B >= 0 && emb_dim >= 1 && emb_dim <= w.shape()[1], "Offsets [S, B+1] / emb_dim <= row stride of W");
  const rec_multislot_desc d = ms_desc(B, S, emb_dim, w.shape(), padding_idx, key_mode);
  auto out = paddle::empty({B, S * emb_dim}, DataType::FLOAT32, w.place());
  auto counts = paddle::empty({B, S}, DataType::INT32, w.place());
  auto seg = paddle::empty({nnz}, DataType::INT32, w.place());
  auto rows = paddle::empty({nnz}, DataType::INT64, w.place());
  auto status = paddle::full({1}, 0, DataType::INT32, w.place());
  if (B == 0) return {out, counts, seg, rows, status};
  auto slot_base = paddle::full({S}, 0, DataType::INT64, w.place());          // Offsets are absolute
  auto none = paddle::full({1}, 0, DataType::INT64, w.place());               // a batch whose every segment is empty
  REC_CALL(rec_multislot_sumpool_fwd(&d, nnz ? values.data<int64_t>() : none.data<int64_t>(), offsets.data<int64_t>(),
                                     slot_base.data<int64_t>(), w.data<float>(), out.data<float>(),
                                     counts.data<int32_t>(), nnz ? seg.data<int32_t>() : nullptr,
                                     nnz ? rows.data<int64_t>() : nullptr, status.data<int32_t>(), w.stream()));
  return {out, counts, seg, rows, status};
}
std::vector<Shape> RecMultislotInferShape(const Shape& values, const Shape& offsets, const Shape& w, int emb_dim,
                                          int64_t padding_idx, int key_mode) {
  int64_t nnz = 1;
  for (auto v : values) nnz *= v;
  const int64_t S = offsets[0], B = offsets[1] - 1;
  return {{B, S * emb_dim}, {B, S}, {nnz}, {nnz}, {1}};
}
std::vector<DataType> RecMultislotInferDtype(DataType values, DataType offsets, DataType w) {
  return {w, DataType::INT32, DataType::INT32, DataType::INT64, DataType::INT32};
}
// gradient: the SelectedRows VALUE of the table, one row per pooled value (its segment's gradient row), rows = the
// forward's Rows output — the merge (MergeAdd) drops the padding row and sums duplicates, exactly as for nn.Embedding.
std::vector<Tensor> RecMultislotBwd(const Tensor& seg, const Tensor& d_out, int emb_dim) {
  want(d_out, DataType::FLOAT32, 2, "Grad(Out)");
  const int64_t B = d_out.shape()[0], nnz = seg.numel();
  PD_CHECK(emb_dim >= 1 && d_out.shape()[1] % emb_dim == 0, "Grad(Out) must be [B, S*emb_dim]");
  const int64_t S = d_out.shape()[1] / emb_dim;
  auto row_grad = paddle::empty({nnz, emb_dim}, DataType::FLOAT32, d_out.place());
  const rec_multislot_desc d = ms_desc(B, S, emb_dim, {1, emb_dim}, -1, 0);
  if (nnz)
    REC_CALL(rec_multislot_sumpool_bwd(&d, nnz, seg.data<int32_t>(), d_out.data<float>(), row_grad.data<float>(),
                                       d_out.stream()));
  return {row_grad};
}
std::vector<Shape> RecMultislotBwdInferShape(const Shape& seg, const Shape& d_out, int emb_dim) {
  return {{seg[0], emb_dim}};
}

// =====================================================================================================================
// rec_ps_pull / its gradient = the push — replaces the gpubox branch of models/rank/dnn/net.py:67-82 (and
// wide_deep/net.py:80): `static.nn.sparse_embedding(size=[N, D+2])` + `continuous_value_model(emb, show_click, False)`
// for ALL slots of the batch, i.e. the pull of the GPU parameter server (core.PSGPU, tools/static_gpubox_trainer.py:
// 152-160,256), on the engine's record table
//   Rec [rows, stride] f32, row = W(D) = [embed_w, embedx(D-1)] | show click g2sum_w g2sum_x state delta_score unseen_days
// (a persistable, non-trainable variable of the program — paddle.static.create_global_var).
//   Keys [B,S] i64 feasigns (= paddle.concat(sparse_inputs, 1)); ShowClick [B,2] f32 = [1, label] (dnn/static_model.py:
//   86-94); Anchor [1] f32: a trainable dummy so that the framework schedules the gradient operator at all — Paddle's
//   own pull_gpups_sparse takes its `W` input for the same reason.
//   -> Out [B,S,D] (the CVM columns are never attached: use_cvm=False strips them), Rows [B*S] i64, Status [1] i32.
// The gradient operator IS the table update: SelectedRows merge of the B*S gradient rows (rec_ids_group) +
// CtrCommonAccessor::Update + SparseAdaGradSGDRule on the touched records (rec_ps_push_rows), in place on Rec, with
// show = 1 per occurrence, click = the sample's label and the gradient of the SUMMED loss (grad_scale = B).
//   accessor = {lr, initial_g2sum, min_bound, max_bound, initial_range,  (embed_w)   5 floats
//               lr, initial_g2sum, min_bound, max_bound, initial_range,  (embedx)     5 floats
//               embedx_threshold, nonclk_coeff, click_coeff, seed}  — table_parameters.embedding.accessor of
//   slot_dnn/config_online.yaml:57-89; Paddle's defaults: {0.05, 3, -10, 10, 1e-4} x 2, 10, 0.1, 1.
// =====================================================================================================================
std::vector<Tensor> RecPsPull(const Tensor& keys, const Tensor& rec, const Tensor& show_click, const Tensor& anchor,
                              int emb_dim, std::vector<float> accessor) {
  want(keys, DataType::INT64, 2, "Keys");
  want(rec, DataType::FLOAT32, 2, "Rec");
  const int64_t B = keys.shape()[0], S = keys.shape()[1], n = B * S, rows_total = rec.shape()[0];
  PD_CHECK(emb_dim >= 2 && emb_dim + 7 <= rec.shape()[1], "Rec rows hold W(emb_dim) + 7 statistics");
  PD_CHECK(show_click.numel() == 2 * B && anchor.numel() == 1, "ShowClick [B,2], Anchor [1]");
  auto out = paddle::empty({B, S, emb_dim}, DataType::FLOAT32, rec.place());
  auto rows = paddle::empty({n}, DataType::INT64, rec.place());
  auto status = paddle::full({1}, 0, DataType::INT32, rec.place());
  if (n == 0) return {out, rows, status};
  void* st = rec.stream();
  REC_CALL(rec_feasign_rows(n, rows_total, keys.data<int64_t>(), rows.data<int64_t>(), st));   // 0 -> row 0 (padding)
  REC_CALL(rec_emb_gather(n, emb_dim, static_cast<int32_t>(rec.shape()[1]), rows_total, /*padding_idx=*/-1,
                          rows.data<int64_t>(), rec.data<float>(), out.data<float>(), 0, 0, status.data<int32_t>(), st));
  return {out, rows, status};
}
std::vector<Shape> RecPsPullInferShape(const Shape& keys, const Shape& rec, const Shape& show_click, const Shape& anchor,
                                       int emb_dim, std::vector<float> accessor) {
  return {{keys[0], keys[1], emb_dim}, {keys[0] * keys[1]}, {1}};
}
std::vector<DataType> RecPsPullInferDtype(DataType keys, DataType rec, DataType show_click, DataType anchor) {
  return {rec, DataType::INT64, DataType::INT32};
}
std::vector<Tensor> RecPsPush(const Tensor& rows, const Tensor& rec, const Tensor& show_click, const Tensor& d_out,
                              int emb_dim, std::vector<float> accessor) {
  want(d_out, DataType::FLOAT32, 3, "Grad(Out)");
  PD_CHECK(accessor.size() == 14, "accessor: 14 floats (see rec_paddle_ops.cc), got ", accessor.size());
  const int64_t B = d_out.shape()[0], S = d_out.shape()[1], n = B * S, rows_total = rec.shape()[0];
  PD_CHECK(rows.numel() == n && d_out.shape()[2] == emb_dim, "Rows / Grad(Out) shapes");
  auto d_anchor = paddle::full({1}, 0, DataType::FLOAT32, rec.place());
  if (n == 0) return {d_anchor};
  void* st = rec.stream();
  size_t ws_bytes = 0;
  REC_CALL(rec_ids_group_workspace_bytes(n, rows_total, &ws_bytes));
  auto ws = workspace(ws_bytes, rec);
  auto spos = paddle::empty({n}, DataType::INT32, rec.place());
  auto uniq = paddle::empty({n}, DataType::INT64, rec.place());
  auto seg = paddle::empty({n + 1}, DataType::INT32, rec.place());
  auto nu = paddle::full({4}, 0, DataType::INT32, rec.place());
  auto status = paddle::full({1}, 0, DataType::INT32, rec.place());
  auto click = paddle::empty({B}, DataType::INT64, rec.place());
  REC_CALL(rec_cast_f32_i64(B, show_click.data<float>() + 1, 2, click.data<int64_t>(), st));
  // feasign 0 — and only feasign 0 — maps to row 0: dropping row 0 drops exactly the padding key
  REC_CALL(rec_ids_group(n, static_cast<int32_t>(S), rows_total, /*padding row*/ 0, rows.data<int64_t>(), nullptr,
                         spos.data<int32_t>(), uniq.data<int64_t>(), seg.data<int32_t>(), nu.data<int32_t>(),
                         status.data<int32_t>(), ws.data<uint8_t>(), ws_bytes, st));
  const rec_ps_layout lay{static_cast<int32_t>(rec.shape()[1]), /*embed_off*/ 0, /*embedx_off*/ 1, emb_dim - 1,
                          /*stat_off*/ emb_dim};
  const float* a = accessor.data();
  rec_ps_accessor acc;
  acc.lr = a[0]; acc.initial_g2sum = a[1]; acc.min_bound = a[2]; acc.max_bound = a[3]; acc.initial_range = a[4];
  acc.x_lr = a[5]; acc.x_initial_g2sum = a[6]; acc.x_min_bound = a[7]; acc.x_max_bound = a[8]; acc.x_initial_range = a[9];
  acc.embedx_threshold = a[10]; acc.nonclk_coeff = a[11]; acc.click_coeff = a[12];
  acc.grad_scale = static_cast<float>(B);      // Paddle pushes the gradient of the SUMMED loss
  acc.show_scale = 1;
  acc.embed_zero_init = 1;
  acc.seed = static_cast<uint64_t>(a[13]);
  acc.row_mul = 1;
  acc.row_add = 0;
  const rec_grad_src gx{d_out.data<float>(), {1, 0, 0, nullptr, nullptr, 0}, emb_dim, 1};   // embedx = columns 1..D-1
  const rec_grad_src gw{d_out.data<float>(), {1, 0, 0, nullptr, nullptr, 0}, emb_dim, 0};   // embed_w = column 0
  // Rec is updated IN PLACE: the PS table has no dense gradient, the "gradient" of the pull is the accessor's push
  REC_CALL(rec_ps_push_rows(n, static_cast<int32_t>(S), &lay, nu.data<int32_t>(), uniq.data<int64_t>(),
                            seg.data<int32_t>(), spos.data<int32_t>(), &gx, &gw, /*show: 1 per occurrence*/ nullptr,
                            click.data<int64_t>(), const_cast<float*>(rec.data<float>()), &acc, st));
  return {d_anchor};
}
std::vector<Shape> RecPsPushInferShape(const Shape& rows, const Shape& rec, const Shape& show_click, const Shape& d_out,
                                       int emb_dim, std::vector<float> accessor) {
  return {{1}};
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

PD_BUILD_OP(rec_multislot_sumpool)
    .Inputs({"Values", "Offsets", "W"})
    .Outputs({"Out", "Counts", "SegOfValue", "Rows", "Status"})
    .Attrs({"emb_dim: int", "padding_idx: int64_t", "key_mode: int"})
    .SetKernelFn(PD_KERNEL(RecMultislotFwd))
    .SetInferShapeFn(PD_INFER_SHAPE(RecMultislotInferShape))
    .SetInferDtypeFn(PD_INFER_DTYPE(RecMultislotInferDtype));
PD_BUILD_GRAD_OP(rec_multislot_sumpool)
    .Inputs({"SegOfValue", paddle::Grad("Out")})
    .Outputs({paddle::Grad("W")})
    .Attrs({"emb_dim: int"})
    .SetKernelFn(PD_KERNEL(RecMultislotBwd))
    .SetInferShapeFn(PD_INFER_SHAPE(RecMultislotBwdInferShape))
    REC_SELECTED_ROWS("W", "Rows");

PD_BUILD_OP(rec_ps_pull)
    .Inputs({"Keys", "Rec", "ShowClick", "Anchor"})
    .Outputs({"Out", "Rows", "Status"})
    .Attrs({"emb_dim: int", "accessor: std::vector<float>"})
    .SetKernelFn(PD_KERNEL(RecPsPull))
    .SetInferShapeFn(PD_INFER_SHAPE(RecPsPullInferShape))
    .SetInferDtypeFn(PD_INFER_DTYPE(RecPsPullInferDtype))
    REC_PS_TABLE("Rec");
PD_BUILD_GRAD_OP(rec_ps_pull)
    .Inputs({"Rows", "Rec", "ShowClick", paddle::Grad("Out")})
    .Outputs({paddle::Grad("Anchor")})
    .Attrs({"emb_dim: int", "accessor: std::vector<float>"})
    .SetKernelFn(PD_KERNEL(RecPsPush))
    .SetInferShapeFn(PD_INFER_SHAPE(RecPsPushInferShape));
