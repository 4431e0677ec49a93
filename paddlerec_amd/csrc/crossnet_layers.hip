// CrossNetV2 / CrossNetMix layers as single C-ABI entry points (`crossnet_v2_layer`, `crossnet_mix_layer`).
//
// Reference: /root/reference/models/rank/dcn_v2/net.py
//   :214-226  CrossNetV2.forward   x_{l+1} = x_l + x_0 * (W_l x_l + b_l)
//   :278-320  CrossNetMix.forward  x_{l+1} = x_l + sum_e p_e(x_l) * x_0 * (U_e tanh(C_e tanh(V_e^T x_l)) + b_l),
//             p = softmax over the experts of Linear(d, 1)_e(x_l)
// and what loss.backward() computes for them.  These functions own NO kernel of their own: they issue the GEMMs
// (rec_gemm_f32 with the CROSS / MOE / BIAS_TANH / DTANH / ADD epilogues) and the streaming glue passes
// (rec_cross_bwd_prep, rec_moe_bwd_prep, rec_softmax_rows{,_bwd}, rec_colsum) of a layer in the order the host mirror
// used to issue them from Python, so a Paddle custom op binds ONE symbol per layer and direction.  All buffers are
// the caller's; the workspace holds the GEMM split-K partials plus the layer's scratch tensors.
#include "rec_common.h"

namespace rec {

__global__ __launch_bounds__(kBlock) void add_into_kernel(int64_t n, float* __restrict__ dst,
                                                          const float* __restrict__ src, int accumulate) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
    dst[i] = accumulate ? dst[i] + src[i] : src[i];
}

static void add_into(int64_t n, float* dst, const float* src, int accumulate, hipStream_t st) {
  int64_t grid = (n + kBlock - 1) / kBlock;
  if (grid > kNumCU * 8) grid = kNumCU * 8;
  hipLaunchKernelGGL(add_into_kernel, dim3((unsigned)grid), dim3(kBlock), 0, st, n, dst, src, accumulate);
}

struct Carve {   // bump allocator over the caller's workspace (256-B aligned pieces)
  char* base;
  size_t off = 0;
  explicit Carve(void* p) : base((char*)p) {}
  float* take(size_t floats) {
    float* r = base ? (float*)(base + off) : nullptr;
    off += align_up(floats * sizeof(float), 256);
    return r;
  }
};

static rec_gemm_desc gd(int64_t m, int n, int k, int lda, int ldb, int ldc, int ta, int tb, int epi) {
  rec_gemm_desc d;
  d.m = m; d.n = n; d.k = k; d.lda = lda; d.ldb = ldb; d.ldc = ldc;
  d.trans_a = ta; d.trans_b = tb; d.epilogue = epi; d.split_k = 0;
  return d;
}

static size_t gemm_ws(const rec_gemm_desc& d) {
  size_t b = 0;
  rec_gemm_f32_workspace_bytes(&d, &b);
  return align_up(b, 256);
}

static size_t max2(size_t a, size_t b) { return a > b ? a : b; }

}  // namespace rec

using namespace rec;

#define REC_TRY(call)        \
  do {                       \
    if (int rc_ = (call)) return rc_; \
  } while (0)

// ------------------------------------------------------------------------------------------ CrossNetV2
static int v2_check(const rec_crossnet_v2_desc* d) {
  REC_REQUIRE(d, REC_EINVAL, "desc is NULL");
  REC_REQUIRE(d->batch >= 0 && d->d > 0, REC_EINVAL, "bad sizes B=%lld d=%d", (long long)d->batch, d->d);
  return REC_OK;
}
static int ldx(int ld, int d) { return ld > 0 ? ld : d; }
static int v2_ldmax(const rec_crossnet_v2_desc* d) {
  int m = d->d;
  for (int v : {d->ld_x0, d->ld_xl, d->ld_out, d->ld_u}) m = v > m ? v : m;
  return m;
}
static int mix_ldmax(const rec_crossnet_mix_desc* d) {
  int m = d->d;
  for (int v : {d->ld_x0, d->ld_xl, d->ld_out}) m = v > m ? v : m;
  return m;
}

extern "C" int rec_crossnet_v2_layer_workspace_bytes(const rec_crossnet_v2_desc* d, size_t* fwd_bytes,
                                                     size_t* bwd_bytes) {
  REC_TRY(v2_check(d));
  const int n = d->d;
  // split-K partials are [splits][M][ldc]: sized for the widest row stride the descriptor names (gradient buffers
  // handed to the backward must not be strided wider than that)
  const int lw = v2_ldmax(d);
  if (fwd_bytes) *fwd_bytes = gemm_ws(gd(d->batch, n, n, lw, n, lw, 0, 0, REC_EPI_CROSS));
  if (bwd_bytes) {
    const size_t g = max2(gemm_ws(gd(n, n, (int)d->batch, lw, n, n, 1, 0, REC_EPI_NONE)),
                          gemm_ws(gd(d->batch, n, n, n, n, lw, 0, 1, REC_EPI_ADD)));
    *bwd_bytes = align_up((size_t)d->batch * n * sizeof(float), 256) + g;
  }
  return REC_OK;
}

extern "C" int rec_crossnet_v2_layer_fwd(const rec_crossnet_v2_desc* d, const float* X0, const float* Xl,
                                         const float* W, const float* bias, float* Xnext, float* U_saved,
                                         void* workspace, size_t workspace_bytes, void* stream) {
  REC_TRY(v2_check(d));
  if (d->batch == 0) return REC_OK;
  REC_REQUIRE(X0 && Xl && W && bias && Xnext, REC_EINVAL, "null pointer argument");
  const int n = d->d;
  const rec_gemm_desc g = gd(d->batch, n, n, ldx(d->ld_xl, n), n, ldx(d->ld_out, n), 0, 0, REC_EPI_CROSS);
  rec_gemm_epilogue_args a = {};
  a.bias = bias;
  a.aux0 = X0; a.ld_aux0 = ldx(d->ld_x0, n);
  a.aux1 = Xl; a.ld_aux1 = ldx(d->ld_xl, n);
  a.out2 = U_saved; a.ld_out2 = ldx(d->ld_u, n);
  return rec_gemm_f32(&g, Xl, W, Xnext, &a, workspace, workspace_bytes, stream);
}

// d X_{l+1} -> d X_l, dW_l, db_l; dX0_acc (+)= d X_{l+1} * U_l.  fold_dx0: also add dX0_acc into d X_l (the first
// layer, whose X_l IS X_0).  dXl may alias dXnext.
extern "C" int rec_crossnet_v2_layer_bwd(const rec_crossnet_v2_desc* d, const float* X0, const float* Xl,
                                         const float* W, const float* U_saved, const float* dXnext,
                                         int32_t ld_dxnext, float* dX0_acc, int32_t ld_acc,
                                         int32_t accumulate_dx0, int32_t fold_dx0, float* dXl, int32_t ld_dxl,
                                         float* dW, float* db, void* workspace, size_t workspace_bytes,
                                         void* stream) {
  REC_TRY(v2_check(d));
  if (d->batch == 0) return REC_OK;
  REC_REQUIRE(X0 && Xl && W && U_saved && dXnext && dX0_acc && dXl && dW && db, REC_EINVAL,
              "null pointer argument");
  const int n = d->d;
  size_t need = 0;
  rec_crossnet_v2_layer_workspace_bytes(d, nullptr, &need);
  REC_REQUIRE(workspace && workspace_bytes >= need, REC_EWORKSPACE, "workspace %zu < %zu", workspace_bytes, need);
  Carve c(workspace);
  float* dU = c.take((size_t)d->batch * n);
  void* gws = (char*)workspace + c.off;
  const size_t gws_bytes = workspace_bytes - c.off;
  REC_TRY(rec_cross_bwd_prep(d->batch, n, dXnext, ldx(ld_dxnext, n), X0, ldx(d->ld_x0, n), U_saved,
                             ldx(d->ld_u, n), dU, n, dX0_acc, ldx(ld_acc, n), accumulate_dx0, stream));
  {   // dW = X_l^T dU, db = colsum(dU)
    const rec_gemm_desc g = gd(n, n, (int)d->batch, ldx(d->ld_xl, n), n, n, 1, 0, REC_EPI_NONE);
    rec_gemm_epilogue_args a = {};
    a.b_colsum = db;
    REC_TRY(rec_gemm_f32(&g, Xl, dU, dW, &a, gws, gws_bytes, stream));
  }
  {   // d X_l = d X_{l+1} + dU W^T (+ dX0_acc)
    const rec_gemm_desc g = gd(d->batch, n, n, n, n, ldx(ld_dxl, n), 0, 1, REC_EPI_ADD);
    rec_gemm_epilogue_args a = {};
    a.aux1 = dXnext; a.ld_aux1 = ldx(ld_dxnext, n);
    if (fold_dx0) { a.aux0 = dX0_acc; a.ld_aux0 = ldx(ld_acc, n); }
    REC_TRY(rec_gemm_f32(&g, dU, W, dXl, &a, gws, gws_bytes, stream));
  }
  return REC_OK;
}

// ------------------------------------------------------------------------------------------ CrossNetMix
static int mix_check(const rec_crossnet_mix_desc* d) {
  REC_REQUIRE(d, REC_EINVAL, "desc is NULL");
  REC_REQUIRE(d->batch >= 0 && d->d > 0 && d->rank > 0 && d->experts > 0 && d->experts <= 64, REC_EINVAL,
              "bad sizes B=%lld d=%d r=%d E=%d", (long long)d->batch, d->d, d->rank, d->experts);
  return REC_OK;
}

static size_t mix_gemm_ws(const rec_crossnet_mix_desc* d) {
  const int n = d->d, r = d->rank, E = d->experts;
  const int64_t B = d->batch;
  const int lw = mix_ldmax(d);     // widest row stride named (split-K partials are [splits][M][ldc])
  size_t g = 0;
  g = max2(g, gemm_ws(gd(B, E, n, lw, E, E, 0, 0, REC_EPI_BIAS)));          // gate
  g = max2(g, gemm_ws(gd(B, r, n, lw, r, E * r, 0, 0, REC_EPI_BIAS_TANH)));  // x V_e
  g = max2(g, gemm_ws(gd(B, r, r, E * r, r, E * r, 0, 1, REC_EPI_BIAS_TANH)));
  g = max2(g, gemm_ws(gd(B, n, r, E * r, r, lw, 0, 1, REC_EPI_MOE)));
  g = max2(g, gemm_ws(gd(n, r, (int)B, lw, E * r, r, 1, 0, REC_EPI_NONE)));  // dU_e, dV_e
  g = max2(g, gemm_ws(gd(r, r, (int)B, r, E * r, r, 1, 0, REC_EPI_NONE)));  // dC_e
  g = max2(g, gemm_ws(gd(B, r, n, n, r, r, 0, 0, REC_EPI_DTANH)));
  g = max2(g, gemm_ws(gd(B, r, r, r, r, r, 0, 0, REC_EPI_DTANH)));
  g = max2(g, gemm_ws(gd(B, n, r, r, r, lw, 0, 1, REC_EPI_ADD)));
  g = max2(g, gemm_ws(gd(n, E, (int)B, lw, E, E, 1, 0, REC_EPI_NONE)));      // d gate_w
  g = max2(g, gemm_ws(gd(B, n, E, E, E, lw, 0, 1, REC_EPI_ADD)));
  size_t cs = 0;
  rec_colsum_workspace_bytes(B, n, &cs);
  return max2(g, align_up(cs, 256));
}

extern "C" int rec_crossnet_mix_layer_workspace_bytes(const rec_crossnet_mix_desc* d, size_t* fwd_bytes,
                                                      size_t* bwd_bytes) {
  REC_TRY(mix_check(d));
  const size_t B = (size_t)d->batch, n = d->d, r = d->rank, E = d->experts;
  const size_t g = mix_gemm_ws(d);
  auto al = [](size_t floats) { return align_up(floats * sizeof(float), 256); };
  if (fwd_bytes) *fwd_bytes = al(B * E) + g;                                      // gate logits
  if (bwd_bytes)   // u, du [B,d] | dp [B,E] | dc, da [B,r] | dgate [B,E] | dbias_e [d] | gw [d,E] | gb [E]
    *bwd_bytes = 2 * al(B * n) + al(B * E) + 2 * al(B * r) + al(B * E) + al(n) + al(n * E) + al(E) + g;
  return REC_OK;
}

// Parameters in the reference's layout: U, V [E, d, r], C [E, r, r], bias [d], gate_w [d, E] (the E Linear(d,1)
// stacked), gate_b [E].  Saved for the backward: t1 = tanh(x_l V_e), t2 = tanh(t1 C_e^T) as [B, E*r], prob [B, E].
extern "C" int rec_crossnet_mix_layer_fwd(const rec_crossnet_mix_desc* d, const float* X0, const float* Xl,
                                          const float* U, const float* V, const float* Cm, const float* bias,
                                          const float* gate_w, const float* gate_b, float* Xnext, float* t1,
                                          float* t2, float* prob, void* workspace, size_t workspace_bytes,
                                          void* stream) {
  REC_TRY(mix_check(d));
  if (d->batch == 0) return REC_OK;
  REC_REQUIRE(X0 && Xl && U && V && Cm && bias && gate_w && gate_b && Xnext && t1 && t2 && prob, REC_EINVAL,
              "null pointer argument");
  const int n = d->d, r = d->rank, E = d->experts;
  const int64_t B = d->batch;
  const int lx0 = ldx(d->ld_x0, n), lxl = ldx(d->ld_xl, n), lo = ldx(d->ld_out, n);
  size_t need = 0;
  rec_crossnet_mix_layer_workspace_bytes(d, &need, nullptr);
  REC_REQUIRE(workspace && workspace_bytes >= need, REC_EWORKSPACE, "workspace %zu < %zu", workspace_bytes, need);
  Carve c(workspace);
  float* gate = c.take((size_t)B * E);
  void* gws = (char*)workspace + c.off;
  const size_t gwb = workspace_bytes - c.off;
  {
    const rec_gemm_desc g = gd(B, E, n, lxl, E, E, 0, 0, REC_EPI_BIAS);
    rec_gemm_epilogue_args a = {};
    a.bias = gate_b;
    REC_TRY(rec_gemm_f32(&g, Xl, gate_w, gate, &a, gws, gwb, stream));
  }
  REC_TRY(rec_softmax_rows(B, E, gate, E, prob, E, stream));                                  // net.py:315
  for (int e = 0; e < E; ++e) {
    {   // t1_e = tanh(x_l V_e)                                                                  net.py:292-296
      const rec_gemm_desc g = gd(B, r, n, lxl, r, E * r, 0, 0, REC_EPI_BIAS_TANH);
      rec_gemm_epilogue_args a = {};
      REC_TRY(rec_gemm_f32(&g, Xl, V + (size_t)e * n * r, t1 + e * r, &a, gws, gwb, stream));
    }
    {   // t2_e = tanh(t1_e C_e^T)                                                               net.py:297-298
      const rec_gemm_desc g = gd(B, r, r, E * r, r, E * r, 0, 1, REC_EPI_BIAS_TANH);
      rec_gemm_epilogue_args a = {};
      REC_TRY(rec_gemm_f32(&g, t1 + e * r, Cm + (size_t)e * r * r, t2 + e * r, &a, gws, gwb, stream));
    }
  }
  for (int e = 0; e < E; ++e) {   // x_{l+1} = x_l + sum_e p_e x_0 (t2_e U_e^T + b)                net.py:301-317
    const rec_gemm_desc g = gd(B, n, r, E * r, r, lo, 0, 1, REC_EPI_MOE);
    rec_gemm_epilogue_args a = {};
    a.bias = bias;
    a.aux0 = X0; a.ld_aux0 = lx0;
    a.aux1 = e == 0 ? Xl : Xnext; a.ld_aux1 = e == 0 ? lxl : lo;
    a.row_scale = prob + e; a.row_scale_stride = E;
    REC_TRY(rec_gemm_f32(&g, t2 + e * r, U + (size_t)e * n * r, Xnext, &a, gws, gwb, stream));
  }
  return REC_OK;
}

// Backward of one CrossNetMix layer.  gU / gV / gC / gbias are written; the gating Linear layers are shared by all
// cross layers (net.py:267-268), so g_gate_w / g_gate_b are overwritten when accumulate_gate == 0 and added to
// otherwise.  dX0_acc / fold_dx0 / aliasing as for rec_crossnet_v2_layer_bwd.
extern "C" int rec_crossnet_mix_layer_bwd(const rec_crossnet_mix_desc* d, const float* X0, const float* Xl,
                                          const float* U, const float* V, const float* Cm, const float* bias,
                                          const float* gate_w, const float* t1, const float* t2,
                                          const float* prob, const float* dXnext, int32_t ld_dxnext,
                                          float* dX0_acc, int32_t ld_acc, int32_t accumulate_dx0,
                                          int32_t fold_dx0, float* dXl, int32_t ld_dxl, float* gU, float* gV,
                                          float* gC, float* gbias, float* g_gate_w, float* g_gate_b,
                                          int32_t accumulate_gate, void* workspace, size_t workspace_bytes,
                                          void* stream) {
  REC_TRY(mix_check(d));
  if (d->batch == 0) return REC_OK;
  REC_REQUIRE(X0 && Xl && U && V && Cm && bias && gate_w && t1 && t2 && prob && dXnext && dX0_acc && dXl && gU &&
                  gV && gC && gbias && g_gate_w && g_gate_b, REC_EINVAL, "null pointer argument");
  REC_REQUIRE(dXl != dXnext, REC_EINVAL, "dXl must not alias dXnext (every expert reads dXnext)");
  const int n = d->d, r = d->rank, E = d->experts;
  const int64_t B = d->batch;
  const int lx0 = ldx(d->ld_x0, n), lxl = ldx(d->ld_xl, n), ldn = ldx(ld_dxnext, n), lacc = ldx(ld_acc, n),
            ldl = ldx(ld_dxl, n);
  hipStream_t st = (hipStream_t)stream;
  size_t need = 0;
  rec_crossnet_mix_layer_workspace_bytes(d, nullptr, &need);
  REC_REQUIRE(workspace && workspace_bytes >= need, REC_EWORKSPACE, "workspace %zu < %zu", workspace_bytes, need);
  Carve c(workspace);
  float* u = c.take((size_t)B * n);
  float* du = c.take((size_t)B * n);
  float* dp = c.take((size_t)B * E);
  float* dc = c.take((size_t)B * r);
  float* da = c.take((size_t)B * r);
  float* dgate = c.take((size_t)B * E);
  float* dbias_e = c.take(n);
  float* gw = c.take((size_t)n * E);
  float* gb = c.take(E);
  void* gws = (char*)workspace + c.off;
  const size_t gwb = workspace_bytes - c.off;
  int acc = accumulate_dx0;
  for (int e = 0; e < E; ++e) {
    const float* t1e = t1 + e * r;
    const float* t2e = t2 + e * r;
    const float* Ue = U + (size_t)e * n * r;
    const float* Ve = V + (size_t)e * n * r;
    const float* Ce = Cm + (size_t)e * r * r;
    rec_gemm_epilogue_args a;
    {   // recompute u_e = t2_e U_e^T + b
      const rec_gemm_desc g = gd(B, n, r, E * r, r, n, 0, 1, REC_EPI_BIAS);
      a = {}; a.bias = bias;
      REC_TRY(rec_gemm_f32(&g, t2e, Ue, u, &a, gws, gwb, stream));
    }
    REC_TRY(rec_moe_bwd_prep(B, n, dXnext, ldn, X0, lx0, u, n, prob + e, E, du, n, dX0_acc, lacc, acc, dp + e, E,
                             stream));
    acc = 1;
    REC_TRY(rec_colsum(B, n, n, du, dbias_e, gws, gwb, stream));
    add_into(n, gbias, dbias_e, e > 0, st);
    {   // dU_e = du^T t2_e
      const rec_gemm_desc g = gd(n, r, (int)B, n, E * r, r, 1, 0, REC_EPI_NONE);
      a = {};
      REC_TRY(rec_gemm_f32(&g, du, t2e, gU + (size_t)e * n * r, &a, gws, gwb, stream));
    }
    {   // dc = (du U_e) * (1 - t2_e^2)
      const rec_gemm_desc g = gd(B, r, n, n, r, r, 0, 0, REC_EPI_DTANH);
      a = {}; a.aux0 = t2e; a.ld_aux0 = E * r;
      REC_TRY(rec_gemm_f32(&g, du, Ue, dc, &a, gws, gwb, stream));
    }
    {   // dC_e = dc^T t1_e
      const rec_gemm_desc g = gd(r, r, (int)B, r, E * r, r, 1, 0, REC_EPI_NONE);
      a = {};
      REC_TRY(rec_gemm_f32(&g, dc, t1e, gC + (size_t)e * r * r, &a, gws, gwb, stream));
    }
    {   // da = (dc C_e) * (1 - t1_e^2)
      const rec_gemm_desc g = gd(B, r, r, r, r, r, 0, 0, REC_EPI_DTANH);
      a = {}; a.aux0 = t1e; a.ld_aux0 = E * r;
      REC_TRY(rec_gemm_f32(&g, dc, Ce, da, &a, gws, gwb, stream));
    }
    {   // dV_e = x_l^T da
      const rec_gemm_desc g = gd(n, r, (int)B, lxl, r, r, 1, 0, REC_EPI_NONE);
      a = {};
      REC_TRY(rec_gemm_f32(&g, Xl, da, gV + (size_t)e * n * r, &a, gws, gwb, stream));
    }
    {   // d x_l (+)= da V_e^T   (starts from d x_{l+1}: the residual path)
      const rec_gemm_desc g = gd(B, n, r, r, r, ldl, 0, 1, REC_EPI_ADD);
      a = {};
      a.aux1 = e == 0 ? dXnext : dXl; a.ld_aux1 = e == 0 ? ldn : ldl;
      REC_TRY(rec_gemm_f32(&g, da, Ve, dXl, &a, gws, gwb, stream));
    }
  }
  REC_TRY(rec_softmax_rows_bwd(B, E, prob, E, dp, E, dgate, E, stream));
  {   // gating: d gate_w (+)= x_l^T dgate, d gate_b (+)= colsum(dgate), d x_l += dgate gate_w^T (+ dX0_acc)
    rec_gemm_epilogue_args a = {};
    const rec_gemm_desc g1 = gd(n, E, (int)B, lxl, E, E, 1, 0, REC_EPI_NONE);
    REC_TRY(rec_gemm_f32(&g1, Xl, dgate, gw, &a, gws, gwb, stream));
    add_into((int64_t)n * E, g_gate_w, gw, accumulate_gate, st);
    REC_TRY(rec_colsum(B, E, E, dgate, gb, gws, gwb, stream));
    add_into(E, g_gate_b, gb, accumulate_gate, st);
    const rec_gemm_desc g2 = gd(B, n, E, E, E, ldl, 0, 1, REC_EPI_ADD);
    a = {};
    a.aux1 = dXl; a.ld_aux1 = ldl;
    if (fold_dx0) { a.aux0 = dX0_acc; a.ld_aux0 = lacc; }
    REC_TRY(rec_gemm_f32(&g2, dgate, gate_w, dXl, &a, gws, gwb, stream));
  }
  return check_launch("rec_crossnet_mix_layer_bwd");
}
