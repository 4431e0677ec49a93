// PS / gpubox table accessor on the device (gfx950): push (Update), Shrink, Save selection of CTR feature values.
//
// Reference configuration (the arithmetic lives in the un-vendored PaddlePaddle PS code [EXT]):
//   /root/reference/models/rank/slot_dnn/config_online.yaml:57-89
//       accessor_class SparseAccessor, fea_dim 11 = show, click, embed_w + embedx(8), embedx_threshold 10,
//       embed_sgd_param / embedx_sgd_param = SparseAdaGradSGDRule(lr 0.05, initial_g2sum 3, initial_range 1e-4,
//       weight_bounds +-10), ctr_accessor_param(nonclk_coeff 0.1, click_coeff 1.0, base_threshold 1.5, delta_threshold
//       0.25, delta_keep_days 16, show_click_decay_rate 0.98, delete_threshold 0.8, delete_after_unseen_days 30)
//   /root/reference/models/rank/slot_dnn/net.py:61-62  ShowClickEntry(show, click): the per-sample show / click
//       values a push adds to the feature's counters
//   /root/reference/tools/static_gpubox_trainer.py:152-160,256  the table lives in the GPU parameter server
// Semantics = the published source of PaddlePaddle release/2.4 (tag v2.4.2), restated in oracle/ps_ref.py with the
// C++ text quoted there: paddle/fluid/distributed/ps/table/{sparse_sgd_rule.cc (SparseAdaGradSGDRule::UpdateValueWork,
// InitValueWork), ctr_accessor.cc (Create, Update, NeedExtendMF, Shrink, Save, UpdateStatAfterSave),
// memory_sparse_table.cc (PullSparse / PushSparse: when the embedx part is created)} and the gpubox twin
// paddle/fluid/framework/fleet/heter_ps/optimizer.cuh.h.  Push of a key with merged gradient g, pushed show / click:
//     new key:  value created WITHOUT embedx: embed_w = 0 (zero_init, Paddle's default; uniform(+-range) if switched
//               off), counters and g2sums 0
//     show += push_show; click += push_click; delta_score += (push_show - push_click)*nonclk + push_click*clk;
//     unseen_days = 0
//     rule(w, g2sum, g, scale = push_show):  double scaled = g * grad_scale / scale;
//               w = float(w - lr * scaled * sqrtf(g0 / (g0 + g2sum))), clipped;  g2sum = float(g2sum + sum(scaled^2)/n)
//         on embed_w (n = 1) and, if the value HAS its embedx part, on embedx (n = embedx_dim)
//     a value without embedx drops its embedx gradient and is extended (embedx = uniform(+-range), embedx_g2sum = 0)
//     at the end of the push that makes (show - click)*nonclk + click*clk >= embedx_threshold
// A key that was never pushed is zero memory: it reads as embed_w = 0, embedx = 0 — exactly what PullSparse returns
// for a missing key — so a 160 GB shard needs no init pass.  Creation values are a pure function of
// (seed, row, element) (ps_init_value) instead of Paddle's unseeded thread-local engine.
#include <stdlib.h>

#include "rec_common.h"

namespace rec {

struct GradSrc {
  const float* grad;
  rec_grad_layout gl;
  int pitch, col;
};

// element offset of the gradient row of lookup position `pos` (see rec_grad_layout)
__device__ __forceinline__ int64_t ps_grad_offset(const rec_grad_layout& gl, int pos, int D) {
  const int p = gl.index ? gl.index[pos] : pos;
  const int q = gl.div > 1 ? p / gl.div : p;
  return gl.group > 0 ? (int64_t)(q / gl.group) * gl.group_stride + (int64_t)(q % gl.group) * D
                      : (int64_t)q * D;
}

template <int VEC>
__device__ __forceinline__ void ps_segment_sum(float (&g)[VEC], int beg, int end,
                                               const int32_t* __restrict__ spos, const GradSrc& s, int c) {
  int k = beg;
  for (; k + 4 <= end; k += 4) {   // four independent gathers in flight, summed in ascending order
    float a[VEC], b[VEC], cc[VEC], d[VEC];
    vload<VEC>(a, s.grad + ps_grad_offset(s.gl, spos[k], s.pitch) + c);
    vload<VEC>(b, s.grad + ps_grad_offset(s.gl, spos[k + 1], s.pitch) + c);
    vload<VEC>(cc, s.grad + ps_grad_offset(s.gl, spos[k + 2], s.pitch) + c);
    vload<VEC>(d, s.grad + ps_grad_offset(s.gl, spos[k + 3], s.pitch) + c);
#pragma unroll
    for (int i = 0; i < VEC; ++i) g[i] = (((g[i] + a[i]) + b[i]) + cc[i]) + d[i];
  }
  for (; k < end; ++k) {
    float a[VEC];
    vload<VEC>(a, s.grad + ps_grad_offset(s.gl, spos[k], s.pitch) + c);
#pragma unroll
    for (int i = 0; i < VEC; ++i) g[i] += a[i];
  }
}

// SparseAdaGradSGDRule::UpdateValueWork on one element: typed evaluation of the C++ text — `double scaled_grad =
// grad[i] / scale;` is a FLOAT division (const float* grad, float scale) of the float pushed gradient g * grad_scale,
// widened afterwards; ratio in float; the update and g2sum in double, float on store.  Returns scaled_grad^2.
struct PsRule {
  float lr, g0, lo, hi;
};
struct PsScale {
  float grad_scale, scale;   // the batch size the trainer multiplies the gradient by; the pushed show (or 1)
};
__device__ __forceinline__ double ps_rule_elem(float& w, float g, PsScale inv, float ratio, const PsRule& R) {
#pragma clang fp contract(off)
  const float pushed = g * inv.grad_scale;
  const double sg = (double)(pushed / inv.scale);
  float nw = (float)((double)w - (double)R.lr * sg * (double)ratio);
  w = fminf(fmaxf(nw, R.lo), R.hi);
  return sg * sg;
}

template <int LANES>
__device__ __forceinline__ double group_sum_f64(double x) {
#pragma unroll
  for (int o = LANES / 2; o > 0; o >>= 1) x += __shfl_xor(x, o, kWave);
  return x;
}

constexpr int kStShow = 0, kStClick = 1, kStG2w = 2, kStG2x = 3, kStState = 4, kStDelta = 5, kStUnseen = 6;

// One row group of LANES lanes per touched feature: the lanes own VEC-float slices of the embedx part, lane 0 of
// the group also owns embed_w and the statistics (same record line).
// STATV: embed_w and the seven statistics are eight consecutive floats on a 16-byte boundary (the DeepFM layout: embed_w
// at D, statistics at D + 1, D a multiple of 4) — read as two float4s by every lane of the group and written as two
// float4s by lane 0 instead of 6 scalar loads per lane + 9 scalar stores (see the narrow kernel's WHOLE mode).
template <int VEC, int LANES, bool STATV = false>
__global__ __launch_bounds__(kBlock) void ps_push_rows_kernel(
    rec_ps_layout L, int S, const int32_t* __restrict__ n_uniq, const int64_t* __restrict__ uniq,
    const int32_t* __restrict__ seg_off, const int32_t* __restrict__ spos, GradSrc gx, GradSrc gw,
    const int64_t* __restrict__ show, const int64_t* __restrict__ click, float* __restrict__ rec,
    rec_ps_accessor A) {
  const int64_t u = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / LANES;
  const int lane = threadIdx.x % kWave;
  const int lg = threadIdx.x % LANES;
  const int d0 = lg * VEC;
  const bool rowok = u < n_uniq[0];
  const int64_t row = rowok ? uniq[u] : 0;
  const int beg = rowok ? seg_off[u] : 0, end = rowok ? seg_off[u + 1] : 0;
  float* r = rec + row * (int64_t)L.row_stride;
  float* st = r + L.stat_off;            // show, click, g2sum_w, g2sum_x, state, delta_score, unseen_days
  const int64_t grow = row * A.row_mul + A.row_add;   // identity of the feature across shards (creation values)
  const int Dx = L.embedx_dim;
  const bool xlane = rowok && d0 < Dx;

  // ---- pushed show / click of the key (lane 0 of the group), shared with the group
  float dshow = 0.f, dclick = 0.f;
  if (rowok && lg == 0) {
    auto smp = [&](int pos) { return (gx.gl.index ? gx.gl.index[pos] : pos) / S; };
    if (show) {
      for (int k = beg; k < end; ++k) dshow += (float)show[smp(spos[k])];
    } else {
      dshow = (float)(end - beg);   // show = 1 per occurrence (dnn/static_model.py:86-94; the reader's "0":[1] slot)
    }
    if (click) {
      int k = beg;
      for (; k + 4 <= end; k += 4) {
        const int64_t l0 = click[smp(spos[k])], l1 = click[smp(spos[k + 1])], l2 = click[smp(spos[k + 2])],
                      l3 = click[smp(spos[k + 3])];
        dclick += (float)(l0 + l1 + l2 + l3);
      }
      for (; k < end; ++k) dclick += (float)click[smp(spos[k])];
    }
  }
  const int head = lane - lg;
  dshow = __shfl(dshow, head, kWave);
  dclick = __shfl(dclick, head, kWave);

  float show0 = 0.f, click0 = 0.f, g2w = 0.f, g2x = 0.f, state = 0.f, delta0 = 0.f, ew0 = 0.f;
  if (rowok) {
    if constexpr (STATV) {
      const float4 q0 = *reinterpret_cast<const float4*>(r + L.embed_off);
      const float4 q1 = *reinterpret_cast<const float4*>(r + L.embed_off + 4);
      ew0 = q0.x; show0 = q0.y; click0 = q0.z; g2w = q0.w; g2x = q1.x; state = q1.y; delta0 = q1.z;
    } else {
      show0 = st[kStShow]; click0 = st[kStClick]; g2w = st[kStG2w]; g2x = st[kStG2x]; state = st[kStState];
      delta0 = st[kStDelta];
    }
  }
  const bool unborn = state == 0.f;
  const bool has_x = state >= 2.f;
  const float show1 = show0 + dshow, click1 = click0 + dclick;
  const float score1 = (show1 - click1) * A.nonclk_coeff + click1 * A.click_coeff;
  const PsScale inv = {A.grad_scale, (A.show_scale && dshow > 0.f) ? dshow : 1.f};

  // ---- embedx part
  float w[VEC], g[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { w[i] = 0.f; g[i] = 0.f; }
  double sq = 0.0;
  const PsRule RX = {A.x_lr, A.x_initial_g2sum, A.x_min_bound, A.x_max_bound};
  if (xlane && has_x) {
    vload<VEC>(w, r + L.embedx_off + d0);
    ps_segment_sum<VEC>(g, beg, end, spos, gx, gx.col + d0);
    const float ratio = sqrtf(RX.g0 / (RX.g0 + g2x));
#pragma unroll
    for (int i = 0; i < VEC; ++i)
      if (d0 + i < Dx) sq += ps_rule_elem(w[i], g[i], inv, ratio, RX);
  }
  sq = group_sum_f64<LANES>(sq);   // (0 everywhere for a value without embedx)
  const bool create_x = !has_x && Dx > 0 && score1 >= A.embedx_threshold;   // NeedExtendMF on the updated counters
  if (xlane) {
    if (create_x) {
#pragma unroll
      for (int i = 0; i < VEC; ++i)
        w[i] = d0 + i < Dx ? fminf(fmaxf(ps_init_value(A.seed, grow, 1 + d0 + i, A.x_initial_range), RX.lo), RX.hi)
                           : 0.f;
    }
    if (has_x || create_x) {
      if (VEC == 1 || d0 + VEC <= Dx) {
        vstore<VEC>(r + L.embedx_off + d0, w);
      } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i)
          if (d0 + i < Dx) r[L.embedx_off + d0 + i] = w[i];
      }
    }
  }
  // ---- embed_w + statistics
  if (rowok && lg == 0) {
    float ew = unborn ? (A.embed_zero_init ? 0.f : ps_init_value(A.seed, grow, 0, A.initial_range))
                      : (STATV ? ew0 : r[L.embed_off]);
    float gwv[1] = {0.f};
    ps_segment_sum<1>(gwv, beg, end, spos, gw, gw.col);
    const PsRule RW = {A.lr, A.initial_g2sum, A.min_bound, A.max_bound};
    const double sqw = ps_rule_elem(ew, gwv[0], inv, sqrtf(RW.g0 / (RW.g0 + g2w)), RW);
    const float g2w1 = (float)((double)g2w + sqw);
    const float g2x1 = has_x ? (float)((double)g2x + sq / (double)Dx) : (create_x ? 0.f : g2x);
    const float state1 = (has_x || create_x) ? 2.f : 1.f;
    const float delta1 = delta0 + ((dshow - dclick) * A.nonclk_coeff + dclick * A.click_coeff);
    if constexpr (STATV) {
      *reinterpret_cast<float4*>(r + L.embed_off) = make_float4(ew, show1, click1, g2w1);
      *reinterpret_cast<float4*>(r + L.embed_off + 4) = make_float4(g2x1, state1, delta1, 0.f);
    } else {
      r[L.embed_off] = ew;
      st[kStShow] = show1;
      st[kStClick] = click1;
      st[kStG2w] = g2w1;
      if (has_x || create_x) st[kStG2x] = g2x1;
      st[kStState] = state1;
      st[kStDelta] = delta1;
      st[kStUnseen] = 0.f;
    }
  }
}

// Narrow features without float4 row groups (slot_dnn: embedx_dim 8 at record offset 1, gradient rows of 9 floats):
// the kernel above then runs 8 lanes per feature = 8 features per wave, each walking uniq -> seg_off -> sorted_pos
// -> grad_index -> gradient row -> record one dependent round trip after the other (4.2 ms for 9 M features).
// Here ONE LANE owns a feature: 64 features per wave in flight, the index chain coalesced across lanes, the 64-B
// record the lane's private line.  Same rule, same ascending-position summation.
// WHOLE: the slot layout (row_stride 16 = one 64-byte record: embed_w at 0, embedx(8) at 1, the seven statistics at 9) —
// the lane reads and writes its record as FOUR float4s instead of 17 + 17 scalar accesses to the same line (with 64
// records per wave instruction and dozens of waves per CU the line does not stay in L1 between them: every scalar access
// was an L2 round trip — 1.80 ms alone for 3.95 GB, half the chip's gather rate).
template <int DX, bool WHOLE = false>   // DX: embedx floats held in registers (>= embedx_dim)
__global__ __launch_bounds__(kBlock) void ps_push_rows_narrow_kernel(
    rec_ps_layout L, int S, const int32_t* __restrict__ n_uniq, const int64_t* __restrict__ uniq,
    const int32_t* __restrict__ seg_off, const int32_t* __restrict__ spos, GradSrc gx, GradSrc gw,
    const int64_t* __restrict__ show, const int64_t* __restrict__ click, float* __restrict__ rec,
    rec_ps_accessor A, bool row9) {
  // n_max is a capacity (the number of touched features is only known on the device): capped grid, strided loop
  const int nu = n_uniq[0];
  const int Dx = L.embedx_dim;
  const PsRule RW = {A.lr, A.initial_g2sum, A.min_bound, A.max_bound};
  const PsRule RX = {A.x_lr, A.x_initial_g2sum, A.x_min_bound, A.x_max_bound};
  for (int64_t u = (int64_t)blockIdx.x * kBlock + threadIdx.x; u < nu; u += (int64_t)gridDim.x * kBlock) {
  const int64_t row = uniq[u];
  const int beg = seg_off[u], end = seg_off[u + 1];
  float* r = rec + row * (int64_t)L.row_stride;
  float* st = r + L.stat_off;            // show, click, g2sum_w, g2sum_x, state, delta_score, unseen_days
  const int64_t grow = row * A.row_mul + A.row_add;
  float R[16];                           // WHOLE: the record in registers
  if constexpr (WHOLE) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 t = reinterpret_cast<const float4*>(r)[q];
      R[4 * q] = t.x; R[4 * q + 1] = t.y; R[4 * q + 2] = t.z; R[4 * q + 3] = t.w;
    }
  }
  const float show0 = WHOLE ? R[9 + kStShow] : st[kStShow], click0 = WHOLE ? R[9 + kStClick] : st[kStClick],
              g2w = WHOLE ? R[9 + kStG2w] : st[kStG2w], g2x = WHOLE ? R[9 + kStG2x] : st[kStG2x],
              state = WHOLE ? R[9 + kStState] : st[kStState], delta0 = WHOLE ? R[9 + kStDelta] : st[kStDelta];
  const bool unborn = state == 0.f;
  const bool has_x = state >= 2.f;
  float w[DX];
#pragma unroll
  for (int d = 0; d < DX; ++d) w[d] = (has_x && d < Dx) ? (WHOLE ? R[1 + (d < 8 ? d : 0)] : r[L.embedx_off + d]) : 0.f;
  float ew = unborn ? (A.embed_zero_init ? 0.f : ps_init_value(A.seed, grow, 0, A.initial_range))
                    : (WHOLE ? R[0] : r[L.embed_off]);

  // ---- one walk over the feature's occurrences: counters, embed_w gradient, embedx gradient
  float dshow = show ? 0.f : (float)(end - beg), dclick = 0.f, gwv = 0.f;
  float g[DX];
#pragma unroll
  for (int d = 0; d < DX; ++d) g[d] = 0.f;
  int64_t lsum = 0;
  for (int k = beg; k < end; ++k) {
    const int pos = spos[k];
    if (show || click) {
      const int smp = (gx.gl.index ? gx.gl.index[pos] : pos) / S;
      if (show) dshow += (float)show[smp];
      if (click) lsum += click[smp];
    }
    if (WHOLE && row9) {
      // [g_embed_w | g_embedx(8)] are nine consecutive floats of one gradient row (4-byte aligned: 36-byte rows): two
      // dwordx4 loads + one dword instead of nine dword loads (gfx950 global loads take any dword alignment)
      typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
      const float* a = gx.grad + ps_grad_offset(gx.gl, pos, gx.pitch);
      const f4u t0 = *reinterpret_cast<const f4u*>(a), t1 = *reinterpret_cast<const f4u*>(a + 4);
      const float t2 = a[8];
      gwv += t0.x;
      if (has_x) {
        const float v[8] = {t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w, t2};
#pragma unroll
        for (int d = 0; d < 8; ++d)
          if (d < DX) g[d] += v[d];
      }
    } else {
      gwv += gw.grad[ps_grad_offset(gw.gl, pos, gw.pitch) + gw.col];
      if (has_x) {
        const float* a = gx.grad + ps_grad_offset(gx.gl, pos, gx.pitch) + gx.col;
#pragma unroll
        for (int d = 0; d < DX; ++d) g[d] += d < Dx ? a[d] : 0.f;
      }
    }
  }
  // (the wide kernel adds the labels four at a time as integers, then to float: any grouping of integer partial
  // sums below 2^24 gives the same float)
  dclick = (float)lsum;
  const float show1 = show0 + dshow, click1 = click0 + dclick;
  const float score1 = (show1 - click1) * A.nonclk_coeff + click1 * A.click_coeff;
  const PsScale inv = {A.grad_scale, (A.show_scale && dshow > 0.f) ? dshow : 1.f};

  double sq = 0.0;
  if (has_x) {
    const float ratio = sqrtf(RX.g0 / (RX.g0 + g2x));
#pragma unroll
    for (int d = 0; d < DX; ++d)
      if (d < Dx) sq += ps_rule_elem(w[d], g[d], inv, ratio, RX);
  }
  const bool create_x = !has_x && Dx > 0 && score1 >= A.embedx_threshold;   // NeedExtendMF on the updated counters
  if (create_x) {
#pragma unroll
    for (int d = 0; d < DX; ++d)
      w[d] = d < Dx ? fminf(fmaxf(ps_init_value(A.seed, grow, 1 + d, A.x_initial_range), RX.lo), RX.hi) : 0.f;
  }
  const double sqw = ps_rule_elem(ew, gwv, inv, sqrtf(RW.g0 / (RW.g0 + g2w)), RW);
  if constexpr (WHOLE) {
    if (has_x || create_x) {
#pragma unroll
      for (int d = 0; d < 8; ++d)
        if (d < DX) R[1 + d] = w[d];
    }
    R[0] = ew;
    R[9 + kStShow] = show1;
    R[9 + kStClick] = click1;
    R[9 + kStG2w] = (float)((double)g2w + sqw);
    if (has_x) R[9 + kStG2x] = (float)((double)g2x + sq / (double)Dx);
    else if (create_x) R[9 + kStG2x] = 0.f;
    R[9 + kStState] = (has_x || create_x) ? 2.f : 1.f;
    R[9 + kStDelta] = delta0 + ((dshow - dclick) * A.nonclk_coeff + dclick * A.click_coeff);
    R[9 + kStUnseen] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      reinterpret_cast<float4*>(r)[q] = make_float4(R[4 * q], R[4 * q + 1], R[4 * q + 2], R[4 * q + 3]);
  } else {
    if (has_x || create_x) {
#pragma unroll
      for (int d = 0; d < DX; ++d)
        if (d < Dx) r[L.embedx_off + d] = w[d];
    }
    r[L.embed_off] = ew;
    st[kStShow] = show1;
    st[kStClick] = click1;
    st[kStG2w] = (float)((double)g2w + sqw);
    if (has_x) st[kStG2x] = (float)((double)g2x + sq / (double)Dx);
    else if (create_x) st[kStG2x] = 0.f;
    st[kStState] = (has_x || create_x) ? 2.f : 1.f;
    st[kStDelta] = delta0 + ((dshow - dclick) * A.nonclk_coeff + dclick * A.click_coeff);
    st[kStUnseen] = 0.f;
  }
  }
}

__global__ __launch_bounds__(kBlock) void ps_shrink_rows_kernel(int64_t N, rec_ps_layout L,
                                                                 float* __restrict__ rec, float decay,
                                                                 float delete_threshold, float delete_after_unseen_days,
                                                                 float nonclk, float clk,
                                                                 int64_t* __restrict__ n_deleted) {
  const int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (row >= N) return;
  float* r = rec + row * (int64_t)L.row_stride;
  float* st = r + L.stat_off;
  if (st[kStState] == 0.f) return;                // no such key: zero memory stays untouched
  const float show = st[kStShow] * decay, click = st[kStClick] * decay;
  const float score = (show - click) * nonclk + click * clk;
  if (score < delete_threshold || st[kStUnseen] > delete_after_unseen_days) {   // delete: the key is gone
    r[L.embed_off] = 0.f;
    for (int d = 0; d < L.embedx_dim; ++d) r[L.embedx_off + d] = 0.f;
#pragma unroll
    for (int i = 0; i < 7; ++i) st[i] = 0.f;
    if (n_deleted) atomicAdd((unsigned long long*)n_deleted, 1ull);
  } else {
    st[kStShow] = show;
    st[kStClick] = click;
  }
}

// CtrCommonAccessor::Save(value, param) + UpdateStatAfterSave(value, param) for every row of the table:
// selected[row] = 1 if a save of kind `param` writes the value (0 all, 1 delta, 2 base, 3 all + a day passes).
__global__ __launch_bounds__(kBlock) void ps_save_select_kernel(int64_t N, rec_ps_layout L, float* __restrict__ rec,
                                                                 int param, float base_threshold,
                                                                 float delta_threshold, float delta_keep_days,
                                                                 float nonclk, float clk,
                                                                 uint8_t* __restrict__ selected,
                                                                 int64_t* __restrict__ n_selected) {
  const int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (row >= N) return;
  float* st = rec + row * (int64_t)L.row_stride + L.stat_off;
  bool sel = false;
  if (st[kStState] != 0.f) {
    if (param == 1 || param == 2) {
      const float show = st[kStShow], click = st[kStClick];
      const float score = (show - click) * nonclk + click * clk;
      const float dth = param == 2 ? 0.f : delta_threshold;
      sel = score >= base_threshold && st[kStDelta] >= dth && st[kStUnseen] <= delta_keep_days;
      if (sel) st[kStDelta] = 0.f;
    } else {
      sel = true;
      if (param == 3) st[kStUnseen] += 1.f;
    }
  }
  if (selected) selected[row] = sel ? 1 : 0;
  if (sel && n_selected) atomicAdd((unsigned long long*)n_selected, 1ull);
}

static int check_layout(const rec_ps_layout* L) {
  REC_REQUIRE(L, REC_EINVAL, "layout is NULL");
  REC_REQUIRE(L->row_stride > 0 && L->embedx_dim >= 0 && L->embed_off >= 0 && L->embedx_off >= 0 &&
                  L->stat_off >= 0,
              REC_EINVAL, "bad record layout");
  REC_REQUIRE(L->embed_off < L->row_stride && L->embedx_off + L->embedx_dim <= L->row_stride &&
                  L->stat_off + 7 <= L->row_stride,
              REC_EINVAL, "record parts do not fit in row_stride %d", L->row_stride);
  // the seven statistics must not overlap the weights
  const bool ov_x = L->stat_off < L->embedx_off + L->embedx_dim && L->embedx_off < L->stat_off + 7;
  const bool ov_w = L->embed_off >= L->stat_off && L->embed_off < L->stat_off + 7;
  REC_REQUIRE(!(ov_x && L->embedx_dim > 0) && !ov_w, REC_EINVAL, "record parts overlap");
  return REC_OK;
}

}  // namespace rec

using namespace rec;

extern "C" int rec_ps_push_rows(int64_t n_max, int32_t num_slots, const rec_ps_layout* layout,
                                const int32_t* n_uniq, const int64_t* uniq_rows,
                                const int32_t* seg_offset, const int32_t* sorted_pos,
                                const rec_grad_src* grad_embedx, const rec_grad_src* grad_embed,
                                const int64_t* show, const int64_t* click, float* rec,
                                const rec_ps_accessor* accessor, void* stream) {
  if (int rc = check_layout(layout)) return rc;
  REC_REQUIRE(n_max >= 0 && num_slots > 0 && accessor && grad_embed && grad_embed->grad, REC_EINVAL,
              "bad arguments");
  REC_REQUIRE(layout->embedx_dim == 0 || (grad_embedx && grad_embedx->grad), REC_EINVAL,
              "embedx gradient missing");
  REC_REQUIRE(n_uniq && uniq_rows && seg_offset && sorted_pos && rec, REC_EINVAL, "null pointer argument");
  REC_REQUIRE(accessor->initial_g2sum > 0.f && accessor->min_bound <= accessor->max_bound &&
                  accessor->initial_range >= 0.f && accessor->x_initial_g2sum > 0.f &&
                  accessor->x_min_bound <= accessor->x_max_bound && accessor->x_initial_range >= 0.f &&
                  accessor->grad_scale > 0.f && accessor->row_mul >= 1 && accessor->row_add >= 0,
              REC_EINVAL, "bad accessor parameters");
  if (n_max == 0) return REC_OK;
  GradSrc gw = {grad_embed->grad, grad_embed->layout, grad_embed->pitch, grad_embed->col};
  GradSrc gx = gw;
  if (layout->embedx_dim > 0) gx = {grad_embedx->grad, grad_embedx->layout, grad_embedx->pitch, grad_embedx->col};
  REC_REQUIRE(gw.pitch >= 1 && gx.pitch >= 1 && gw.col >= 0 && gx.col >= 0 && gw.gl.div >= 1 && gx.gl.div >= 1,
              REC_EINVAL, "bad gradient source");
  gw.gl.partials = gx.gl.partials = nullptr;
  const int Dx = layout->embedx_dim > 0 ? layout->embedx_dim : 1;
  const bool vec = layout->embedx_dim > 0 && layout->embedx_off % 4 == 0 && layout->row_stride % 4 == 0 &&
                   gx.pitch % 4 == 0 && gx.col % 4 == 0 && ((uintptr_t)gx.grad) % 16 == 0 &&
                   ((uintptr_t)rec) % 16 == 0 && (gx.gl.group <= 0 || gx.gl.group_stride % 4 == 0);
  const int lanes = pow2_ceil(vec ? (Dx + 3) / 4 : Dx);
  REC_REQUIRE(lanes <= 64, REC_ESHAPE, "embedx_dim %d too large", Dx);
  hipStream_t st = (hipStream_t)stream;
  static const bool narrow_ok = [] { const char* v = getenv("REC_NARROW_ROWS"); return !(v && *v == '0'); }();
  if (!vec && Dx <= 16 && narrow_ok) {   // no float4 row groups: one lane per feature
    int64_t grid = (n_max + kBlock - 1) / kBlock;
    if (grid > (int64_t)kNumCU * 32) grid = (int64_t)kNumCU * 32;   // grid-stride loop in the kernel
#define REC_PS_NARROW(DX_)                                                                               \
  hipLaunchKernelGGL((ps_push_rows_narrow_kernel<DX_>), dim3((unsigned)grid), dim3(kBlock), 0, st, *layout, \
                     num_slots, n_uniq, uniq_rows, seg_offset, sorted_pos, gx, gw, show, click, rec, *accessor, false)
    static const bool whole_env = [] { const char* v = getenv("REC_PS_WHOLE_RECORD"); return !(v && *v == '0'); }();
    const bool whole = whole_env && layout->row_stride == 16 && layout->embed_off == 0 && layout->embedx_off == 1 &&
                       layout->embedx_dim == 8 && layout->stat_off == 9 && ((uintptr_t)rec) % 16 == 0;
    // both gradients from ONE 9-float row (slot_dnn: d(pooled)[sample, slot] = [g_embed_w | g_embedx(8)])
    const bool row9 = gx.grad == gw.grad && gx.pitch == 9 && gw.pitch == 9 && gx.col == 1 && gw.col == 0 &&
                      gx.gl.div == gw.gl.div && gx.gl.group == gw.gl.group && gx.gl.group_stride == gw.gl.group_stride &&
                      gx.gl.index == gw.gl.index;
    if (whole)
      hipLaunchKernelGGL((ps_push_rows_narrow_kernel<8, true>), dim3((unsigned)grid), dim3(kBlock), 0, st, *layout,
                         num_slots, n_uniq, uniq_rows, seg_offset, sorted_pos, gx, gw, show, click, rec, *accessor, row9);
    else if (Dx <= 4) REC_PS_NARROW(4); else if (Dx <= 8) REC_PS_NARROW(8);
    else if (Dx <= 12) REC_PS_NARROW(12); else REC_PS_NARROW(16);
#undef REC_PS_NARROW
    return check_launch("rec_ps_push_rows (narrow)");
  }
  static const bool statv_env = [] { const char* v = getenv("REC_PS_WHOLE_RECORD"); return !(v && *v == '0'); }();
  const bool statv = statv_env && layout->embed_off % 4 == 0 && layout->stat_off == layout->embed_off + 1 &&
                     layout->row_stride % 4 == 0 && ((uintptr_t)rec) % 16 == 0;
#define REC_PS_CASE(V, L_)                                                                             \
  if (lanes == L_) {                                                                                   \
    const int64_t grid = (n_max * L_ + kBlock - 1) / kBlock;                                           \
    REC_REQUIRE(grid < (1ll << 31), REC_ESHAPE, "too many rows");                                      \
    if (statv)                                                                                         \
      hipLaunchKernelGGL((ps_push_rows_kernel<V, L_, true>), dim3((unsigned)grid), dim3(kBlock), 0, st, \
                         *layout, num_slots, n_uniq, uniq_rows, seg_offset, sorted_pos, gx, gw, show,  \
                         click, rec, *accessor);                                                       \
    else                                                                                               \
      hipLaunchKernelGGL((ps_push_rows_kernel<V, L_>), dim3((unsigned)grid), dim3(kBlock), 0, st,      \
                         *layout, num_slots, n_uniq, uniq_rows, seg_offset, sorted_pos, gx, gw, show,  \
                         click, rec, *accessor);                                                       \
    return check_launch("rec_ps_push_rows");                                                           \
  }
  if (vec) {
    REC_PS_CASE(4, 1) REC_PS_CASE(4, 2) REC_PS_CASE(4, 4) REC_PS_CASE(4, 8) REC_PS_CASE(4, 16)
    REC_PS_CASE(4, 32) REC_PS_CASE(4, 64)
  } else {
    REC_PS_CASE(1, 1) REC_PS_CASE(1, 2) REC_PS_CASE(1, 4) REC_PS_CASE(1, 8) REC_PS_CASE(1, 16)
    REC_PS_CASE(1, 32) REC_PS_CASE(1, 64)
  }
#undef REC_PS_CASE
  set_error("embedx_dim %d unsupported", Dx);
  return REC_ESHAPE;
}

extern "C" float rec_ps_init_value_host(uint64_t seed, int64_t row, int32_t element, float initial_range) {
  return ps_init_value(seed, row, element, initial_range);
}

extern "C" int rec_ps_shrink_rows(int64_t num_rows, const rec_ps_layout* layout, float* rec,
                                  float show_click_decay_rate, float delete_threshold,
                                  float delete_after_unseen_days, const rec_ps_accessor* accessor,
                                  int64_t* n_deleted, void* stream) {
  if (int rc = check_layout(layout)) return rc;
  REC_REQUIRE(num_rows >= 0 && rec && accessor && show_click_decay_rate >= 0.f, REC_EINVAL, "bad arguments");
  if (num_rows == 0) return REC_OK;
  const int64_t grid = (num_rows + kBlock - 1) / kBlock;
  REC_REQUIRE(grid < (1ll << 31), REC_ESHAPE, "too many rows");
  hipLaunchKernelGGL(ps_shrink_rows_kernel, dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream,
                     num_rows, *layout, rec, show_click_decay_rate, delete_threshold, delete_after_unseen_days,
                     accessor->nonclk_coeff, accessor->click_coeff, n_deleted);
  return check_launch("rec_ps_shrink_rows");
}

extern "C" int rec_ps_save_select(int64_t num_rows, const rec_ps_layout* layout, float* rec, int32_t param,
                                  float base_threshold, float delta_threshold, float delta_keep_days,
                                  const rec_ps_accessor* accessor, uint8_t* selected, int64_t* n_selected,
                                  void* stream) {
  if (int rc = check_layout(layout)) return rc;
  REC_REQUIRE(num_rows >= 0 && rec && accessor && param >= 0 && param <= 3, REC_EINVAL, "bad arguments");
  if (num_rows == 0) return REC_OK;
  const int64_t grid = (num_rows + kBlock - 1) / kBlock;
  REC_REQUIRE(grid < (1ll << 31), REC_ESHAPE, "too many rows");
  hipLaunchKernelGGL(ps_save_select_kernel, dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream, num_rows,
                     *layout, rec, param, base_threshold, delta_threshold, delta_keep_days, accessor->nonclk_coeff,
                     accessor->click_coeff, selected, n_selected);
  return check_launch("rec_ps_save_select");
}
