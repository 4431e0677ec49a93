// PS / gpubox table accessor on the device (gfx950): push (update) and shrink of CTR feature values.
//
// Reference (the rule is configured in the repo, its arithmetic lives in the un-vendored Paddle PS code [EXT]):
//   /root/reference/models/rank/slot_dnn/config_online.yaml:57-89
//       accessor_class SparseAccessor, fea_dim 11 = show, click, embed_w + embedx(8), embedx_threshold 10,
//       embed_sgd_param / embedx_sgd_param = SparseAdaGradSGDRule(lr 0.05, initial_g2sum 3, initial_range 1e-4,
//       weight_bounds +-10), ctr_accessor_param(nonclk_coeff 0.1, click_coeff 1.0, show_click_decay_rate 0.98,
//       delete_threshold 0.8, ...)
//   /root/reference/models/rank/slot_dnn/net.py:61-62  ShowClickEntry(show, click): the per-sample show / click
//       values a push adds to the feature's counters
//   /root/reference/tools/static_gpubox_trainer.py:152-160,256  the table lives in the GPU parameter server
// Semantics restated (CtrCommonAccessor::Update / NeedExtendMF / Shrink, SparseAdaGradSGDRule::UpdateValueWork):
//   push of a feature with merged gradient g and batch counters (dshow, dclick):
//     show += dshow; click += dclick
//     embed_w  -= lr * g_w * sqrt(g0 / (g0 + g2sum_w)), clipped to the bounds;  g2sum_w += g_w^2
//     if the feature HAS its embedx part:
//       embedx_w -= lr * g_x * sqrt(g0 / (g0 + g2sum_x)), clipped;  g2sum_x += sum(g_x^2) / embedx_dim
//   the embedx part is created (uniform(+-initial_range)) by the first pull that finds
//     (show - click) * nonclk_coeff + click * click_coeff >= embedx_threshold;   before that it reads as zeros and
//     its gradient is dropped.  Here the creation is done at the END of the push that crosses the threshold — the
//     same table state the next pull would produce.
//   a feature itself is created at its first pull (embed_w = uniform(+-initial_range), counters 0): rows are born
//   lazily from zeroed memory, with values that are a pure function of (seed, row, element) — see ps_init_value.
//   shrink (end of a pass / day): show *= decay, click *= decay; features whose score fell below delete_threshold
//   are deleted (row zeroed: unborn again).  unseen_days / delta_score bookkeeping (SSD tiering, delta saves) is
//   storage-engine state outside this path.
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

// One row group of LANES lanes per touched feature: the lanes own VEC-float slices of the embedx part, lane 0 of
// the group also owns embed_w and the counters (same record line).
template <int VEC, int LANES>
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
  float* st = r + L.stat_off;            // show, click, g2sum_w, g2sum_x, state
  const int64_t grow = row * A.row_mul + A.row_add;   // identity of the feature across shards (creation values)
  const int Dx = L.embedx_dim;
  const bool xlane = rowok && d0 < Dx;

  // ---- counters of the batch (lane 0 of the group), shared with the group
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

  float show0 = 0.f, click0 = 0.f, g2w = 0.f, g2x = 0.f, state = 0.f;
  if (rowok) { show0 = st[0]; click0 = st[1]; g2w = st[2]; g2x = st[3]; state = st[4]; }
  const float score0 = (show0 - click0) * A.nonclk_coeff + click0 * A.click_coeff;
  // birth (what the pull of this step did in the reference): embed_w always, embedx if the score allows it
  const bool unborn = state == 0.f;
  bool has_x = state >= 2.f || (unborn && score0 >= A.embedx_threshold);

  // ---- embedx part
  float w[VEC], g[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { w[i] = 0.f; g[i] = 0.f; }
  float sq = 0.f;
  if (xlane) {
    if (unborn) {
#pragma unroll
      for (int i = 0; i < VEC; ++i)
        w[i] = (has_x && d0 + i < Dx) ? ps_init_value(A.seed, grow, 1 + d0 + i, A.initial_range) : 0.f;
    } else {
      vload<VEC>(w, r + L.embedx_off + d0);
    }
    if (has_x) {
      ps_segment_sum<VEC>(g, beg, end, spos, gx, gx.col + d0);
      const float sc = sqrtf(A.initial_g2sum / (A.initial_g2sum + g2x));
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        if (d0 + i < Dx) {
          w[i] = fminf(fmaxf(w[i] - A.lr * g[i] * sc, A.min_bound), A.max_bound);
          sq += g[i] * g[i];
        }
      }
    }
  }
  sq = group_sum<LANES>(sq);
  const float show1 = show0 + dshow, click1 = click0 + dclick;
  const float score1 = (show1 - click1) * A.nonclk_coeff + click1 * A.click_coeff;
  const bool create_x = !has_x && score1 >= A.embedx_threshold;   // what the NEXT pull would do
  if (xlane) {
    if (create_x) {
#pragma unroll
      for (int i = 0; i < VEC; ++i)
        w[i] = d0 + i < Dx ? ps_init_value(A.seed, grow, 1 + d0 + i, A.initial_range) : 0.f;
    }
    if (has_x || create_x || unborn) {
      if (VEC == 1 || d0 + VEC <= Dx) {
        vstore<VEC>(r + L.embedx_off + d0, w);
      } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i)
          if (d0 + i < Dx) r[L.embedx_off + d0 + i] = w[i];
      }
    }
  }
  // ---- embed_w + counters
  if (rowok && lg == 0) {
    float ew = unborn ? ps_init_value(A.seed, grow, 0, A.initial_range) : r[L.embed_off];
    float gwv[1] = {0.f};
    ps_segment_sum<1>(gwv, beg, end, spos, gw, gw.col);
    const float sc = sqrtf(A.initial_g2sum / (A.initial_g2sum + g2w));
    ew = fminf(fmaxf(ew - A.lr * gwv[0] * sc, A.min_bound), A.max_bound);
    r[L.embed_off] = ew;
    st[0] = show1;
    st[1] = click1;
    st[2] = g2w + gwv[0] * gwv[0];
    if (has_x) st[3] = g2x + sq / (float)Dx;
    st[4] = (has_x || create_x) ? 2.f : 1.f;
  }
}

// Narrow features without float4 row groups (slot_dnn: embedx_dim 8 at record offset 1, gradient rows of 9 floats):
// the kernel above then runs 8 lanes per feature = 8 features per wave, each walking uniq -> seg_off -> sorted_pos
// -> grad_index -> gradient row -> record one dependent round trip after the other (4.2 ms for 9 M features).
// Here ONE LANE owns a feature: 64 features per wave in flight, the index chain coalesced across lanes, the 64-B
// record the lane's private line.  Same rule, same ascending-position summation.
template <int DX>   // embedx floats held in registers (>= embedx_dim)
__global__ __launch_bounds__(kBlock) void ps_push_rows_narrow_kernel(
    rec_ps_layout L, int S, const int32_t* __restrict__ n_uniq, const int64_t* __restrict__ uniq,
    const int32_t* __restrict__ seg_off, const int32_t* __restrict__ spos, GradSrc gx, GradSrc gw,
    const int64_t* __restrict__ show, const int64_t* __restrict__ click, float* __restrict__ rec,
    rec_ps_accessor A) {
  // n_max is a capacity (the number of touched features is only known on the device): capped grid, strided loop
  const int nu = n_uniq[0];
  const int Dx = L.embedx_dim;
  for (int64_t u = (int64_t)blockIdx.x * kBlock + threadIdx.x; u < nu; u += (int64_t)gridDim.x * kBlock) {
  const int64_t row = uniq[u];
  const int beg = seg_off[u], end = seg_off[u + 1];
  float* r = rec + row * (int64_t)L.row_stride;
  float* st = r + L.stat_off;            // show, click, g2sum_w, g2sum_x, state
  const int64_t grow = row * A.row_mul + A.row_add;
  const float show0 = st[0], click0 = st[1], g2w = st[2], g2x = st[3], state = st[4];
  const bool unborn = state == 0.f;
  float w[DX];
#pragma unroll
  for (int d = 0; d < DX; ++d) w[d] = (!unborn && d < Dx) ? r[L.embedx_off + d] : 0.f;
  float ew = unborn ? 0.f : r[L.embed_off];

  // ---- one walk over the feature's occurrences: counters, embed_w gradient, embedx gradient
  const float score0 = (show0 - click0) * A.nonclk_coeff + click0 * A.click_coeff;
  const bool has_x = state >= 2.f || (unborn && score0 >= A.embedx_threshold);
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
    gwv += gw.grad[ps_grad_offset(gw.gl, pos, gw.pitch) + gw.col];
    if (has_x) {
      const float* a = gx.grad + ps_grad_offset(gx.gl, pos, gx.pitch) + gx.col;
#pragma unroll
      for (int d = 0; d < DX; ++d) g[d] += d < Dx ? a[d] : 0.f;
    }
  }
  // (the wide kernel adds the labels four at a time as integers, then to float: any grouping of integer partial
  // sums below 2^24 gives the same float)
  dclick = (float)lsum;

  float sq = 0.f;
  if (unborn) {
#pragma unroll
    for (int d = 0; d < DX; ++d) w[d] = (has_x && d < Dx) ? ps_init_value(A.seed, grow, 1 + d, A.initial_range) : 0.f;
    ew = ps_init_value(A.seed, grow, 0, A.initial_range);
  }
  if (has_x) {
    const float sc = sqrtf(A.initial_g2sum / (A.initial_g2sum + g2x));
#pragma unroll
    for (int d = 0; d < DX; ++d) {
      if (d < Dx) {
        w[d] = fminf(fmaxf(w[d] - A.lr * g[d] * sc, A.min_bound), A.max_bound);
        sq += g[d] * g[d];
      }
    }
  }
  const float show1 = show0 + dshow, click1 = click0 + dclick;
  const float score1 = (show1 - click1) * A.nonclk_coeff + click1 * A.click_coeff;
  const bool create_x = !has_x && score1 >= A.embedx_threshold;   // what the NEXT pull would do
  if (create_x) {
#pragma unroll
    for (int d = 0; d < DX; ++d) w[d] = d < Dx ? ps_init_value(A.seed, grow, 1 + d, A.initial_range) : 0.f;
  }
  if (has_x || create_x || unborn) {
#pragma unroll
    for (int d = 0; d < DX; ++d)
      if (d < Dx) r[L.embedx_off + d] = w[d];
  }
  const float scw = sqrtf(A.initial_g2sum / (A.initial_g2sum + g2w));
  ew = fminf(fmaxf(ew - A.lr * gwv * scw, A.min_bound), A.max_bound);
  r[L.embed_off] = ew;
  st[0] = show1;
  st[1] = click1;
  st[2] = g2w + gwv * gwv;
  if (has_x) st[3] = g2x + sq / (float)Dx;
  st[4] = (has_x || create_x) ? 2.f : 1.f;
  }
}

__global__ __launch_bounds__(kBlock) void ps_shrink_rows_kernel(int64_t N, rec_ps_layout L,
                                                                 float* __restrict__ rec, float decay,
                                                                 float delete_threshold, float nonclk,
                                                                 float clk, int64_t* __restrict__ n_deleted) {
  const int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (row >= N) return;
  float* r = rec + row * (int64_t)L.row_stride;
  float* st = r + L.stat_off;
  if (st[4] == 0.f) return;                       // unborn rows stay untouched (zero memory)
  const float show = st[0] * decay, click = st[1] * decay;
  const float score = (show - click) * nonclk + click * clk;
  if (score < delete_threshold) {                 // delete: the row is unborn again
    r[L.embed_off] = 0.f;
    for (int d = 0; d < L.embedx_dim; ++d) r[L.embedx_off + d] = 0.f;
    st[0] = st[1] = st[2] = st[3] = st[4] = 0.f;
    if (n_deleted) atomicAdd((unsigned long long*)n_deleted, 1ull);
  } else {
    st[0] = show;
    st[1] = click;
  }
}

static int check_layout(const rec_ps_layout* L) {
  REC_REQUIRE(L, REC_EINVAL, "layout is NULL");
  REC_REQUIRE(L->row_stride > 0 && L->embedx_dim >= 0 && L->embed_off >= 0 && L->embedx_off >= 0 &&
                  L->stat_off >= 0,
              REC_EINVAL, "bad record layout");
  REC_REQUIRE(L->embed_off < L->row_stride && L->embedx_off + L->embedx_dim <= L->row_stride &&
                  L->stat_off + 5 <= L->row_stride,
              REC_EINVAL, "record parts do not fit in row_stride %d", L->row_stride);
  // the five statistics must not overlap the weights
  const bool ov_x = L->stat_off < L->embedx_off + L->embedx_dim && L->embedx_off < L->stat_off + 5;
  const bool ov_w = L->embed_off >= L->stat_off && L->embed_off < L->stat_off + 5;
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
                  accessor->initial_range >= 0.f && accessor->row_mul >= 1 && accessor->row_add >= 0,
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
                     num_slots, n_uniq, uniq_rows, seg_offset, sorted_pos, gx, gw, show, click, rec, *accessor)
    if (Dx <= 4) REC_PS_NARROW(4); else if (Dx <= 8) REC_PS_NARROW(8);
    else if (Dx <= 12) REC_PS_NARROW(12); else REC_PS_NARROW(16);
#undef REC_PS_NARROW
    return check_launch("rec_ps_push_rows (narrow)");
  }
#define REC_PS_CASE(V, L_)                                                                             \
  if (lanes == L_) {                                                                                   \
    const int64_t grid = (n_max * L_ + kBlock - 1) / kBlock;                                           \
    REC_REQUIRE(grid < (1ll << 31), REC_ESHAPE, "too many rows");                                      \
    hipLaunchKernelGGL((ps_push_rows_kernel<V, L_>), dim3((unsigned)grid), dim3(kBlock), 0, st,        \
                       *layout, num_slots, n_uniq, uniq_rows, seg_offset, sorted_pos, gx, gw, show,    \
                       click, rec, *accessor);                                                         \
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
                                  const rec_ps_accessor* accessor, int64_t* n_deleted, void* stream) {
  if (int rc = check_layout(layout)) return rc;
  REC_REQUIRE(num_rows >= 0 && rec && accessor && show_click_decay_rate >= 0.f, REC_EINVAL, "bad arguments");
  if (num_rows == 0) return REC_OK;
  const int64_t grid = (num_rows + kBlock - 1) / kBlock;
  REC_REQUIRE(grid < (1ll << 31), REC_ESHAPE, "too many rows");
  hipLaunchKernelGGL(ps_shrink_rows_kernel, dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream,
                     num_rows, *layout, rec, show_click_decay_rate, delete_threshold, accessor->nonclk_coeff,
                     accessor->click_coeff, n_deleted);
  return check_launch("rec_ps_shrink_rows");
}
