"""NumPy restatement of the PS / gpubox sparse table accessor (oracle — test infrastructure only).

Reference call sites in /root/reference (configuration only — the arithmetic is NOT in that tree):
  models/rank/slot_dnn/config_online.yaml:57-89   accessor_class "SparseAccessor", fea_dim 11, embedx_dim 8,
      embedx_threshold 10, embed_sgd_param / embedx_sgd_param "SparseAdaGradSGDRule" (learning_rate 0.05,
      initial_g2sum 3.0, initial_range 1e-4, weight_bounds +-10), ctr_accessor_param (nonclk_coeff 0.1, click_coeff 1,
      base_threshold 1.5, delta_threshold 0.25, delta_keep_days 16, show_click_decay_rate 0.98, delete_threshold 0.8,
      delete_after_unseen_days 30)
  models/rank/dnn/net.py:67-82, dnn/static_model.py:86-94   sparse_embedding(size=[N, D+2]) + continuous_value_model,
      show = ones, click = label
  tools/static_gpubox_trainer.py:152-160,256       the table lives in the GPU parameter server (core.PSGPU)

The accessor itself is third-party code of the un-vendored dependency **PaddlePaddle** (README_EN.md:46,55 pins only
">= 2.0"; the PS classes named by the YAML exist from release/2.3 on).  This file restates the PUBLISHED source of
PaddlePaddle release/2.4 (github.com/PaddlePaddle/Paddle, tag v2.4.2):
  paddle/fluid/distributed/ps/table/sparse_sgd_rule.cc
      SparseAdaGradSGDRule::UpdateValueWork(w, sgd, grad, scale):
          float& g2sum = sgd[G2SumIndex()];  double add_g2sum = 0;
          for i < dim:  double scaled_grad = grad[i] / scale;
                        w[i] -= learning_rate_ * scaled_grad * sqrt(_initial_g2sum / (_initial_g2sum + g2sum));
                        BoundValue(w[i]);  add_g2sum += scaled_grad * scaled_grad;
          g2sum += add_g2sum / dim;
      SparseAdaGradSGDRule::InitValueWork(value, sgd, zero_init):
          value[i] = zero_init ? 0 : (uniform_real<double>() * 2 - 1) * _initial_range;  BoundValue;  sgd[g2sum] = 0
  paddle/fluid/distributed/ps/table/ctr_accessor.cc  (sparse_accessor.cc: the same Create / Update / Shrink / Save with
      its own value struct)
      Create:  unseen_days = delta_score = show = click = 0, slot = -1,
               _embed_sgd_rule->InitValue(embed_w, embed_g2sum)  [zero_init defaults to true: embed_w = 0],
               _embedx_sgd_rule->InitValue(embedx_w, embedx_g2sum, false)  [uniform(+-initial_range)]
      Update:  show += push_show; click += push_click; slot = push_slot;
               delta_score += (push_show - push_click) * nonclk_coeff + push_click * click_coeff;  unseen_days = 0;
               _embed_sgd_rule->UpdateValue(embed_w, embed_g2sum, push_embed_g, push_show)      [show_scale]
               _embedx_sgd_rule->UpdateValue(embedx_w, embedx_g2sum, push_embedx_g, push_show)
      NeedExtendMF:  (show - click) * nonclk_coeff + click * click_coeff >= embedx_threshold
      Shrink:  show *= show_click_decay_rate; click *= decay;
               delete if ShowClickScore(show, click) < delete_threshold || unseen_days > delete_after_unseen_days
      Save(param): 0 all; 1 (delta) / 2 (base): ShowClickScore >= base_threshold && delta_score >= delta_threshold
               (0 for param 2) && unseen_days <= delta_keep_days; 3 all
      UpdateStatAfterSave(param): 1: delta_score = 0 on the rows Save(1) selected; 3: unseen_days++
  paddle/fluid/distributed/ps/table/memory_sparse_table.cc
      PullSparse: a missing key is created WITHOUT its embedx part and reads as embed_w = 0, embedx = 0;
      PushSparse: a value without embedx is updated (its embedx gradient is dropped), then extended
               (embedx = uniform, embedx_g2sum = 0) when NeedExtendMF holds on the UPDATED counters.
  paddle/fluid/framework/fleet/heter_ps/optimizer.cuh.h  (the gpubox path of tools/static_gpubox_trainer.py):
      SparseAdagradOptimizer::update_value_work / dy_mf_update_value — the same rule (double scaled_grad = g / g_show,
      ratio = lr * sqrt(g0 / (g0 + g2sum)), bounds, g2sum += add / n), embedx created inside the push once the score
      reaches mf_create_thresholds.
  The pushed gradient is the gradient of the SUMMED loss: the trainer multiplies by the batch size before the push
  (heter_ps feature_value.cu PushCopy: `* -1. * bs`; CPU workers: scale_sparse_gradient_with_batch_size = true) —
  `grad_scale` below — and the rule divides it by the pushed show (= occurrences of the key in the batch).

**Parity unpinned**: Paddle is not installable here and /root/reference holds no expected values for this path, so
this restatement is anchored on the published source text above and on hand-computed known answers
(tests/test_ps_accessor_kat.py), not on outputs of the reference.  Deliberate, stated differences:
  * creation values are a counter-based function of (seed, row, element) instead of Paddle's unseeded thread-local
    engine (not reproducible between two Paddle runs either); distribution uniform(+-initial_range) as in
    sparse_sgd_rule.cc (heter_ps draws curand_uniform * range, i.e. (0, range]);
  * `slot` (an annotation for the save converter, no arithmetic) is not stored;
  * push_show = 0 (cannot happen: every occurrence carries show >= 1) divides by 1 instead of 0.
Arithmetic: the typed evaluation of the C++ text (round 4: `grad[i] / scale` is the float division the text has, on the
float pushed gradient — round 3 divided in double) — `_initial_g2sum / (_initial_g2sum + g2sum)` and its sqrt in
float (all operands are float), scaled_grad, the product with the learning rate and add_g2sum in double, `w[i] -=` and
`g2sum +=` rounded to float on store.
"""
import numpy as np

# statistics of a feature value, consecutive floats at lay["stat_off"]
SHOW, CLICK, G2SUM_W, G2SUM_X, STATE, DELTA_SCORE, UNSEEN_DAYS = range(7)
NUM_STATS = 7
# STATE: 0 = no such key (zero memory), 1 = value without its embedx part, 2 = embedx exists


def cvm_lookup(rec, ids, D, padding_idx=None):
    """sparse_embedding returns [show, click, embed...]; CVM(use_cvm=False) drops the two CVM columns (App. B-8)."""
    out = rec[ids][..., 4:4 + D]      # engine record: [show | click | g2sum_w | g2sum_x | W(D) | pad]
    if padding_idx is not None:
        out = out * (ids != padding_idx)[..., None]
    return out


def adagrad_rows(rec, D, uniq, merged, shows, clicks, lr=0.05, initial_g2sum=3.0, bounds=(-10.0, 10.0)):
    """rec_sparse_adagrad_rows (the r01 CVM-record rule, kept for its kernel): in place on rec [N, stride] rows `uniq`
    with merged gradients [U,D], show/click increments [U]; float arithmetic, no show scaling."""
    f = np.float32
    for u, row in enumerate(uniq):
        r = rec[row]
        r[0] += f(shows[u])
        r[1] += f(clicks[u])
        g2w, g2x = r[2], r[3]
        sw = np.sqrt(f(initial_g2sum) / (f(initial_g2sum) + g2w))
        sx = np.sqrt(f(initial_g2sum) / (f(initial_g2sum) + g2x))
        g = merged[u].astype(f)
        scale = np.full(D, sx, f)
        scale[0] = sw
        r[4:4 + D] = np.clip(r[4:4 + D] - f(lr) * g * scale, f(bounds[0]), f(bounds[1]))
        r[2] = g2w + g[0] * g[0]
        if D > 1:
            acc = f(0)
            for d in range(1, D):
                acc = acc + g[d] * g[d]
            r[3] = g2x + acc / f(D - 1)


_M64 = (1 << 64) - 1


def _mix64(k):
    k &= _M64
    k ^= k >> 33
    k = (k * 0xff51afd7ed558ccd) & _M64
    k ^= k >> 33
    k = (k * 0xc4ceb9fe1a85ec53) & _M64
    k ^= k >> 33
    return k


def init_value(seed, row, d, rng_range):
    """Creation value of element d of feature `row` (uniform(+-range), counter based): element 0 = embed_w (used only
    when the accessor is configured with embed_zero_init = False), element 1+j = embedx[j]."""
    h = _mix64(seed ^ _mix64((int(row) * 0x9E3779B97F4A7C15 + (d + 1)) & _M64))
    u = np.float32(h >> 40) * np.float32(1.0 / 16777216.0)
    return (np.float32(2.0) * u - np.float32(1.0)) * np.float32(rng_range)


def _rule(acc, part):
    """(lr, initial_g2sum, lo, hi, initial_range) of embed_sgd_param / embedx_sgd_param; embedx_* keys default to the
    embed ones (config_online.yaml gives both rules the same numbers)."""
    p = "embedx_" if part == "embedx" else ""
    g = lambda k: acc.get(p + k, acc[k])
    b = g("bounds")
    return np.float32(g("lr")), np.float32(g("initial_g2sum")), np.float32(b[0]), np.float32(b[1]), g("initial_range")


def update_value_work(w, g2sum, grad, scale, lr, g0, lo, hi):
    """SparseAdaGradSGDRule::UpdateValueWork.  w: float32 array (updated in place); g2sum: float32 scalar;
    grad: the pushed gradient (float32); scale: the pushed show.  -> new g2sum (float32)."""
    ratio = np.float64(np.sqrt(np.float32(g0) / (np.float32(g0) + np.float32(g2sum))))     # float expression
    add = np.float64(0.0)
    for i in range(len(w)):
        # `double scaled_grad = grad[i] / scale;` with `const float* grad` and `float scale`: a FLOAT division, widened
        # afterwards (a double division differs by up to one float ulp of the quotient)
        sg = np.float64(np.float32(grad[i]) / np.float32(scale))
        w[i] = np.float32(np.float64(w[i]) - np.float64(lr) * sg * ratio)
        w[i] = min(max(w[i], lo), hi)
        add += sg * sg
    return np.float32(np.float64(g2sum) + add / np.float64(len(w)))


def score(show, click, acc):
    f = np.float32
    return (f(show) - f(click)) * f(acc["nonclk_coeff"]) + f(click) * f(acc["click_coeff"])


def push_rows(rec, lay, uniq, g_embed, g_embedx, dshow, dclick, acc):
    """MemorySparseTable::PushSparse + CtrCommonAccessor::Update on rows `uniq` of rec [N, stride] (in place).
    lay = dict(embed_off, embedx_off, embedx_dim, stat_off); acc = dict(lr, initial_g2sum, bounds, initial_range,
    [embedx_lr, embedx_initial_g2sum, embedx_bounds, embedx_initial_range,] embedx_threshold, nonclk_coeff,
    click_coeff, seed, [show_scale = True, grad_scale = 1.0, embed_zero_init = True, row_mul, row_add]);
    g_embed [U], g_embedx [U, Dx]: gradients merged over the key's occurrences (float32 sums in ascending position);
    dshow / dclick [U]: pushed show / click."""
    f = np.float32
    Dx, eo, xo, so = lay["embedx_dim"], lay["embed_off"], lay["embedx_off"], lay["stat_off"]
    lr_w, g0_w, lo_w, hi_w, range_w = _rule(acc, "embed")
    lr_x, g0_x, lo_x, hi_x, range_x = _rule(acc, "embedx")
    mul, add = int(acc.get("row_mul", 1)), int(acc.get("row_add", 0))   # identity of a shard's row: row*mul + add
    gs = np.float64(acc.get("grad_scale", 1.0))
    for u, row in enumerate(uniq):
        r = rec[row]
        grow = int(row) * mul + add
        st = r[so:so + NUM_STATS]
        if st[STATE] == 0:                       # Create: a value without embedx; counters, g2sums, embedx stay 0
            r[eo] = f(0) if acc.get("embed_zero_init", True) else init_value(acc["seed"], grow, 0, range_w)
            st[STATE] = f(1)
        push_show, push_click = f(dshow[u]), f(dclick[u])
        st[SHOW] += push_show
        st[CLICK] += push_click
        st[DELTA_SCORE] += (push_show - push_click) * f(acc["nonclk_coeff"]) + push_click * f(acc["click_coeff"])
        st[UNSEEN_DAYS] = f(0)
        scale = f(push_show) if (acc.get("show_scale", True) and push_show > 0) else f(1.0)
        # the pushed gradient = merged gradient x grad_scale (the batch size: gradient of the SUMMED loss) is a FLOAT
        # in the push value (heter_ps PushCopy stores `g * -1. * bs` into a float field): one rounding of the product
        ew = r[eo:eo + 1]
        st[G2SUM_W] = update_value_work(ew, st[G2SUM_W], [f(np.float64(g_embed[u]) * gs)], scale, lr_w, g0_w, lo_w, hi_w)
        if st[STATE] >= 2:
            gx = (np.asarray(g_embedx[u], np.float64) * gs).astype(np.float32)
            xw = r[xo:xo + Dx]
            st[G2SUM_X] = update_value_work(xw, st[G2SUM_X], gx, scale, lr_x, g0_x, lo_x, hi_x)
        elif Dx > 0 and score(st[SHOW], st[CLICK], acc) >= f(acc["embedx_threshold"]):   # NeedExtendMF on the UPDATED value
            for j in range(Dx):
                r[xo + j] = min(max(init_value(acc["seed"], grow, 1 + j, range_x), lo_x), hi_x)
            st[G2SUM_X] = f(0)
            st[STATE] = f(2)


def pull_value(rec, lay, row, acc, D_lookup):
    """What a lookup of `row` returns for the slot layout (W = [embed_w, embedx...]) — PullSparse + Select: the stored
    floats; a missing key reads as zeros (embed_zero_init = False: embed_w reads as its creation value, which the
    first push then stores)."""
    r = rec[row]
    out = r[lay["embed_off"]:lay["embed_off"] + D_lookup].copy()
    if r[lay["stat_off"] + STATE] == 0 and not acc.get("embed_zero_init", True):
        g = int(row) * int(acc.get("row_mul", 1)) + int(acc.get("row_add", 0))
        out[0] = init_value(acc["seed"], g, 0, _rule(acc, "embed")[4])
    return out


def pull_deepfm(rec, lay, rows, acc):
    """Lookup of the 'deepfm' record layout (embedx = the D-dim embedding, embed_w = the first-order weight):
    -> (W [n, Dx], W1 [n])."""
    Dx, eo, xo, so = lay["embedx_dim"], lay["embed_off"], lay["embedx_off"], lay["stat_off"]
    mul, add = int(acc.get("row_mul", 1)), int(acc.get("row_add", 0))
    W = np.zeros((len(rows), Dx), np.float32)
    W1 = np.zeros(len(rows), np.float32)
    for i, row in enumerate(rows):
        r = rec[int(row)]
        W[i], W1[i] = r[xo:xo + Dx], r[eo]
        if r[so + STATE] == 0 and not acc.get("embed_zero_init", True):
            W1[i] = init_value(acc["seed"], int(row) * mul + add, 0, _rule(acc, "embed")[4])
    return W, W1


def shrink_rows(rec, lay, acc, decay, delete_threshold, delete_after_unseen_days=np.inf):
    """CtrCommonAccessor::Shrink over the table: decay the counters of every existing value; delete (zero) those whose
    score fell below delete_threshold or that were not seen for more than delete_after_unseen_days.  -> number deleted."""
    f = np.float32
    so = lay["stat_off"]
    deleted = 0
    for row in range(rec.shape[0]):
        r = rec[row]
        st = r[so:so + NUM_STATS]
        if st[STATE] == 0:
            continue
        show, click = st[SHOW] * f(decay), st[CLICK] * f(decay)
        if score(show, click, acc) < f(delete_threshold) or st[UNSEEN_DAYS] > f(delete_after_unseen_days):
            r[lay["embed_off"]] = 0
            r[lay["embedx_off"]:lay["embedx_off"] + lay["embedx_dim"]] = 0
            st[:] = 0
            deleted += 1
        else:
            st[SHOW], st[CLICK] = show, click
    return deleted


def save_select(rec, lay, acc, param, base_threshold=1.5, delta_threshold=0.25, delta_keep_days=16.0):
    """CtrCommonAccessor::Save(value, param) for every row -> bool mask of the rows a save of this kind writes, then
    UpdateStatAfterSave(value, param) applied in place (1: delta_score = 0 on the selected rows; 2: delta_score = 0 on
    the selected rows [done by Save itself in the C++]; 3: unseen_days += 1 on every existing row)."""
    f = np.float32
    so = lay["stat_off"]
    N = rec.shape[0]
    mask = np.zeros(N, bool)
    for row in range(N):
        st = rec[row, so:so + NUM_STATS]
        if st[STATE] == 0:
            continue
        if param in (1, 2):
            dth = f(0) if param == 2 else f(delta_threshold)
            ok = (score(st[SHOW], st[CLICK], acc) >= f(base_threshold) and st[DELTA_SCORE] >= dth
                  and st[UNSEEN_DAYS] <= f(delta_keep_days))
            mask[row] = ok
            if ok:
                st[DELTA_SCORE] = f(0)
        else:
            mask[row] = True
            if param == 3:
                st[UNSEEN_DAYS] += f(1)
    return mask
