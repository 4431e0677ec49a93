"""NumPy restatement of the PS / gpubox sparse path (oracle — test infrastructure only).

Reference call sites: models/rank/dnn/net.py:67-82 (sparse_embedding(size=[N, D+2]) +
continuous_value_model(emb, show_click, use_cvm=False)), dnn/static_model.py:86-94 (show = ones, click =
label), slot_dnn/config_online.yaml:57-79 (SparseAdaGradSGDRule lr 0.05, initial_g2sum 3.0, bounds +-10).
The accessor arithmetic itself is in the un-vendored PaddlePaddle PS code — **parity unpinned**; the
formula is the one fixed by SURVEY.md App. B-13.
"""
import numpy as np


def cvm_lookup(rec, ids, D, padding_idx=None):
    """sparse_embedding returns [show, click, embed...]; CVM(use_cvm=False) drops the two CVM columns (App. B-8)."""
    out = rec[ids][..., 4:4 + D]      # engine record: [show | click | g2sum_w | g2sum_x | W(D) | pad]
    if padding_idx is not None:
        out = out * (ids != padding_idx)[..., None]
    return out


def adagrad_rows(rec, D, uniq, merged, shows, clicks, lr=0.05, initial_g2sum=3.0, bounds=(-10.0, 10.0)):
    """In place on rec [N, stride] rows `uniq` with merged gradients [U,D], show/click increments [U]."""
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


# ------------------------------------------------------------------------------------------------
# The full accessor (slot_dnn/config_online.yaml:57-89): lazy birth, embedx_threshold, shrink.
# Restated from the published behaviour of Paddle's CtrCommonAccessor / SparseAdaGradSGDRule [EXT] — parity
# unpinned (no reference source or vector exists in /root/reference); the engine's kernels are held to THIS text.
# ------------------------------------------------------------------------------------------------
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
    """Creation value of element d of feature `row` (uniform(+-range), counter based): element 0 = embed_w,
    element 1+j = embedx[j]."""
    h = _mix64(seed ^ _mix64((int(row) * 0x9E3779B97F4A7C15 + (d + 1)) & _M64))
    u = np.float32(h >> 40) * np.float32(1.0 / 16777216.0)
    return (np.float32(2.0) * u - np.float32(1.0)) * np.float32(rng_range)


def push_rows(rec, lay, uniq, g_embed, g_embedx, dshow, dclick, acc):
    """CtrCommonAccessor::Update on rows `uniq` of rec [N, stride] (in place).
    lay = dict(embed_off, embedx_off, embedx_dim, stat_off); acc = dict(lr, initial_g2sum, bounds, initial_range,
    embedx_threshold, nonclk_coeff, click_coeff, seed); g_embed [U], g_embedx [U, Dx] merged gradients."""
    f = np.float32
    Dx, eo, xo, so = lay["embedx_dim"], lay["embed_off"], lay["embedx_off"], lay["stat_off"]
    g0, lr = f(acc["initial_g2sum"]), f(acc["lr"])
    lo, hi = f(acc["bounds"][0]), f(acc["bounds"][1])
    mul, add = int(acc.get("row_mul", 1)), int(acc.get("row_add", 0))   # identity of a shard's row: row*mul + add
    for u, row in enumerate(uniq):
        r = rec[row]
        row = int(row) * mul + add
        show0, click0, g2w, g2x, state = r[so], r[so + 1], r[so + 2], r[so + 3], r[so + 4]
        score0 = (show0 - click0) * f(acc["nonclk_coeff"]) + click0 * f(acc["click_coeff"])
        unborn = state == 0
        has_x = state >= 2 or (unborn and score0 >= f(acc["embedx_threshold"]))
        if unborn:                                    # the pull of this step created the feature
            r[eo] = init_value(acc["seed"], row, 0, acc["initial_range"])
            for j in range(Dx):
                r[xo + j] = init_value(acc["seed"], row, 1 + j, acc["initial_range"]) if has_x else f(0)
        show1, click1 = show0 + f(dshow[u]), click0 + f(dclick[u])
        gw = f(g_embed[u])
        r[eo] = np.clip(r[eo] - lr * gw * np.sqrt(g0 / (g0 + g2w)), lo, hi)
        r[so + 2] = g2w + gw * gw
        if has_x:
            gx = g_embedx[u].astype(f)
            r[xo:xo + Dx] = np.clip(r[xo:xo + Dx] - lr * gx * np.sqrt(g0 / (g0 + g2x)), lo, hi)
            acc_sq = f(0)
            for j in range(Dx):
                acc_sq = acc_sq + gx[j] * gx[j]
            r[so + 3] = g2x + acc_sq / f(Dx)
        r[so], r[so + 1] = show1, click1
        score1 = (show1 - click1) * f(acc["nonclk_coeff"]) + click1 * f(acc["click_coeff"])
        if not has_x and score1 >= f(acc["embedx_threshold"]):   # the next pull would extend the value
            for j in range(Dx):
                r[xo + j] = init_value(acc["seed"], row, 1 + j, acc["initial_range"])
            has_x = True
        r[so + 4] = f(2.0) if has_x else f(1.0)


def pull_value(rec, lay, row, acc, D_lookup):
    """What a lookup of `row` returns for the slot layout (W = [embed_w, embedx...]): an unborn row reads as its
    creation values (embedx only if the threshold allows creation at score 0)."""
    r = rec[row]
    if r[lay["stat_off"] + 4] != 0:
        return r[lay["embed_off"]:lay["embed_off"] + D_lookup].copy()
    row = int(row) * int(acc.get("row_mul", 1)) + int(acc.get("row_add", 0))
    dims = D_lookup if acc["embedx_threshold"] <= 0 else 1
    return np.array([init_value(acc["seed"], row, d, acc["initial_range"]) if d < dims else np.float32(0)
                     for d in range(D_lookup)], np.float32)


def shrink_rows(rec, lay, acc, decay, delete_threshold):
    """Shrink: decay the counters of every born row; delete (zero) rows whose score fell below the threshold."""
    f = np.float32
    so = lay["stat_off"]
    deleted = 0
    for row in range(rec.shape[0]):
        r = rec[row]
        if r[so + 4] == 0:
            continue
        show, click = r[so] * f(decay), r[so + 1] * f(decay)
        score = (show - click) * f(acc["nonclk_coeff"]) + click * f(acc["click_coeff"])
        if score < f(delete_threshold):
            r[lay["embed_off"]] = 0
            r[lay["embedx_off"]:lay["embedx_off"] + lay["embedx_dim"]] = 0
            r[so:so + 5] = 0
            deleted += 1
        else:
            r[so], r[so + 1] = show, click
    return deleted


def pull_deepfm(rec, lay, rows, acc):
    """Lookup of the 'deepfm' record layout (embedx = the D-dim embedding, embed_w = the first-order weight):
    -> (W [n, Dx], W1 [n]); unborn rows read as their creation values."""
    Dx, eo, xo, so = lay["embedx_dim"], lay["embed_off"], lay["embedx_off"], lay["stat_off"]
    mul, add = int(acc.get("row_mul", 1)), int(acc.get("row_add", 0))
    W = np.zeros((len(rows), Dx), np.float32)
    W1 = np.zeros(len(rows), np.float32)
    for i, row in enumerate(rows):
        r = rec[int(row)]
        if r[so + 4] != 0:
            W[i], W1[i] = r[xo:xo + Dx], r[eo]
        else:
            g = int(row) * mul + add
            W1[i] = init_value(acc["seed"], g, 0, acc["initial_range"])
            if acc["embedx_threshold"] <= 0:
                W[i] = [init_value(acc["seed"], g, 1 + j, acc["initial_range"]) for j in range(Dx)]
    return W, W1
