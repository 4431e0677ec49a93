"""NumPy restatement of the DeepFM hot path (oracle — test infrastructure only).

Every function cites the reference lines it follows (paths relative to
/root/reference).  Paddle-op semantics marked [EXT] live in the un-vendored
`paddlepaddle` wheel; they are taken from SURVEY.md Appendix B.

All arithmetic runs in the dtype of the inputs (float32 for parity runs,
float64 to bound the float32 rounding noise when a tolerance is chosen).
"""
import numpy as np

PADDING_IDX = 0  # models/rank/deepfm/net.py:69,81  (padding_idx=0)


# --------------------------------------------------------------------------
# E — embedding lookup                      models/rank/deepfm/net.py:66-86,107-108,117
# --------------------------------------------------------------------------
def effective_rows(ids, padding_idx=PADDING_IDX, slot_offsets=None):
    """ids [B,S] int64 -> (rows [B,S] int64, valid [B,S] bool).

    `valid` is False where the lookup hits padding_idx (output row is zero and
    the row receives no gradient: nn.Embedding(padding_idx=) [EXT], Appendix B-1).
    slot_offsets (optional, [S] int64) is the "26 tables x 1M rows == one table
    with slot offset s*1e6" layout of BASELINE config 2a; the padding test is on
    the raw id, the offset is applied afterwards.
    """
    ids = np.asarray(ids, dtype=np.int64)
    valid = np.ones(ids.shape, dtype=bool) if padding_idx is None or padding_idx < 0 \
        else ids != padding_idx
    rows = ids.copy()
    if slot_offsets is not None:
        rows = rows + np.asarray(slot_offsets, dtype=np.int64)[None, :]
    return rows, valid


def embedding_lookup(W, ids, padding_idx=PADDING_IDX, slot_offsets=None):
    """out[b,s,:] = 0 if ids[b,s]==padding_idx else W[row(b,s),:]   (deepfm/net.py:108,117)."""
    rows, valid = effective_rows(ids, padding_idx, slot_offsets)
    out = W[np.where(valid, rows, 0)]
    out = out * valid[..., None].astype(W.dtype)
    return out


# --------------------------------------------------------------------------
# F1/F2 — FM first and second order         models/rank/deepfm/net.py:105-139
# --------------------------------------------------------------------------
def fm_forward(ids, dense, W1, W, dense_w_one, dense_w,
               padding_idx=PADDING_IDX, slot_offsets=None):
    """Returns y_first_order [B,1], y_second_order [B,1], feat_embeddings [B,F,D].

    ids   [B,S] int64   = paddle.concat(sparse_inputs, axis=1)         net.py:107
    dense [B,Dn] f32
    W1 [N,1], W [N,D]   embedding_one / embedding weights              net.py:66-86
    dense_w_one [Dn], dense_w [1,Dn,D]                                 net.py:89-103
    """
    dt = W.dtype
    dense = dense.astype(dt)
    # first order  (net.py:107-114)
    sparse_emb_one = embedding_lookup(W1, ids, padding_idx, slot_offsets)      # [B,S,1]
    dense_emb_one = (dense * dense_w_one.astype(dt))[:, :, None]               # [B,Dn,1]
    y1 = sparse_emb_one.sum(axis=1, dtype=dt) + dense_emb_one.sum(axis=1, dtype=dt)  # [B,1]
    # second order (net.py:117-137)
    sparse_embeddings = embedding_lookup(W, ids, padding_idx, slot_offsets)    # [B,S,D]
    dense_embeddings = dense[:, :, None] * dense_w.astype(dt)                  # [B,Dn,D]
    feat = np.concatenate([sparse_embeddings, dense_embeddings], axis=1)       # [B,F,D]
    summed = feat.sum(axis=1, dtype=dt)                                        # [B,D]
    summed_sq = np.square(summed)
    sq_sum = np.square(feat).sum(axis=1, dtype=dt)
    y2 = (0.5 * (summed_sq - sq_sum).sum(axis=1, keepdims=True, dtype=dt)).astype(dt)
    return y1.astype(dt), y2, feat


# --------------------------------------------------------------------------
# M — top MLP                               models/rank/deepfm/net.py:142-174
# --------------------------------------------------------------------------
def dnn_forward(feat, weights, biases, return_acts=False):
    """Linear(+bias) -> ReLU ... -> Linear.  Paddle Linear.weight is [in,out] (App. B-2)."""
    x = feat.reshape(feat.shape[0], -1)                   # net.py:170-171
    acts = [x]
    n = len(weights)
    for i, (w, b) in enumerate(zip(weights, biases)):
        x = x @ w + b
        if i < n - 1:                                      # acts = relu..., None  net.py:152-153
            x = np.maximum(x, 0)
        acts.append(x)
    return (x, acts) if return_acts else x


def sigmoid(z):
    return 1.0 / (1.0 + np.exp(-z))


def deepfm_forward(ids, dense, params, padding_idx=PADDING_IDX, slot_offsets=None):
    """predict = sigmoid(y1 + y2 + y_dnn)       net.py:41-49  (self.bias unused: App. B-14)."""
    y1, y2, feat = fm_forward(ids, dense, params["W1"], params["W"],
                              params["dense_w_one"], params["dense_w"],
                              padding_idx, slot_offsets)
    y_dnn = dnn_forward(feat, params["mlp_w"], params["mlp_b"])
    z = y1 + y2 + y_dnn
    return sigmoid(z), z, (y1, y2, feat, y_dnn)


# --------------------------------------------------------------------------
# L — loss                                   models/rank/deepfm/dygraph_model.py:53-58
# --------------------------------------------------------------------------
LOG_LOSS_EPS = 1e-4  # paddle.nn.functional.log_loss default epsilon [EXT] (App. B-4)


def log_loss_mean(pred, label, eps=LOG_LOSS_EPS):
    t = label.astype(pred.dtype)
    cost = -t * np.log(pred + pred.dtype.type(eps)) - (1 - t) * np.log(1 - pred + pred.dtype.type(eps))
    return cost.mean(dtype=pred.dtype)


def log_loss_mean_grad_z(pred, label, eps=LOG_LOSS_EPS):
    """d mean(log_loss(sigmoid(z), t)) / dz   ([B,1])."""
    t = label.astype(pred.dtype)
    eps = pred.dtype.type(eps)
    dp = (-t / (pred + eps) + (1 - t) / (1 - pred + eps)) / pred.dtype.type(pred.shape[0])
    return dp * pred * (1 - pred)


# --------------------------------------------------------------------------
# G — backward of DNN + FM + embedding (what loss.backward() does, tools/trainer.py:151)
# --------------------------------------------------------------------------
def dnn_backward(dy, acts, weights):
    """Returns (dfeat_flat [B,F*D], dW list, db list)."""
    n = len(weights)
    dws, dbs = [None] * n, [None] * n
    g = dy
    for i in reversed(range(n)):
        if i < n - 1:
            g = g * (acts[i + 1] > 0)
        dws[i] = acts[i].T @ g
        dbs[i] = g.sum(axis=0)
        g = g @ weights[i].T
    return g, dws, dbs


def fm_backward(ids, dense, feat, d_feat_dnn, dy1, dy2,
                padding_idx=PADDING_IDX, slot_offsets=None):
    """Gradient of (y1, y2, feat) w.r.t. FM parameters.

    d_feat_dnn [B,F,D] : gradient arriving at feat_embeddings from the DNN branch
    dy1, dy2   [B,1]   : gradients of the two FM logits
    Returns dict with
      rows      [B*S] int64  flattened row ids (SelectedRows.rows, duplicates kept; App. B-1)
      row_valid [B*S] bool   False for padding_idx hits (those rows get no gradient)
      row_grad  [B*S,D]      SelectedRows.value for `embedding`
      row_grad1 [B*S,1]      SelectedRows.value for `embedding_one`
      d_dense_w [1,Dn,D], d_dense_w_one [Dn], d_feat [B,F,D]
    """
    B, S = ids.shape
    dt = feat.dtype
    summed = feat.sum(axis=1, keepdims=True, dtype=dt)                     # [B,1,D]
    d_feat = d_feat_dnn + dy2[:, :, None] * (summed - feat)                # d y2/d feat = S - feat
    rows, valid = effective_rows(ids, padding_idx, slot_offsets)
    row_grad = d_feat[:, :S, :].reshape(B * S, -1)
    row_grad1 = np.repeat(dy1, S, axis=1).reshape(B * S, 1)
    d_dense_w = (dense.astype(dt)[:, :, None] * d_feat[:, S:, :]).sum(axis=0, dtype=dt)[None]
    d_dense_w_one = (dense.astype(dt) * dy1).sum(axis=0, dtype=dt)
    return dict(rows=rows.reshape(-1), row_valid=valid.reshape(-1),
                row_grad=row_grad, row_grad1=row_grad1,
                d_dense_w=d_dense_w, d_dense_w_one=d_dense_w_one, d_feat=d_feat)


def deepfm_loss_and_grads(ids, dense, label, params, padding_idx=PADDING_IDX, slot_offsets=None):
    """One train_forward + backward (deepfm/dygraph_model.py:76-88, tools/trainer.py:148-151)."""
    pred, z, (y1, y2, feat, y_dnn) = deepfm_forward(ids, dense, params, padding_idx, slot_offsets)
    loss = log_loss_mean(pred, label)
    dz = log_loss_mean_grad_z(pred, label)
    _, acts = dnn_forward(feat, params["mlp_w"], params["mlp_b"], return_acts=True)
    dflat, dws, dbs = dnn_backward(dz, acts, params["mlp_w"])
    g = fm_backward(ids, dense, feat, dflat.reshape(feat.shape), dz, dz, padding_idx, slot_offsets)
    g.update(mlp_dw=dws, mlp_db=dbs, loss=loss, pred=pred, dz=dz, feat=feat, y1=y1, y2=y2)
    return g


# --------------------------------------------------------------------------
# SelectedRows merge (MergeAdd) [EXT] — duplicate rows summed, rows sorted ascending
# --------------------------------------------------------------------------
def merge_rows(rows, valid, values):
    """Returns (uniq_rows [U] ascending, merged [U,D], counts [U]).

    Summation order inside one row is ascending position (b*S+s) — the order the
    engine's stable grouping reproduces, so float32 results are comparable bit-for-bit
    for short segments and within 1e-6 relative otherwise.
    """
    rows = rows[valid]
    values = values[valid]
    order = np.argsort(rows, kind="stable")
    srows = rows[order]
    uniq, start, counts = np.unique(srows, return_index=True, return_counts=True)
    merged = np.zeros((len(uniq), values.shape[1]), dtype=values.dtype)
    sv = values[order]
    for u in range(len(uniq)):          # sequential fp sum in position order
        acc = np.zeros(values.shape[1], dtype=values.dtype)
        for k in range(start[u], start[u] + counts[u]):
            acc = acc + sv[k]
        merged[u] = acc
    return uniq, merged, counts


def group_ids(rows, valid):
    """Integer part of the merge: (sorted positions, unique rows, segment offsets).  Bit-exact target."""
    pos = np.nonzero(valid)[0]
    order = np.argsort(rows[pos], kind="stable")
    spos = pos[order]
    srows = rows[spos]
    uniq, start = np.unique(srows, return_index=True)
    offs = np.concatenate([start, [len(srows)]]).astype(np.int64)
    return spos.astype(np.int64), uniq.astype(np.int64), offs


# --------------------------------------------------------------------------
# O — optimizer: paddle.optimizer.Adam [EXT]  (deepfm/dygraph_model.py:61-65; App. B-3)
# --------------------------------------------------------------------------
def adam_update(p, m, v, g, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8):
    """In-place dense Adam, Paddle formula:
         lr_t = lr*sqrt(1-b2^t)/(1-b1^t);  p -= lr_t * m / (sqrt(v) + eps*sqrt(1-b2^t))
    `step` is t (1-based).  Arrays are updated in place and returned."""
    dt = p.dtype
    b1p = dt.type(beta1) ** step
    b2p = dt.type(beta2) ** step
    lr_t = dt.type(lr) * np.sqrt(dt.type(1) - b2p) / (dt.type(1) - b1p)
    m[...] = dt.type(beta1) * m + (dt.type(1) - dt.type(beta1)) * g
    v[...] = dt.type(beta2) * v + (dt.type(1) - dt.type(beta2)) * g * g
    p[...] = p - lr_t * (m / (np.sqrt(v) + dt.type(eps) * np.sqrt(dt.type(1) - b2p)))
    return p, m, v


def adam_update_rows(P, M, V, uniq_rows, merged_grad, step, **kw):
    """lazy_mode=True Adam (deepfm/static_model.py:83-84): only rows present in the merged
    gradient are touched."""
    p, m, v = P[uniq_rows], M[uniq_rows], V[uniq_rows]
    adam_update(p, m, v, merged_grad, step, **kw)
    P[uniq_rows], M[uniq_rows], V[uniq_rows] = p, m, v


def adam_update_dense_equivalent(P, M, V, uniq_rows, merged_grad, step, **kw):
    """lazy_mode=False (dygraph default): every row moves, absent rows use g=0."""
    g = np.zeros_like(P)
    g[uniq_rows] = merged_grad
    adam_update(P, M, V, g, step, **kw)


# --------------------------------------------------------------------------
# A — AUC     paddle.metric.Auc [EXT] App. B-5; tools/utils/utils_single.py:160-206
# --------------------------------------------------------------------------
def auc_histogram(pred, label, num_thresholds=4095):
    """bucket = int(p * num_thresholds); stat_pos/neg int64 [num_thresholds+1]."""
    p = np.asarray(pred, dtype=np.float32).reshape(-1)
    t = np.asarray(label).reshape(-1)
    bucket = (p * np.float32(num_thresholds)).astype(np.int64)
    bucket = np.clip(bucket, 0, num_thresholds)
    pos = np.bincount(bucket[t != 0], minlength=num_thresholds + 1).astype(np.int64)
    neg = np.bincount(bucket[t == 0], minlength=num_thresholds + 1).astype(np.int64)
    return pos, neg


def auc_from_buckets(stat_pos, stat_neg):
    """Trapezoid sweep from the top bucket down — utils_single.py:183-204 line by line."""
    num_bucket = len(stat_pos)
    area = 0.0
    pos = 0.0
    neg = 0.0
    total = 0
    for i in range(num_bucket):
        index = num_bucket - 1 - i
        new_pos = pos + float(stat_pos[index])
        total += int(stat_pos[index])
        new_neg = neg + float(stat_neg[index])
        total += int(stat_neg[index])
        area += (new_neg - neg) * (pos + new_pos) / 2
        pos = new_pos
        neg = new_neg
    if pos * neg == 0 or total == 0:
        return 0.5
    return area / (pos * neg)


# --------------------------------------------------------------------------
# P — multi-slot variable-length lookup + sum-pool   models/rank/slot_dnn/net.py:63-75
# --------------------------------------------------------------------------
def sequence_pool_sum(W, ids, lod, padding_idx=PADDING_IDX):
    """bow[b,:] = sum_{k in [lod[b], lod[b+1])} W[ids[k],:]   (sparse_embedding + sequence_pool('sum'));
    padding_idx rows contribute zero; empty segment -> zeros (App. B-7).
    Returns (bow [B,D], counts [B] int64 = number of non-padding ids pooled per sample)."""
    B = len(lod) - 1
    out = np.zeros((B, W.shape[1]), dtype=W.dtype)
    cnt = np.zeros(B, dtype=np.int64)
    for b in range(B):
        acc = np.zeros(W.shape[1], dtype=W.dtype)
        for k in range(int(lod[b]), int(lod[b + 1])):
            i = int(ids[k])
            if padding_idx is not None and padding_idx >= 0 and i == padding_idx:
                continue
            acc = acc + W[i]
            cnt[b] += 1
        out[b] = acc
    return out, cnt


# --------------------------------------------------------------------------
# R — slot-text reader                       models/rank/deepfm/criteo_reader.py:61-103
# --------------------------------------------------------------------------
def parse_slot_line(line, n_sparse=26, n_dense=13, padding=0):
    """'click:L dense_feature:v x13 1:id ... 26:id' -> (label, ids[26] int64, dense[13] f32).
    A missing sparse slot is padded with one `padding` id, a missing dense slot with zeros."""
    label = []
    sparse = [[] for _ in range(n_sparse)]
    dense = []
    for tok in line.strip().split(" "):
        slot, _, val = tok.partition(":")
        if slot == "click":
            label.append(int(val))
        elif slot == "dense_feature":
            dense.append(float(val))
        elif slot.isdigit() and 1 <= int(slot) <= n_sparse:
            sparse[int(slot) - 1].append(int(val))
    if not label:
        label = [padding]
    if not dense:
        dense = [padding] * n_dense
    ids = np.array([s[0] if s else padding for s in sparse], dtype=np.int64)
    return np.int64(label[0]), ids, np.array(dense, dtype=np.float32)


# --------------------------------------------------------------------------
# R/P — multi-value slot reader               models/rank/slot_dnn/queuedataset_reader.py:56-82
# --------------------------------------------------------------------------
def parse_feasign_line(line, first_slot=1, num_slots=301, padding=0):
    """'feasign:slot feasign:slot ...' -> list over the slots first_slot .. first_slot+num_slots-1 of the slot's
    feasigns in token order (line_process, :56-82: `slots` = "1".."slot_num+1", :39-50); a slot that does not occur
    holds [padding]; tokens of other slots are dropped.  Feasigns are uint64 (returned as Python ints).  The
    constant show slot ("0", [1]) the reference prepends (:81) is not part of the output."""
    out = [[] for _ in range(num_slots)]
    for tok in line.strip().split(" "):
        if not tok:
            continue
        fs, _, slot = tok.partition(":")
        if not slot.isdigit() or not fs.isdigit():
            continue
        s = int(slot) - first_slot
        if 0 <= s < num_slots:
            out[s].append(int(fs))
    return [v if v else [padding] for v in out]


def feasign_row(f, hash_rows):
    """Row of a hashed table with `hash_rows` rows: 0 stays the padding row, any other feasign lands in [1, rows)
    (engine convention — the reference's PS table is an exact hash map keyed by the feasign [EXT])."""
    return 0 if f == 0 else 1 + f % (hash_rows - 1)
