"""NumPy restatement of DIN (oracle — test infrastructure only).

Follows /root/reference/models/rank/din/net.py:139-184 line by line; the mask is what
din/dinReader.py:81-84,99 produces (0 for valid positions, -1e9 for padding, cast to int64 and back to
float32 by net.py:164).  Paddle semantics from SURVEY.md App. B (B-9: the attention Linear layers are not
registered parameters in dygraph mode; B-11: padded positions gather row 0 and vanish through the
softmax).  Pinned against tests/golden/din.npz (logit + loss of the reference's unmodified net.py).
"""
import numpy as np


def sigmoid(z):
    return 1.0 / (1.0 + np.exp(-z))


def attention_pool(h, q, mask, att_w, att_b, return_weights=False):
    """net.py:149-171.  h, q [B,T,E]; mask [B,T] (0 / -1e9); att_w/att_b: 3 Linear layers
    (4E->80 sigmoid, 80->40 sigmoid, 40->1).  Returns out [B,E] (= softmax(...) @ h)."""
    E = h.shape[2]
    x = np.concatenate([h, q, h - q, h * q], axis=2)                     # net.py:155-161
    a = sigmoid(x @ att_w[0] + att_b[0])                                 # net.py:163-164 (attention_layer)
    a = sigmoid(a @ att_w[1] + att_b[1])
    s = a @ att_w[2] + att_b[2]                                          # [B,T,1]
    s = s + mask[..., None].astype(s.dtype)                              # net.py:166
    s = np.transpose(s, (0, 2, 1)) * np.asarray(E ** -0.5, dtype=s.dtype)   # net.py:167-168
    s = s - s.max(axis=2, keepdims=True)
    w = np.exp(s)
    w = w / w.sum(axis=2, keepdims=True)                                 # net.py:169
    out = (w @ h).reshape(h.shape[0], E)                                 # net.py:171-173
    return (out, w[:, 0, :]) if return_weights else out


def forward(p, att, hist_item, hist_cat, target_item, target_cat, mask, target_item_seq=None,
            target_cat_seq=None):
    """DINLayer.forward (net.py:139-184) -> logit [B,1].  p: registered parameters by name;
    att: (weights, biases) of the attention MLP (not in state_dict, App. B-9)."""
    B, T = hist_item.shape
    if target_item_seq is None:
        target_item_seq = np.repeat(target_item[:, None], T, 1)          # dinReader.py: target repeated T times
        target_cat_seq = np.repeat(target_cat[:, None], T, 1)
    h = np.concatenate([p["hist_item_emb_attr.weight"][hist_item],
                        p["hist_cat_emb_attr.weight"][hist_cat]], axis=2)           # net.py:141-142,149
    q = np.concatenate([p["target_item_seq_emb_attr.weight"][target_item_seq],
                        p["target_cat_seq_emb_attr.weight"][target_cat_seq]], axis=2)   # net.py:145-146,150-151
    tc = np.concatenate([p["target_item_emb_attr.weight"][target_item],
                         p["target_cat_emb_attr.weight"][target_cat]], axis=1)      # net.py:143-144,152-153
    item_b = p["item_b_attr.weight"][target_item]                                    # net.py:147
    pooled = attention_pool(h, q, mask.reshape(B, T), att[0], att[1])
    c = pooled @ p["linearCon.weight"] + p["linearCon.bias"]                         # net.py:175-176
    e = np.concatenate([c, tc], axis=1)                                              # net.py:178
    e = sigmoid(e @ p["linear_0.weight"] + p["linear_0.bias"])                       # net.py:180-181 (con_layer[1:])
    e = sigmoid(e @ p["linear_1.weight"] + p["linear_1.bias"])
    e = e @ p["linear_2.weight"] + p["linear_2.bias"]
    return e + item_b                                                                # net.py:183


def bce_with_logits_mean(logit, label):
    """paddle.nn.functional.binary_cross_entropy_with_logits(reduction='mean') (din/dygraph_model.py:58-61)."""
    z, t = logit.astype(np.float64), label.astype(np.float64)
    return np.mean(np.maximum(z, 0) - z * t + np.log1p(np.exp(-np.abs(z))))


# --------------------------------------------------------------------------
# backward (what loss.backward() computes for din/net.py:139-184)
# --------------------------------------------------------------------------
def attention_pool_backward(h, q, mask, att_w, att_b, dout):
    """Gradients of attention_pool w.r.t. h and q (and the attention MLP parameters).
    Returns dict(dh [B,T,E], dq [B,T,E], dW=[...], db=[...])."""
    B, T, E = h.shape
    x = np.concatenate([h, q, h - q, h * q], axis=2)
    a1 = sigmoid(x @ att_w[0] + att_b[0])
    a2 = sigmoid(a1 @ att_w[1] + att_b[1])
    c = np.asarray(E ** -0.5, dtype=h.dtype)
    s = ((a2 @ att_w[2] + att_b[2])[..., 0] + mask.astype(h.dtype)) * c
    s = s - s.max(axis=1, keepdims=True)
    p = np.exp(s)
    p = p / p.sum(axis=1, keepdims=True)                    # [B,T]
    dp = (h * dout[:, None, :]).sum(axis=2)                 # [B,T]
    dh = p[..., None] * dout[:, None, :]
    ds = p * (dp - (p * dp).sum(axis=1, keepdims=True))
    dl = (ds * c)[..., None]                                # [B,T,1]
    dz2 = (dl @ att_w[2].T) * a2 * (1 - a2)
    dz1 = (dz2 @ att_w[1].T) * a1 * (1 - a1)
    dx = dz1 @ att_w[0].T                                   # [B,T,4E]
    dh = dh + dx[..., :E] + dx[..., 2 * E:3 * E] + dx[..., 3 * E:] * q
    dq = dx[..., E:2 * E] - dx[..., 2 * E:3 * E] + dx[..., 3 * E:] * h
    dW = [np.einsum("btk,btj->kj", x, dz1), np.einsum("btk,btj->kj", a1, dz2), np.einsum("btk,btj->kj", a2, dl)]
    db = [dz1.sum((0, 1)), dz2.sum((0, 1)), dl.sum((0, 1))]
    return dict(dh=dh, dq=dq, dW=dW, db=db, p=p)


def backward(p, att, hist_item, hist_cat, target_item, target_cat, mask, label):
    """Gradients of mean BCE-with-logits (din/dygraph_model.py:58-61) w.r.t. the REGISTERED parameters
    (embedding gradients returned dense, as Paddle does for is_sparse=False)."""
    B, T = hist_item.shape
    tis = np.repeat(target_item[:, None], T, 1)
    tcs = np.repeat(target_cat[:, None], T, 1)
    h = np.concatenate([p["hist_item_emb_attr.weight"][hist_item], p["hist_cat_emb_attr.weight"][hist_cat]], 2)
    q = np.concatenate([p["target_item_seq_emb_attr.weight"][tis], p["target_cat_seq_emb_attr.weight"][tcs]], 2)
    tc = np.concatenate([p["target_item_emb_attr.weight"][target_item], p["target_cat_emb_attr.weight"][target_cat]], 1)
    m2 = mask.reshape(B, T)
    pooled = attention_pool(h, q, m2, att[0], att[1])
    c = pooled @ p["linearCon.weight"] + p["linearCon.bias"]
    e0 = np.concatenate([c, tc], axis=1)
    e1 = sigmoid(e0 @ p["linear_0.weight"] + p["linear_0.bias"])
    e2 = sigmoid(e1 @ p["linear_1.weight"] + p["linear_1.bias"])
    logit = e2 @ p["linear_2.weight"] + p["linear_2.bias"] + p["item_b_attr.weight"][target_item]
    g = {}
    dlogit = (sigmoid(logit) - label) / np.asarray(B, dtype=logit.dtype)
    g["linear_2.weight"], g["linear_2.bias"] = e2.T @ dlogit, dlogit.sum(0)
    d2 = (dlogit @ p["linear_2.weight"].T) * e2 * (1 - e2)
    g["linear_1.weight"], g["linear_1.bias"] = e1.T @ d2, d2.sum(0)
    d1 = (d2 @ p["linear_1.weight"].T) * e1 * (1 - e1)
    g["linear_0.weight"], g["linear_0.bias"] = e0.T @ d1, d1.sum(0)
    de0 = d1 @ p["linear_0.weight"].T
    E = h.shape[2]
    dc, dtc = de0[:, :E], de0[:, E:]
    g["linearCon.weight"], g["linearCon.bias"] = pooled.T @ dc, dc.sum(0)
    dpooled = dc @ p["linearCon.weight"].T
    ab = attention_pool_backward(h, q, m2, att[0], att[1], dpooled)
    Ei = p["hist_item_emb_attr.weight"].shape[1]

    def scatter(name, ids, vals):
        out = np.zeros_like(p[name])
        np.add.at(out, ids.reshape(-1), vals.reshape(-1, vals.shape[-1]))
        g[name] = out

    scatter("hist_item_emb_attr.weight", hist_item, ab["dh"][..., :Ei])
    scatter("hist_cat_emb_attr.weight", hist_cat, ab["dh"][..., Ei:])
    scatter("target_item_seq_emb_attr.weight", tis, ab["dq"][..., :Ei])
    scatter("target_cat_seq_emb_attr.weight", tcs, ab["dq"][..., Ei:])
    scatter("target_item_emb_attr.weight", target_item, dtc[:, :Ei])
    scatter("target_cat_emb_attr.weight", target_cat, dtc[:, Ei:])
    scatter("item_b_attr.weight", target_item, dlogit)
    g["_att"] = ab
    g["_logit"] = logit
    return g


# --------------------------------------------------------------------------
# reader                                              models/rank/din/dinReader.py:46-147
# --------------------------------------------------------------------------
def reader_batches(lines, batch_size):
    """dinReader.py restated as plain loops: groups of 20*batch_size lines are sorted by history length
    (stable), cut into batches, padded with id 0 to the batch's longest history; mask = 0 / -1e9 (cast to
    int64 as dinReader.py:99 does); the last group drops its incomplete batch.
    Yields dicts of arrays for whole batches (the per-sample `yield res` + DataLoader collate of the
    reference regroup exactly these)."""
    res0 = []
    for line in lines:
        line = line.strip().split(";")
        if len(line) < 5:
            continue
        res0.append([line[0].split(), line[1].split(), line[2], line[3], float(line[4])])
    group_size = batch_size * 20
    out = []

    def emit(group, upto):
        sortb = sorted(group, key=lambda x: len(x[0]))
        for i in range(0, upto, batch_size):
            b = sortb[i:i + batch_size]
            max_len = max(len(x[0]) for x in b)
            item = np.array([x[0] + [0] * (max_len - len(x[0])) for x in b]).astype("int64").reshape([-1, max_len])
            cat = np.array([x[1] + [0] * (max_len - len(x[1])) for x in b]).astype("int64").reshape([-1, max_len])
            mask = np.array([[0] * len(x[0]) + [-1e9] * (max_len - len(x[0])) for x in b]).reshape([-1, max_len, 1])
            out.append(dict(
                hist_item=item, hist_cat=cat,
                target_item=np.array([x[2] for x in b]).astype("int64"),
                target_cat=np.array([x[3] for x in b]).astype("int64"),
                label=np.array([x[4] for x in b]).astype("float32").reshape(-1, 1),
                mask=mask.astype("int64"),
                target_item_seq=np.array([[x[2]] * max_len for x in b]).astype("int64").reshape([-1, max_len]),
                target_cat_seq=np.array([[x[3]] * max_len for x in b]).astype("int64").reshape([-1, max_len])))

    bg = []
    for rec in res0:
        bg.append(rec)
        if len(bg) == group_size:
            emit(bg, group_size)
            bg = []
    if bg:
        emit(bg, len(bg) - len(bg) % batch_size)
    return out
