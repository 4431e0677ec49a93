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
