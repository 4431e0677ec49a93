"""NumPy restatement of DCN-v2 (oracle — test infrastructure only).

Follows /root/reference/models/rank/dcn_v2/net.py line by line (row-vector form of its column-vector
batched matmuls); Paddle semantics from SURVEY.md Appendix B (Linear.weight is [in,out]; Embedding
padding_idx=0 gives a zero row and no gradient; Dropout is identity in eval mode — B-10: the reference
applies Dropout(0.5) after every MLP sublayer in train mode, so logits are only comparable in eval()).
Pinned against tests/golden/dcn_v2_{v2,mix}.npz (outputs and autograd gradients of the reference's
unmodified net.py over oracle/paddle_shim).
"""
import numpy as np

from .deepfm_ref import embedding_lookup, sigmoid

P = "DeepCrossLayer_.crossNet."


def _lin(x, w, b):
    return x @ w + b


# --------------------------------------------------------------------------
# feature embedding                                   dcn_v2/net.py:89-108
# --------------------------------------------------------------------------
def feat_embeddings(ids, dense, p):
    """concat(reshape(Embedding(ids), [-1, S*D]), Linear(dense))  -> [B, (S+Dn)*D]"""
    emb = embedding_lookup(p["embedding.weight"], ids, 0)                  # net.py:93-96
    B = ids.shape[0]
    sparse = emb.reshape(B, -1)                                            # net.py:100-102
    dense_emb = _lin(dense, p["dense_emb.weight"], p["dense_emb.bias"])    # net.py:104-105
    return np.concatenate([sparse, dense_emb], axis=1)                     # net.py:107-108


# --------------------------------------------------------------------------
# CrossNetV2                                          dcn_v2/net.py:214-226
# --------------------------------------------------------------------------
def cross_v2_forward(x0, p, num_layers):
    xs, us = [x0], []
    x = x0
    for i in range(num_layers):
        u = _lin(x, p[P + "cross_layers.%d.weight" % i], p[P + "cross_layers.%d.bias" % i])
        x = x + x0 * u                                                     # net.py:225
        us.append(u)
        xs.append(x)
    return x, (xs, us)


def cross_v2_backward(dout, x0, saved, p, num_layers):
    """Returns (dx0 total, grads dict)."""
    xs, us = saved
    g = {}
    dx = dout
    dx0 = np.zeros_like(x0)
    for i in reversed(range(num_layers)):
        w = p[P + "cross_layers.%d.weight" % i]
        du = dx * x0
        dx0 = dx0 + dx * us[i]
        g[P + "cross_layers.%d.weight" % i] = xs[i].T @ du
        g[P + "cross_layers.%d.bias" % i] = du.sum(axis=0)
        dx = dx + du @ w.T
    return dx0 + dx, g


# --------------------------------------------------------------------------
# CrossNetMix                                         dcn_v2/net.py:229-320
# --------------------------------------------------------------------------
def _softmax(z):
    z = z - z.max(axis=1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(axis=1, keepdims=True)


def cross_mix_forward(x0, p, layer_num, num_experts):
    x = x0
    saved = []
    for i in range(layer_num):
        U, V, C = p[P + "U_list.%d" % i], p[P + "V_list.%d" % i], p[P + "C_list.%d" % i]
        bias = p[P + "bias.%d" % i].reshape(1, -1)
        gate = np.concatenate([_lin(x, p[P + "gating.%d.weight" % e], p[P + "gating.%d.bias" % e])
                               for e in range(num_experts)], axis=1)                   # net.py:287-288
        t1 = [np.tanh(x @ V[e]) for e in range(num_experts)]                           # net.py:292-296
        t2 = [np.tanh(t1[e] @ C[e].T) for e in range(num_experts)]                     # net.py:297-298
        u = [t2[e] @ U[e].T + bias for e in range(num_experts)]                        # net.py:301-304
        prob = _softmax(gate)                                                          # net.py:315
        moe = sum(prob[:, e:e + 1] * (x0 * u[e]) for e in range(num_experts))          # net.py:305,314-316
        saved.append((x, t1, t2, u, prob))
        x = moe + x                                                                    # net.py:317
    return x, saved


def cross_mix_backward(dout, x0, saved, p, layer_num, num_experts):
    g = {}
    dx = dout
    dx0 = np.zeros_like(x0)
    for i in reversed(range(layer_num)):
        xl, t1, t2, u, prob = saved[i]
        U, V, C = p[P + "U_list.%d" % i], p[P + "V_list.%d" % i], p[P + "C_list.%d" % i]
        dU, dV, dC = np.zeros_like(U), np.zeros_like(V), np.zeros_like(C)
        dbias = np.zeros(x0.shape[1], dtype=x0.dtype)
        dp = np.stack([(dx * x0 * u[e]).sum(axis=1) for e in range(num_experts)], axis=1)   # [B,E]
        dgate = prob * (dp - (prob * dp).sum(axis=1, keepdims=True))
        dxl = dx.copy()
        for e in range(num_experts):
            do = dx * prob[:, e:e + 1]
            du = do * x0
            dx0 = dx0 + do * u[e]
            dbias += du.sum(axis=0)
            dU[e] = du.T @ t2[e]
            dc = (du @ U[e]) * (1 - t2[e] ** 2)
            dC[e] = dc.T @ t1[e]
            da = (dc @ C[e]) * (1 - t1[e] ** 2)
            dV[e] = xl.T @ da
            dxl = dxl + da @ V[e].T
            wg = p[P + "gating.%d.weight" % e]
            gk = P + "gating.%d." % e      # gating layers are shared by all cross layers (net.py:267-268)
            g[gk + "weight"] = g.get(gk + "weight", 0) + xl.T @ dgate[:, e:e + 1]
            g[gk + "bias"] = g.get(gk + "bias", 0) + dgate[:, e].sum(keepdims=True)
            dxl = dxl + dgate[:, e:e + 1] @ wg.T
        g[P + "U_list.%d" % i], g[P + "V_list.%d" % i], g[P + "C_list.%d" % i] = dU, dV, dC
        g[P + "bias.%d" % i] = dbias.reshape(-1, 1)
        dx = dxl
    return dx0 + dx, g


# --------------------------------------------------------------------------
# DNN + head                                          dcn_v2/net.py:110-137,140-184
# --------------------------------------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(k):
    k = k.astype(np.uint64)
    with np.errstate(over="ignore"):
        k = k ^ (k >> np.uint64(33))
        k = k * np.uint64(0xff51afd7ed558ccd)
        k = k ^ (k >> np.uint64(33))
        k = k * np.uint64(0xc4ceb9fe1a85ec53)
        k = k ^ (k >> np.uint64(33))
    return k


def dropout_keep(shape, p, seed, stream):
    """The engine's counter-based keep mask (csrc/cross_ops.hip dropout_kernel): element e of a [rows, cols] matrix is
    kept when bits 32.. of mix64(key + e) >= p * 2^32 with the per-stream key = mix64(seed ^ mix64(stream + golden
    ratio)), all mod 2^64 — any stream id.  (Paddle's own masks come from the device generator and cannot be reproduced
    from the reference; the distribution is the same Bernoulli(1 - p).)"""
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        key = _mix64(np.array([np.uint64(seed) ^ _mix64(np.array([np.uint64(stream) + np.uint64(0x9E3779B97F4A7C15)]))[0]]))[0]
        e = np.arange(n, dtype=np.uint64) + key
    h = _mix64(e)
    return ((h >> np.uint64(32)) >= np.uint64(int(float(p) * 4294967296.0))).reshape(shape)


def dnn_forward(x, p, n_layers, dropout=None):
    """DNNLayer.forward (net.py:178-184).  dropout = None: eval mode.  dropout = (p, seed, step): train mode —
    `y = drop_out(layer(y))` after the Linear AND after its ReLU (upscale_in_train [EXT]); mask stream of (layer i,
    position j) = (step * n_layers + i) * 2 + j."""
    acts = [x]
    for i in range(n_layers):
        z = _lin(x, p["DNN_.linear_%d.weight" % i], p["DNN_.linear_%d.bias" % i])
        if dropout is not None:
            pr, seed, step = dropout
            s = np.float32(1.0) / (np.float32(1.0) - np.float32(pr))
            z = z * dropout_keep(z.shape, pr, seed, (step * n_layers + i) * 2) * s          # after the Linear
            x = np.maximum(z, 0)
            x = x * dropout_keep(x.shape, pr, seed, (step * n_layers + i) * 2 + 1) * s      # after the ReLU
        else:
            x = np.maximum(z, 0)
        acts.append(x)
    return x, acts


def dnn_backward(dy, acts, p, n_layers, dropout=None):
    g = {}
    d = dy
    for i in reversed(range(n_layers)):
        if dropout is not None:       # through both dropouts and the ReLU between them
            pr, seed, step = dropout
            s = np.float32(1.0) / (np.float32(1.0) - np.float32(pr))
            k = dropout_keep(d.shape, pr, seed, (step * n_layers + i) * 2) & \
                dropout_keep(d.shape, pr, seed, (step * n_layers + i) * 2 + 1)
            d = d * k * (s * s)
        d = d * (acts[i + 1] > 0)
        g["DNN_.linear_%d.weight" % i] = acts[i].T @ d
        g["DNN_.linear_%d.bias" % i] = d.sum(axis=0)
        d = d @ p["DNN_.linear_%d.weight" % i].T
    return d, g


def config_of(p):
    mix = (P + "U_list.0") in p
    n_cross = len([k for k in p if k.startswith(P + ("U_list." if mix else "cross_layers.")) and
                   (mix or k.endswith("weight"))])
    n_dnn = len([k for k in p if k.startswith("DNN_.linear_") and k.endswith("weight")])
    n_exp = p[P + "U_list.0"].shape[0] if mix else 0
    d = p["DNN_.linear_0.weight"].shape[0]
    stacked = p["fc.weight"].shape[0] == p["DNN_.linear_%d.weight" % (n_dnn - 1)].shape[1]
    return dict(mix=mix, n_cross=n_cross, n_dnn=n_dnn, n_exp=n_exp, d=d, stacked=stacked)


def forward(ids, dense, p, return_saved=False, dropout=None):
    """DCN_V2Layer.forward (net.py:89-137) -> predict [B,1]; eval mode, or train mode with dropout = (p, seed, step)."""
    c = config_of(p)
    feat = feat_embeddings(ids, dense, p)
    if c["mix"]:
        cross, csaved = cross_mix_forward(feat, p, c["n_cross"], c["n_exp"])
    else:
        cross, csaved = cross_v2_forward(feat, p, c["n_cross"])
    dnn_in = cross if c["stacked"] else feat
    dnn_out, acts = dnn_forward(dnn_in, p, c["n_dnn"], dropout)
    last = dnn_out if c["stacked"] else np.concatenate([dnn_out, cross], axis=1)   # net.py:129
    logit = _lin(last, p["fc.weight"], p["fc.bias"])
    pred = sigmoid(logit)
    if return_saved:
        return pred, dict(feat=feat, cross=cross, csaved=csaved, acts=acts, last=last, logit=logit, cfg=c,
                          dropout=dropout)
    return pred


def backward(ids, dense, p, saved, dpred):
    """Gradients of all parameters given d loss / d predict ([B,1]).  The embedding gradient is returned
    dense ([N,D], padding row untouched) for comparison with autograd."""
    c = saved["cfg"]
    pred = sigmoid(saved["logit"])
    dlogit = dpred * pred * (1 - pred)
    g = {"fc.weight": saved["last"].T @ dlogit, "fc.bias": dlogit.sum(axis=0)}
    dlast = dlogit @ p["fc.weight"].T
    n_out = saved["acts"][-1].shape[1]
    ddnn = dlast[:, :n_out]
    dcross = None if c["stacked"] else dlast[:, n_out:]
    d_in, gd = dnn_backward(ddnn, saved["acts"], p, c["n_dnn"], saved.get("dropout"))
    g.update(gd)
    if c["stacked"]:
        dcross, dfeat = d_in, 0
    else:
        dfeat = d_in
    if c["mix"]:
        dx0, gc = cross_mix_backward(dcross, saved["feat"], saved["csaved"], p, c["n_cross"], c["n_exp"])
    else:
        dx0, gc = cross_v2_backward(dcross, saved["feat"], saved["csaved"], p, c["n_cross"])
    g.update(gc)
    dfeat = dfeat + dx0
    B, S = ids.shape
    D = p["embedding.weight"].shape[1]
    g["dense_emb.weight"] = dense.T @ dfeat[:, S * D:]
    g["dense_emb.bias"] = dfeat[:, S * D:].sum(axis=0)
    gW = np.zeros_like(p["embedding.weight"])
    rows = ids.reshape(-1)
    vals = dfeat[:, :S * D].reshape(B * S, D)
    np.add.at(gW, rows[rows != 0], vals[rows != 0])
    g["embedding.weight"] = gW
    g["_row_grad"] = vals
    return g
