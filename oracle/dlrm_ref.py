"""NumPy restatement of rank/dlrm (oracle — test infrastructure only).

Follows /root/reference/models/rank/dlrm/net.py:
  MLPLayer (:128-178)   every layer is Linear -> ReLU -> BatchNorm1D: `i != len(units_list) - 1` (:145) is true for
                        every i of `enumerate(units_list[:-1])`, so the `else` branch (:158-176) is dead code and the
                        LAST layer — the 2 class scores of top_mlp included — is ReLU'd and batch-normalised too
  DLRMLayer.forward (:82-125)  x = bot_mlp(dense) [B,D]; T = [emb(s_1) .. emb(s_26), x] [B,27,D]; Z = bmm(T, T^T);
                        Zflat = strictly upper triangle of Z in row-major order (triu(Z,1) + MIN_FLOAT below / on the
                        diagonal, masked_select > MIN_FLOAT, self_interaction=False); R = concat(x, Zflat);
                        y = top_mlp(R) [B,2]
  loss (dygraph_model.py:53-57)  mean softmax cross-entropy of the two raw scores
  Embedding (:70-77)    ONE shared table, no padding_idx, TruncatedNormal() [EXT: mean 0, std 1, cut at +-2 std]
BatchNorm1D [EXT Paddle batch_norm]: momentum 0.9, epsilon 1e-5; training normalises with the batch mean and the biased
batch variance and moves the running statistics by (1 - momentum) towards them (biased variance there too).
Pinned against tests/golden/dlrm_D16.npz (forward outputs and autograd gradients of the reference's unmodified net.py
over oracle/paddle_shim, whose BatchNorm1D states the same [EXT] semantics).
"""
import numpy as np

BN_MOMENTUM, BN_EPS = 0.9, 1e-5


def batchnorm_forward(x, gamma, beta, running_mean, running_var, training=True, momentum=BN_MOMENTUM, eps=BN_EPS):
    """-> (y, mean used, invstd used); running_mean / running_var are updated IN PLACE when training."""
    dt = x.dtype
    if training:
        mean = x.mean(axis=0, dtype=dt)
        var = ((x - mean) ** 2).mean(axis=0, dtype=dt)              # biased
        running_mean[...] = dt.type(momentum) * running_mean + dt.type(1 - momentum) * mean
        running_var[...] = dt.type(momentum) * running_var + dt.type(1 - momentum) * var
    else:
        mean, var = running_mean.astype(dt), running_var.astype(dt)
    invstd = (1.0 / np.sqrt(var + dt.type(eps))).astype(dt)
    return (x - mean) * invstd * gamma + beta, mean, invstd


def batchnorm_backward(x, dy, gamma, mean, invstd):
    m = x.shape[0]
    xhat = (x - mean) * invstd
    dbeta = dy.sum(axis=0, dtype=x.dtype)
    dgamma = (dy * xhat).sum(axis=0, dtype=x.dtype)
    dx = gamma * invstd * (dy - dbeta / m - xhat * dgamma / m)
    return dx.astype(x.dtype), dgamma, dbeta


def mlp_forward(x, layers, training=True):
    """layers: list of dict(w, b, gamma, beta, mean, var).  -> (y, cache)"""
    cache = []
    for L in layers:
        h = np.maximum(x @ L["w"] + L["b"], 0)
        y, mu, invstd = batchnorm_forward(h, L["gamma"], L["beta"], L["mean"], L["var"], training)
        cache.append((x, h, mu, invstd))
        x = y
    return x, cache


def mlp_backward(dy, layers, cache):
    """-> (dx, [dict(dw, db, dgamma, dbeta)])"""
    grads = [None] * len(layers)
    for i in reversed(range(len(layers))):
        L = layers[i]
        x, h, mu, invstd = cache[i]
        dh, dg, dbt = batchnorm_backward(h, dy, L["gamma"], mu, invstd)
        dpre = dh * (h > 0)
        grads[i] = dict(dw=x.T @ dpre, db=dpre.sum(axis=0, dtype=x.dtype), dgamma=dg, dbeta=dbt)
        dy = dpre @ L["w"].T
    return dy, grads


def _pairs(F):
    return [(i, j) for i in range(F) for j in range(i + 1, F)]      # row-major over the strict upper triangle


def dot_interact(T):
    """T [B,F,D] -> R [B, D + F(F-1)/2] = [T[:, F-1] | <T_i, T_j>, i < j]     (net.py:96-123)"""
    B, F, D = T.shape
    Z = np.einsum("bid,bjd->bij", T, T)
    iu = np.array(_pairs(F))
    return np.concatenate([T[:, F - 1, :], Z[:, iu[:, 0], iu[:, 1]]], axis=1).astype(T.dtype)


def dot_interact_backward(T, dR):
    B, F, D = T.shape
    iu = np.array(_pairs(F))
    dZ = np.zeros((B, F, F), T.dtype)
    dZ[:, iu[:, 0], iu[:, 1]] = dR[:, D:]
    dZ = dZ + dZ.transpose(0, 2, 1)
    dT = np.einsum("bij,bjd->bid", dZ, T).astype(T.dtype)
    dT[:, F - 1, :] += dR[:, :D]
    return dT


def forward(ids, dense, p, training=True):
    """p: W [N,D], bot [layers], top [layers].  -> (raw [B,2], cache)"""
    x, cb = mlp_forward(dense, p["bot"], training)
    T = np.concatenate([p["W"][ids], x[:, None, :]], axis=1)       # [B, S+1, D]
    R = dot_interact(T)
    raw, ct = mlp_forward(R, p["top"], training)
    return raw, dict(x=x, cb=cb, T=T, R=R, ct=ct)


def softmax_ce_mean(raw, label):
    z = raw - raw.max(axis=1, keepdims=True)
    lse = np.log(np.exp(z).sum(axis=1, keepdims=True))
    t = label.reshape(-1)
    loss = -(z[np.arange(len(t)), t] - lse[:, 0]).mean(dtype=raw.dtype)
    prob = np.exp(z - lse)
    d = prob.copy()
    d[np.arange(len(t)), t] -= 1
    return raw.dtype.type(loss), prob, (d / len(t)).astype(raw.dtype)


def loss_and_grads(ids, dense, label, p, training=True):
    """One train_forward + backward (dlrm/dygraph_model.py:74-91, tools/trainer.py:148-151)."""
    raw, c = forward(ids, dense, p, training)
    loss, prob, draw = softmax_ce_mean(raw, label)
    dR, gt = mlp_backward(draw, p["top"], c["ct"])
    dT = dot_interact_backward(c["T"], dR)
    S = ids.shape[1]
    _, gb = mlp_backward(dT[:, S, :], p["bot"], c["cb"])
    return dict(loss=loss, raw=raw, pred=prob[:, 1:2], rows=ids.reshape(-1), row_grad=dT[:, :S, :].reshape(-1, dT.shape[2]),
                top=gt, bot=gb, x=c["x"], R=c["R"])
