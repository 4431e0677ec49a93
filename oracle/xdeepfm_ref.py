"""NumPy restatement of rank/xdeepfm (oracle — test infrastructure only).

Follows /root/reference/models/rank/xdeepfm/net.py:
  Linear (:58-124)  == DeepFM's first-order term + feat_embeddings (deepfm_ref.fm_forward; its y2 is unused), built
                    WITHOUT padding_idx (:75-93) and with Constant(1.0) dense weights (:96-104)
  CIN (:127-202)    per layer  Z[b,d,f,s] = X0[b,f,d] Xk[b,s,d] (:163-175),  X_{k+1}[b,c,d] = sum_{f,s} Wc[c, f*S+s] Z
                    (1x1 Conv2D without bias, :176-190), pooled = concat_k sum_d X_k (:195-198), y_cin = cnn_fc(pooled)
  DNN (:205-242)    Linear+ReLU ... Linear(1) on feat.reshape(B, F*D)
  predict (:54)     sigmoid(y_linear + bias + y_cin + y_dnn)
Loss: xdeepfm/dygraph_model.py:53-58 (log_loss, mean).  L2Decay(1e-4) on the CIN / cnn_fc / DNN weights (:139,150,219):
Paddle adds coeff * w to those gradients in the optimizer [EXT] — `l2_decay_grads` below, NOT part of autograd's
gradients (the golden holds autograd's).  Pinned against tests/golden/xdeepfm_D9.npz (forward outputs and autograd
gradients of the reference's unmodified net.py over oracle/paddle_shim).
"""
import numpy as np

from . import deepfm_ref as R

NO_PADDING = None
L2_COEFF = 1e-4


def cin_forward(feat, cin_w):
    """feat [B,F,D]; cin_w[k] [C_k, F*S_k] (the Conv2D weight [C, F*S, 1, 1] viewed 2-D).
    -> (pooled [B, sum C], [X_1..X_L] with X_k [B, C_k, D])"""
    B, F, D = feat.shape
    xs, xk = [], feat
    for w in cin_w:
        S = xk.shape[1]
        w3 = w.reshape(w.shape[0], F, S)                       # channel index = f*S + s   (net.py:170-175)
        xk = np.einsum("cfs,bfd,bsd->bcd", w3, feat, xk)
        xs.append(xk)
    pooled = np.concatenate([x.sum(axis=2, dtype=x.dtype) for x in xs], axis=1)
    return pooled, xs


def cin_backward(feat, cin_w, xs, dpooled):
    """-> (d feat [B,F,D] from the CIN, [dWc_k])"""
    B, F, D = feat.shape
    L = len(cin_w)
    offs = np.cumsum([0] + [w.shape[0] for w in cin_w])
    dfeat = np.zeros_like(feat)
    dws = [None] * L
    dx_next = None                                             # gradient w.r.t. X_{i+1} from layer i+1's use of it
    for i in reversed(range(L)):
        C = cin_w[i].shape[0]
        dxo = np.repeat(dpooled[:, offs[i]:offs[i] + C, None], D, axis=2)       # sum over d, broadcast back
        if dx_next is not None:
            dxo = dxo + dx_next
        xk = feat if i == 0 else xs[i - 1]
        S = xk.shape[1]
        w3 = cin_w[i].reshape(C, F, S)
        dws[i] = np.einsum("bcd,bfd,bsd->cfs", dxo, feat, xk).reshape(C, F * S)
        dfeat += np.einsum("bcd,cfs,bsd->bfd", dxo, w3, xk)
        dxk = np.einsum("bcd,cfs,bfd->bsd", dxo, w3, feat)
        if i == 0:
            dfeat += dxk
        else:
            dx_next = dxk
    return dfeat, dws


def forward(ids, dense, p):
    """p: W, W1, dense_w, dense_w_one, bias, cin_w [list], fc_w [sumC,1], fc_b [1], mlp_w, mlp_b.
    -> (pred, z, cache)"""
    y1, _, feat = R.fm_forward(ids, dense, p["W1"], p["W"], p["dense_w_one"], p["dense_w"], NO_PADDING, None)
    pooled, xs = cin_forward(feat, p["cin_w"])
    y_cin = pooled @ p["fc_w"] + p["fc_b"].reshape(1, 1)
    y_dnn, acts = R.dnn_forward(feat, p["mlp_w"], p["mlp_b"], return_acts=True)
    z = y1 + p["bias"].reshape(1, 1) + y_cin + y_dnn
    return R.sigmoid(z), z, dict(y1=y1, feat=feat, pooled=pooled, xs=xs, y_cin=y_cin, y_dnn=y_dnn, acts=acts)


def loss_and_grads(ids, dense, label, p):
    """One train_forward + backward (xdeepfm/dygraph_model.py:77-90, tools/trainer.py:148-151): autograd's gradients."""
    pred, z, c = forward(ids, dense, p)
    loss = R.log_loss_mean(pred, label)
    dz = R.log_loss_mean_grad_z(pred, label)
    d_fc_w = c["pooled"].T @ dz
    d_b = dz.sum(axis=0, dtype=dz.dtype).reshape(1)
    dpooled = dz @ p["fc_w"].T
    dflat, dws, dbs = R.dnn_backward(dz, c["acts"], p["mlp_w"])
    dfeat_cin, d_cin_w = cin_backward(c["feat"], p["cin_w"], c["xs"], dpooled)
    g = R.fm_backward(ids, dense, c["feat"], dflat.reshape(c["feat"].shape) + dfeat_cin, dz, np.zeros_like(dz),
                      NO_PADDING, None)
    g.update(mlp_dw=dws, mlp_db=dbs, d_cin_w=d_cin_w, d_fc_w=d_fc_w, d_fc_b=d_b, d_bias=d_b.copy(), loss=loss,
             pred=pred, dz=dz, feat=c["feat"], y1=c["y1"], y_cin=c["y_cin"], y_dnn=c["y_dnn"], pooled=c["pooled"])
    return g


def l2_decay_grads(g, p, coeff=L2_COEFF):
    """The gradients the optimizer sees: + coeff * w on the regularised weights (net.py:139,150,219) [EXT L2Decay]."""
    out = dict(g)
    out["d_cin_w"] = [d + coeff * w for d, w in zip(g["d_cin_w"], p["cin_w"])]
    out["d_fc_w"] = g["d_fc_w"] + coeff * p["fc_w"]
    out["mlp_dw"] = [d + coeff * w for d, w in zip(g["mlp_dw"], p["mlp_w"])]
    return out
