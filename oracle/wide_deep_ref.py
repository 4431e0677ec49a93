"""NumPy restatement of rank/wide_deep (oracle — test infrastructure only).

Follows /root/reference/models/rank/wide_deep/net.py (non-gpubox branch):
    wide  = Linear(dense)                                                     net.py:76
    deep  = MLP(concat([Embedding(s_i).reshape(-1,D) for i in slots] + [dense], axis=1))   net.py:78-99
    pred  = sigmoid(wide + deep)                                              net.py:101-103
The Embedding has no padding_idx (net.py:48-54: id 0 is an ordinary row) and is Uniform(-1,1)-initialised [EXT];
Linear.weight is [in,out] (App. B-2).  Loss: wide_deep/dygraph_model.py:54-59 (log_loss, mean).
Pinned against tests/golden/wide_deep_D9.npz (the reference's unmodified net.py over oracle/paddle_shim).
"""
import numpy as np

from . import deepfm_ref as R


def features(ids, dense, W):
    """concat(rows of all slots, dense) -> [B, S*D + Dn]                      net.py:78-96"""
    B = ids.shape[0]
    emb = R.embedding_lookup(W, ids, None)
    return np.concatenate([emb.reshape(B, -1), dense.astype(W.dtype)], axis=1)


def forward(ids, dense, p):
    x = features(ids, dense, p["W"])
    wide = dense.astype(p["wide_w"].dtype) @ p["wide_w"] + p["wide_b"]
    deep, acts = R.dnn_forward(x, p["mlp_w"], p["mlp_b"], return_acts=True)
    z = wide + deep
    return R.sigmoid(z), z, (x, acts, wide, deep)


def loss_and_grads(ids, dense, label, p):
    """One train_forward + backward (wide_deep/dygraph_model.py:73-85, tools/trainer.py:148-151)."""
    pred, z, (x, acts, wide, deep) = forward(ids, dense, p)
    loss = R.log_loss_mean(pred, label)
    dz = R.log_loss_mean_grad_z(pred, label)
    dx, dws, dbs = R.dnn_backward(dz, acts, p["mlp_w"])
    B, S = ids.shape
    D = p["W"].shape[1]
    rows, valid = R.effective_rows(ids, None)
    return dict(loss=loss, pred=pred, dz=dz, x=x, mlp_dw=dws, mlp_db=dbs,
                d_wide_w=dense.astype(dz.dtype).T @ dz, d_wide_b=dz.sum(axis=0),
                rows=rows.reshape(-1), row_valid=valid.reshape(-1),
                row_grad=np.ascontiguousarray(dx[:, :S * D]).reshape(B * S, D))
