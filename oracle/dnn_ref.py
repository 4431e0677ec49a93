"""NumPy restatement of rank/dnn (oracle — test infrastructure only).

Follows /root/reference/models/rank/dnn/net.py `DNNLayer` (non-gpubox branch, :38-95): the same lookup + concat + MLP
as wide_deep (wide_deep_ref.features) but the last Linear has TWO outputs and there is no wide part; the loss is
softmax cross-entropy over them (dnn/dygraph_model.py:53-58: paddle.nn.functional.cross_entropy, hard int64 labels,
mean), the AUC is taken on softmax(raw)[:,1] (:78-80).
Pinned against tests/golden/dnn_D9.npz (the reference's unmodified net.py over oracle/paddle_shim; the loss line is
torch's cross_entropy, the same definition [EXT]).
Identity the engine uses: for two classes  CE(z, t) = BCE_with_logits(z1 - z0, t)  and softmax(z)[:,1] = sigmoid(z1 - z0).
"""
import numpy as np

from . import deepfm_ref as R
from .wide_deep_ref import features


def forward(ids, dense, p):
    """-> raw [B,2] logits (DNNLayer.forward returns them unnormalised), (x, acts)."""
    x = features(ids, dense, p["W"])
    raw, acts = R.dnn_forward(x, p["mlp_w"], p["mlp_b"], return_acts=True)
    return raw, (x, acts)


def softmax(z):
    e = np.exp(z - z.max(axis=1, keepdims=True))
    return e / e.sum(axis=1, keepdims=True)


def loss_and_grads(ids, dense, label, p):
    """One train_forward + backward (dnn/dygraph_model.py:74-86, tools/trainer.py:148-151)."""
    raw, (x, acts) = forward(ids, dense, p)
    B = raw.shape[0]
    t = label.reshape(-1).astype(np.int64)
    z = raw.astype(np.float64)
    lse = np.log(np.exp(z - z.max(axis=1, keepdims=True)).sum(axis=1)) + z.max(axis=1)
    loss = np.float32((lse - z[np.arange(B), t]).mean())
    prob = softmax(raw)
    draw = prob.copy()
    draw[np.arange(B), t] -= 1
    draw = (draw / np.asarray(B, raw.dtype)).astype(raw.dtype)
    dx, dws, dbs = R.dnn_backward(draw, acts, p["mlp_w"])
    S, D = ids.shape[1], p["W"].shape[1]
    rows, valid = R.effective_rows(ids, None)
    return dict(loss=loss, raw=raw, pred=prob[:, 1:2], mlp_dw=dws, mlp_db=dbs, rows=rows.reshape(-1),
                row_valid=valid.reshape(-1), row_grad=np.ascontiguousarray(dx[:, :S * D]).reshape(B * S, D))
