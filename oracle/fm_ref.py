"""NumPy restatement of rank/fm (oracle — test infrastructure only).

Follows /root/reference/models/rank/fm/net.py: the FM block (net.py:41-124) is line for line the one of
deepfm/net.py:52-139 (restated in deepfm_ref.fm_forward / fm_backward); FMLayer.forward (net.py:35-38) adds the
trainable scalar `bias` — which, unlike DeepFM's (App. B-14), IS used — and there is no DNN tower.  Two more
differences from DeepFM: the Embeddings are built WITHOUT padding_idx (net.py:55-73), so id 0 is an ordinary row
that is looked up and trained, and the dense weights are Constant(1.0)-initialised (net.py:78-88).  Loss: fm/dygraph_model.py:53-58 (log_loss, mean).
Pinned against tests/golden/fm_D9.npz (forward outputs and autograd gradients of the reference's unmodified
net.py over oracle/paddle_shim).
"""
import numpy as np

from . import deepfm_ref as R

NO_PADDING = None      # fm/net.py:55-73: nn.Embedding(...) without padding_idx


def fm_predict(ids, dense, p, padding_idx=NO_PADDING, slot_offsets=None):
    """sigmoid(y_first_order + y_second_order + bias)                       fm/net.py:31-38"""
    y1, y2, feat = R.fm_forward(ids, dense, p["W1"], p["W"], p["dense_w_one"], p["dense_w"],
                                padding_idx, slot_offsets)
    z = y1 + y2 + p["bias"].reshape(1, 1)
    return R.sigmoid(z), z, (y1, y2, feat)


def fm_loss_and_grads(ids, dense, label, p, padding_idx=NO_PADDING, slot_offsets=None):
    """One train_forward + backward (fm/dygraph_model.py:74-88, tools/trainer.py:148-151)."""
    pred, z, (y1, y2, feat) = fm_predict(ids, dense, p, padding_idx, slot_offsets)
    loss = R.log_loss_mean(pred, label)
    dz = R.log_loss_mean_grad_z(pred, label)
    g = R.fm_backward(ids, dense, feat, np.zeros_like(feat), dz, dz, padding_idx, slot_offsets)
    g.update(d_bias=dz.sum(axis=0, dtype=dz.dtype).reshape(1), loss=loss, pred=pred, dz=dz, feat=feat, y1=y1, y2=y2)
    return g
