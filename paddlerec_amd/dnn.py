"""rank/dnn on the engine — the second net the reference wires for gpubox, non-gpubox branch.

Host mirror of /root/reference/models/rank/dnn/net.py (`DNNLayer`, :20-95) and dnn/dygraph_model.py (`DygraphModel`):
the lookup + concat + MLP of wide_deep.py (shared base `SlotMLPBase`) with a last Linear of TWO outputs, softmax
cross-entropy (dygraph_model.py:53-58) and the AUC on softmax(raw)[:,1] (:78-80).

For two classes the head needs no softmax kernel (oracle/dnn_ref.py states and checks the identity):
    CE(raw, t) = BCE_with_logits(raw[:,1] - raw[:,0], t),   softmax(raw)[:,1] = sigmoid(raw[:,1] - raw[:,0])
so  d = raw @ [-1, +1]^T  (rec_gemm_f32, N = 1),  rec_bce_with_logits(d, t) -> pred, d loss/d d,
    d raw = (d loss/d d) @ [-1, +1]   (rec_gemm_f32, K = 1),  then the usual MLP backward chain.
"""
import torch

from .deepfm import NUM_THRESHOLDS, _OnSide, auc_metrics, slot_feeds
from .wide_deep import SlotMLPBase


class DNNLayer(SlotMLPBase):
    """dnn/net.py:20-95.  forward(sparse_inputs, dense_inputs) -> raw [B,2] (unnormalised class scores)."""

    def __init__(self, sparse_feature_number, sparse_feature_dim, dense_feature_dim, num_field, layer_sizes,
                 device="cuda", kernels=None):
        self._build(sparse_feature_number, sparse_feature_dim, dense_feature_dim, num_field, layer_sizes, 2, [],
                    device, kernels)
        self._diff = torch.tensor([[-1.0], [1.0]], dtype=torch.float32, device=self.device)     # raw -> raw1 - raw0
        self._diff_t = self._diff.t().contiguous()

    def forward(self, sparse_inputs, dense_inputs):
        ids = self._concat_ids(sparse_inputs)
        raw, _ = self.k.mlp_forward(self._features(ids, dense_inputs), self.mlp_w, self.mlp_b, self.ws_mlp)
        return raw

    __call__ = forward

    def predict(self, raw):
        """softmax(raw)[:, 1:2] (dygraph_model.py:78) = sigmoid(raw1 - raw0)."""
        return torch.sigmoid(self.k.gemm(raw, self._diff, self.ws))

    def train_step(self, sparse_inputs, dense_inputs, label, lr=1e-3, auc_stats=None):
        """dnn/dygraph_model.py:74-86 + tools/trainer.py:148-152.  label [B,1] int64.
        Returns (loss [1] device tensor, pred [B,1] = P(click))."""
        k = self.k
        ids = self._concat_ids(sparse_inputs)
        B, S = ids.shape
        t, on_gpu, cur, side, groups = self._begin_step(ids)
        x = self._features(ids, dense_inputs)
        with _OnSide(side, cur):                                   # merge keys depend on the ids only
            k.ids_group(ids, self.sparse_feature_number, None, self.ws_group, None, self.status, groups)
        raw, acts = k.mlp_forward(x, self.mlp_w, self.mlp_b, self.ws_mlp)
        d = k.gemm(raw, self._diff, self.ws)                                               # [B,1] = raw1 - raw0
        pred, dd, loss = k.bce_with_logits(d, label.to(torch.float32).reshape(B, 1).contiguous(), self.ws)
        if auc_stats is not None:
            k.auc_histogram(pred, label, auc_stats[0], auc_stats[1], NUM_THRESHOLDS)
        draw = k.gemm(dd, self._diff_t, self.ws)                                           # [B,2] = (-dd, +dd)
        dx = k.mlp_backward(draw, acts, self.mlp_w, self.mlp_dw, self.mlp_db, self.ws_mlp)
        self._finish_step(groups, dx, S, t, lr, on_gpu, cur, side)
        return loss, pred


class DygraphModel:
    """dnn/dygraph_model.py:23-98 — same method names; tensors are torch device tensors."""

    def create_model(self, config, device="cuda", kernels=None):
        g = config.get
        return DNNLayer(g("hyper_parameters.sparse_feature_number"), g("hyper_parameters.sparse_feature_dim"),
                        g("hyper_parameters.dense_input_dim"), g("hyper_parameters.sparse_inputs_slots") - 1,
                        g("hyper_parameters.fc_sizes"), device=device, kernels=kernels)

    def create_feeds(self, batch_data, config, device="cuda"):
        return slot_feeds(batch_data, config, device)

    def create_metrics(self, device="cuda"):
        return auc_metrics(device)

    def train_forward(self, dy_model, metrics_list, batch_data, config):
        label, sparse, dense = self.create_feeds(batch_data, config, dy_model.device)
        lr = config.get("hyper_parameters.optimizer.learning_rate", 0.001)
        loss, _ = dy_model.train_step(sparse, dense, label, lr, metrics_list[0] if metrics_list else None)
        return loss, metrics_list, {"loss": loss}

    def infer_forward(self, dy_model, metrics_list, batch_data, config):
        label, sparse, dense = self.create_feeds(batch_data, config, dy_model.device)
        pred = dy_model.predict(dy_model.forward(sparse, dense))
        if metrics_list:
            dy_model.k.auc_histogram(pred.contiguous(), label.contiguous(), metrics_list[0][0], metrics_list[0][1],
                                     NUM_THRESHOLDS)
        return metrics_list, None
