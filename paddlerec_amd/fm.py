"""rank/fm on the engine — a sibling net that reuses the DeepFM kernels.

Host mirror of /root/reference/models/rank/fm/net.py (`FMLayer`, `FM`) and fm/dygraph_model.py (`DygraphModel`):
    predict = sigmoid(y_first_order + y_second_order + bias)                      net.py:31-38
with the FM block of net.py:90-124 = `rec_deepfm_fm_fwd` / `_bwd` (the same arithmetic as deepfm/net.py:105-139).
What differs from DeepFM (oracle/fm_ref.py): no DNN tower (the backward gets a zero d_feat_dnn), the scalar `bias`
IS part of the logit, the Embeddings have NO padding_idx (id 0 is an ordinary, trained row), and the dense weights
start at Constant(1.0).  Optimizer: Adam (dygraph_model.py:60-65), lazy rows by default as in deepfm.py.
There is no autograd tape and no CPU fallback.
"""
import torch

from . import ops
from .deepfm import FM, NUM_THRESHOLDS, _FlatParams, _OnSide, auc_metrics, slot_feeds


class FMLayer:
    """fm/net.py:20-38.  forward(sparse_inputs, dense_inputs) -> predict [B,1]."""

    def __init__(self, sparse_feature_number, sparse_feature_dim, dense_feature_dim, sparse_num_field,
                 device="cuda", kernels=None):
        self.device = torch.device(device)
        self.k = kernels if kernels is not None else ops     # tests may inject a stand-in backend (host logic only)
        self.sparse_feature_number = sparse_feature_number
        self.sparse_feature_dim = sparse_feature_dim
        self.dense_feature_dim = dense_feature_dim
        self.sparse_num_field = sparse_num_field
        self.fm = FM(sparse_feature_number, sparse_feature_dim, dense_feature_dim, sparse_num_field, self.device,
                     None, zero_padding_row=False)
        self.fm.padding_idx = None                                           # net.py:55-73: no padding_idx
        D, Dn = sparse_feature_dim, dense_feature_dim
        self.dense = _FlatParams([("fm.dense_w_one", (Dn,)), ("fm.dense_w", (1, Dn, D)), ("bias", (1,))],
                                 self.device)
        self.dense.p["fm.dense_w_one"].fill_(1.0)                            # net.py:78-82 Constant(1.0)
        self.dense.p["fm.dense_w"].fill_(1.0)                                # net.py:84-88
        self.sparse_state = None
        self.lazy_mode = True
        self.ws = self.k.Workspace(self.device)
        self.ws_group = self.k.Workspace(self.device)
        self.status = self.k.new_status(self.device)
        self.step_count = 0
        self._side = None
        self._groups = None
        self._zero_dfeat = None

    # -- parameters under the reference's state_dict keys ---------------------------------------
    def state_dict(self):
        sd = {"fm.embedding_one.weight": self.fm.embedding_one, "fm.embedding.weight": self.fm.embedding}
        sd.update(self.dense.p)
        return sd

    def set_dict(self, sd):
        for k, v in sd.items():
            dst = self.state_dict()[k]
            dst.copy_(torch.as_tensor(v).to(dst.device).reshape(dst.shape))

    def parameters(self):
        return list(self.state_dict().values())

    @staticmethod
    def _concat_ids(sparse_inputs):
        if isinstance(sparse_inputs, (list, tuple)):
            return torch.cat(list(sparse_inputs), dim=1).contiguous()        # net.py:92
        return sparse_inputs

    def _fm_fwd(self, ids, dense_inputs):
        return self.k.deepfm_fm_fwd(ids, dense_inputs, self.fm.embedding, self.fm.embedding_one,
                                    self.dense.p["fm.dense_w"], self.dense.p["fm.dense_w_one"], None, None,
                                    self.status)

    def forward(self, sparse_inputs, dense_inputs):
        ids = self._concat_ids(sparse_inputs)
        y1, y2, _, _, _ = self._fm_fwd(ids, dense_inputs)
        return torch.sigmoid(y1 + y2 + self.dense.p["bias"])

    __call__ = forward

    def _ensure_sparse_state(self):
        if self.sparse_state is None:
            D = self.sparse_feature_dim
            Dp = (D + 3) // 4 * 4
            mv = torch.zeros(self.fm.rec.shape[0], (2 * Dp + 31) // 32 * 32, dtype=torch.float32, device=self.device)
            self.sparse_state = dict(mv=mv, m=mv[:, :D], v=mv[:, Dp:Dp + D],
                                     m1=self.fm.rec[:, D + 1:D + 2], v1=self.fm.rec[:, D + 2:D + 3])

    # -- one full training step: train_forward + backward + optimizer.step ----------------------
    def train_step(self, sparse_inputs, dense_inputs, label, lr=1e-3, auc_stats=None):
        """fm/dygraph_model.py:74-88 + tools/trainer.py:148-152.  label [B,1] int64.
        Returns (loss [1] device tensor, pred [B,1])."""
        k = self.k
        ids = self._concat_ids(sparse_inputs)
        B, S = ids.shape
        D, Dn = self.sparse_feature_dim, self.dense_feature_dim
        self._ensure_sparse_state()
        self.step_count += 1
        t = self.step_count
        on_gpu = self.device.type == "cuda"
        cur = torch.cuda.current_stream() if on_gpu else None
        if on_gpu and self._side is None:
            self._side = k.concurrent_stream(self.device)
        side = self._side if on_gpu else None
        if self._groups is None or self._groups.n != B * S:
            self._groups = k.IdGroups(B * S, self.device)
        groups = self._groups
        y1, y2, feat, sum_emb, _ = self._fm_fwd(ids, dense_inputs)
        with _OnSide(side, cur):                                   # merge keys depend on the ids only
            k.ids_group(ids, self.sparse_feature_number, None, self.ws_group, None, self.status, groups)
        bias_col = self.dense.p["bias"].expand(B, 1).contiguous()  # the logit's third term, one value per sample
        pred, dz, loss = k.sigmoid_logloss(y1, y2, bias_col, label, self.ws)
        if auc_stats is not None:
            k.auc_histogram(pred, label, auc_stats[0], auc_stats[1], NUM_THRESHOLDS)
        k.colsum(dz, self.ws, out=self.dense.g["bias"])            # d loss / d bias = sum_b dz[b]
        if self._zero_dfeat is None or self._zero_dfeat.shape != feat.shape:
            self._zero_dfeat = torch.zeros_like(feat)              # no DNN tower: d_feat_dnn = 0
        row_grad, _, _ = k.deepfm_fm_bwd(
            dense_inputs, feat, sum_emb, self._zero_dfeat, dz, dz, S, self.ws,
            out=(self._row_grad_buf(B * S), self.dense.g["fm.dense_w"].view(Dn, D), self.dense.g["fm.dense_w_one"]))
        st = self.sparse_state
        with _OnSide(side, cur):
            upd = k.sparse_adam_rows if self.lazy_mode else k.adam_rows_all
            pp = self._pp = k.segment_partials(groups, row_grad, D, out=getattr(self, "_pp", None))
            pp1 = self._pp1 = k.segment_partials(groups, dz, 1, grad_div=S, out=getattr(self, "_pp1", None))
            upd(groups, row_grad, 1, self.fm.embedding, st["m"], st["v"], t, lr, partials=pp)
            upd(groups, dz, S, self.fm.embedding_one, st["m1"], st["v1"], t, lr, partials=pp1)
        k.adam_dense(self.dense.data, self.dense.m, self.dense.v, self.dense.grad, t, lr)
        if on_gpu:
            cur.wait_stream(self._side)
        return loss, pred

    def _row_grad_buf(self, n):
        b = getattr(self, "_rg", None)
        if b is None or b.shape[0] != n:
            self._rg = torch.empty(n, self.sparse_feature_dim, dtype=torch.float32, device=self.device)
        return self._rg


class DygraphModel:
    """fm/dygraph_model.py:23-100 — same method names; tensors are torch device tensors."""

    def create_model(self, config, device="cuda", kernels=None):
        return FMLayer(config.get("hyper_parameters.sparse_feature_number"),
                       config.get("hyper_parameters.sparse_feature_dim"),
                       config.get("hyper_parameters.dense_input_dim"),
                       config.get("hyper_parameters.sparse_inputs_slots") - 1, device=device, kernels=kernels)

    def create_feeds(self, batch_data, config, device="cuda"):
        return slot_feeds(batch_data, config, device)

    def create_metrics(self, device="cuda"):
        return auc_metrics(device)

    def train_forward(self, dy_model, metrics_list, batch_data, config):
        label, sparse, dense = self.create_feeds(batch_data, config, dy_model.device)
        lr = config.get("hyper_parameters.optimizer.learning_rate", 0.001)
        loss, _ = dy_model.train_step(sparse, dense, label, lr, metrics_list[0] if metrics_list else None)
        return loss, metrics_list, None

    def infer_forward(self, dy_model, metrics_list, batch_data, config):
        label, sparse, dense = self.create_feeds(batch_data, config, dy_model.device)
        pred = dy_model.forward(sparse, dense)
        if metrics_list:
            dy_model.k.auc_histogram(pred.contiguous(), label.contiguous(), metrics_list[0][0], metrics_list[0][1],
                                     NUM_THRESHOLDS)
        return metrics_list, None
