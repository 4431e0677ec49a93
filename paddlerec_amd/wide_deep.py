"""rank/wide_deep on the engine — a sibling net that reuses the lookup / GEMM / sparse-optimizer kernels
(one of the nets the reference wires for gpubox).

Host mirror of /root/reference/models/rank/wide_deep/net.py (`WideDeepLayer`, non-gpubox branch) and
wide_deep/dygraph_model.py (`DygraphModel`):
    wide = Linear(dense)                                     net.py:76     rec_gemm_f32 (N = 1 row kernel, bias fused)
    x    = concat(rows of the 26 slots, dense)               net.py:78-96  rec_emb_gather straight into the row head
    deep = Linear+ReLU ... Linear(x)                         net.py:97-99  rec_gemm_f32 (bias / ReLU fused)
    pred = sigmoid(wide + deep)                              net.py:101-103
The Embedding has no padding_idx (id 0 is an ordinary, trained row) and starts Uniform(-1,1) (net.py:48-54).
Backward: explicit chain (MLP dX/dW GEMMs with the ReLU mask and bias sums fused, the embedding gradient is the
first 26*D columns of d x read in place through rec_grad_layout, merged + lazy Adam per touched row).
Parameter keys follow the reference's sublayer names: wide_part.{weight,bias}, embedding.weight, linear_i.{weight,bias}.
"""
import math

import torch

from . import ops
from .deepfm import NUM_THRESHOLDS, _FlatParams, _OnSide, _round_up, auc_metrics, slot_feeds


class SlotMLPBase:
    """What rank/wide_deep and rank/dnn share (wide_deep/net.py:43-99 == dnn/net.py:38-95): one Uniform-initialised
    embedding table without padding_idx, x = concat(rows of the slots, dense), an MLP of Linear+ReLU ... Linear(n_out),
    lazy Adam on the touched rows read through rec_grad_layout, dense Adam on one flat buffer."""

    def _build(self, sparse_feature_number, sparse_feature_dim, dense_feature_dim, num_field, layer_sizes, n_out,
               extra_shapes, device, kernels):
        self.device = torch.device(device)
        self.k = kernels if kernels is not None else ops     # tests may inject a stand-in backend (host logic only)
        self.sparse_feature_number = N = sparse_feature_number
        self.sparse_feature_dim = D = sparse_feature_dim
        self.dense_feature_dim = Dn = dense_feature_dim
        self.num_field = S = num_field
        self.layer_sizes = list(layer_sizes)
        self.width = S * D + Dn                                                   # net.py:56
        self.rec = torch.zeros(N, _round_up(D, 32), dtype=torch.float32, device=self.device)   # line-aligned rows
        self.embedding = self.rec[:, :D]
        self.embedding.uniform_(-1.0, 1.0)                                        # net.py:48-54 Uniform() [EXT -1..1]
        sizes = [self.width] + self.layer_sizes + [n_out]
        shapes = list(extra_shapes)
        for i in range(len(sizes) - 1):
            shapes += [("linear_%d.weight" % i, (sizes[i], sizes[i + 1])), ("linear_%d.bias" % i, (sizes[i + 1],))]
        self.dense = _FlatParams(shapes, self.device)
        self.n_linear = len(sizes) - 1
        for i in range(self.n_linear):                                            # net.py:60-66
            self.dense.p["linear_%d.weight" % i].normal_(0.0, 1.0 / math.sqrt(sizes[i]))
        p, g = self.dense.p, self.dense.g
        self.mlp_w = [p["linear_%d.weight" % i] for i in range(self.n_linear)]
        self.mlp_b = [p["linear_%d.bias" % i] for i in range(self.n_linear)]
        self.mlp_dw = [g["linear_%d.weight" % i] for i in range(self.n_linear)]
        self.mlp_db = [g["linear_%d.bias" % i] for i in range(self.n_linear)]
        self.sparse_state = None
        self.ws = self.k.Workspace(self.device)
        self.ws_group = self.k.Workspace(self.device)
        self.ws_mlp = self.k.Workspace(self.device)
        self.status = self.k.new_status(self.device)
        self.step_count = 0
        self._side = None
        self._groups = None
        self._zeros = None

    # -- parameters under the reference's state_dict keys ---------------------------------------
    def state_dict(self):
        sd = {"embedding.weight": self.embedding}
        sd.update(self.dense.p)
        return sd

    def set_dict(self, sd):
        cur = self.state_dict()
        for k, v in sd.items():
            cur[k].copy_(torch.as_tensor(v).to(self.device).reshape(cur[k].shape))

    def parameters(self):
        return list(self.state_dict().values())

    @staticmethod
    def _concat_ids(sparse_inputs):
        if isinstance(sparse_inputs, (list, tuple)):
            return torch.cat(list(sparse_inputs), dim=1).contiguous()
        return sparse_inputs

    def _features(self, ids, dense_inputs):
        """net.py:78-96: every slot's row written straight into the head of the sample's feature row (no [B,26,D]
        intermediate, no concat pass), the raw dense values behind them."""
        B, S = ids.shape
        D = self.sparse_feature_dim
        x = torch.empty(B, self.width, dtype=torch.float32, device=self.device)
        self.k.emb_gather(ids.reshape(-1), self.embedding, None, self.status, out=x, out_group=S,
                          out_group_stride=self.width)
        x[:, S * D:].copy_(dense_inputs)
        return x

    def _ensure_sparse_state(self):
        if self.sparse_state is None:
            D = self.sparse_feature_dim
            Dp = _round_up(D, 4)
            mv = torch.zeros(self.rec.shape[0], _round_up(2 * Dp, 32), dtype=torch.float32, device=self.device)
            self.sparse_state = dict(mv=mv, m=mv[:, :D], v=mv[:, Dp:Dp + D])

    def _begin_step(self, ids):
        """Bookkeeping every train_step starts with -> (t, on_gpu, cur, side, groups)."""
        k = self.k
        B, S = ids.shape
        self._ensure_sparse_state()
        self.step_count += 1
        on_gpu = self.device.type == "cuda"
        cur = torch.cuda.current_stream() if on_gpu else None
        if on_gpu and self._side is None:
            self._side = k.concurrent_stream(self.device)
        if self._groups is None or self._groups.n != B * S:
            self._groups = k.IdGroups(B * S, self.device)
        return self.step_count, on_gpu, cur, (self._side if on_gpu else None), self._groups

    def _finish_step(self, groups, dx, S, t, lr, on_gpu, cur, side):
        """Lazy Adam on the touched embedding rows (SelectedRows.value = the first S*D columns of d x, read in place,
        on the side stream) + dense Adam."""
        k, D, st = self.k, self.sparse_feature_dim, self.sparse_state
        with _OnSide(side, cur):
            pp = self._pp = k.segment_partials(groups, dx, D, grad_group=S, grad_group_stride=self.width,
                                               out=getattr(self, "_pp", None))
            k.sparse_adam_rows(groups, dx, 1, self.embedding, st["m"], st["v"], t, lr, grad_group=S,
                               grad_group_stride=self.width, partials=pp)
        k.adam_dense(self.dense.data, self.dense.m, self.dense.v, self.dense.grad, t, lr)
        if on_gpu:
            cur.wait_stream(self._side)


class WideDeepLayer(SlotMLPBase):
    """wide_deep/net.py:20-104.  forward(sparse_inputs, dense_inputs) -> predict [B,1]."""

    def __init__(self, sparse_feature_number, sparse_feature_dim, dense_feature_dim, num_field, layer_sizes,
                 device="cuda", kernels=None):
        Dn = dense_feature_dim
        self._build(sparse_feature_number, sparse_feature_dim, dense_feature_dim, num_field, layer_sizes, 1,
                    [("wide_part.weight", (Dn, 1)), ("wide_part.bias", (1,))], device, kernels)
        std = 1.0 / math.sqrt(Dn)                                                 # net.py:36-41
        torch.nn.init.trunc_normal_(self.dense.p["wide_part.weight"], 0.0, std, -2 * std, 2 * std)

    def _wide(self, dense_inputs):
        return self.k.gemm(dense_inputs, self.dense.p["wide_part.weight"], self.ws, epilogue="bias",
                           bias=self.dense.p["wide_part.bias"])

    def forward(self, sparse_inputs, dense_inputs):
        ids = self._concat_ids(sparse_inputs)
        wide = self._wide(dense_inputs)
        deep, _ = self.k.mlp_forward(self._features(ids, dense_inputs), self.mlp_w, self.mlp_b, self.ws_mlp)
        return torch.sigmoid(wide + deep)

    __call__ = forward

    # -- one full training step: train_forward + backward + optimizer.step ----------------------
    def train_step(self, sparse_inputs, dense_inputs, label, lr=1e-3, auc_stats=None):
        """wide_deep/dygraph_model.py:73-85 + tools/trainer.py:148-152.  label [B,1] int64.
        Returns (loss [1] device tensor, pred [B,1])."""
        k = self.k
        ids = self._concat_ids(sparse_inputs)
        B, S = ids.shape
        t, on_gpu, cur, side, groups = self._begin_step(ids)
        x = self._features(ids, dense_inputs)
        with _OnSide(side, cur):                                   # merge keys depend on the ids only
            k.ids_group(ids, self.sparse_feature_number, None, self.ws_group, None, self.status, groups)
        wide = self._wide(dense_inputs)
        deep, acts = k.mlp_forward(x, self.mlp_w, self.mlp_b, self.ws_mlp)
        if self._zeros is None or self._zeros.shape[0] != B:
            self._zeros = torch.zeros(B, 1, dtype=torch.float32, device=self.device)
        pred, dz, loss = k.sigmoid_logloss(wide, self._zeros, deep, label, self.ws)     # logit = wide + 0 + deep
        if auc_stats is not None:
            k.auc_histogram(pred, label, auc_stats[0], auc_stats[1], NUM_THRESHOLDS)
        dx = k.mlp_backward(dz, acts, self.mlp_w, self.mlp_dw, self.mlp_db, self.ws_mlp)  # [B, width]
        k.gemm(dense_inputs, dz, self.ws, trans_a=True, out=self.dense.g["wide_part.weight"],
               b_colsum=self.dense.g["wide_part.bias"])                                # wide part: dW, db
        self._finish_step(groups, dx, S, t, lr, on_gpu, cur, side)
        return loss, pred


class DygraphModel:
    """wide_deep/dygraph_model.py:23-96 — same method names; tensors are torch device tensors."""

    def create_model(self, config, device="cuda", kernels=None):
        g = config.get
        return WideDeepLayer(g("hyper_parameters.sparse_feature_number"), g("hyper_parameters.sparse_feature_dim"),
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
        pred = dy_model.forward(sparse, dense)
        if metrics_list:
            dy_model.k.auc_histogram(pred.contiguous(), label.contiguous(), metrics_list[0][0], metrics_list[0][1],
                                     NUM_THRESHOLDS)
        return metrics_list, None
