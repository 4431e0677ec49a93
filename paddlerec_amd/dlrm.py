"""rank/dlrm on the engine (the bmm + triu sibling).

Host mirror of /root/reference/models/rank/dlrm/net.py (`DLRMLayer` :23-125, `MLPLayer` :128-178) and
dlrm/dygraph_model.py (`DygraphModel`).

    x   = bot_mlp(dense)                      every MLP layer is Linear -> ReLU -> BatchNorm1D (net.py:142-156; the
                                              `else` at :158 is dead code, so the last layer is normalised too):
                                              rec_gemm_f32 (bias + ReLU epilogue) -> rec_batchnorm_fwd
    T   = [emb(s_1) .. emb(s_26), x]          rec_emb_gather writes the 26 rows straight into T [B,27,D]; the last
                                              BatchNorm of bot_mlp writes x into T's last field (no concat pass)
    R   = [x | <T_i, T_j>, i < j]             rec_dot_interact_fwd (bmm + triu + masked_select + concat, net.py:96-123)
    raw = top_mlp(R)  [B,2]                   two class scores (ReLU'd and batch-normalised like every layer)
    loss = mean softmax CE(raw, label)        dygraph_model.py:53-57 — exactly BCE-with-logits on raw1 - raw0 for two
                                              classes (the identity rank/dnn uses, oracle/dnn_ref.py)
Backward is the explicit chain (rec_batchnorm_bwd with the ReLU mask folded in, dW / dX GEMMs, rec_dot_interact_bwd,
the embedding gradient read in place from d T through rec_grad_layout); optimizer: the reference's
`paddle.optimizer.Adam(parameters=...)` (dygraph_model.py:60-64) = non-lazy Adam: every table row's moments move each
step (rec_adam_rows_all), dense parameters in one flat Adam.  BatchNorm running statistics are buffers, not parameters.
Parameter keys follow the reference's sublayer names: embedding.weight, {bot,top}_mlp.dense_i.{weight,bias},
{bot,top}_mlp.norm_i.{weight,bias,_mean,_variance}.
"""
import math

import torch

from . import ops
from .deepfm import NUM_THRESHOLDS, _FlatParams, _OnSide, _round_up, auc_from_buckets, auc_metrics, slot_feeds

BN_MOMENTUM, BN_EPS = 0.9, 1e-5       # paddle.nn.BatchNorm1D defaults [EXT]


class DLRMLayer:
    """dlrm/net.py:23-125.  forward(sparse_inputs, dense_inputs) -> raw [B,2] (unnormalised class scores)."""

    def __init__(self, dense_feature_dim, bot_layer_sizes, sparse_feature_number, sparse_feature_dim, top_layer_sizes,
                 num_field, sync_mode=None, self_interaction=False, device="cuda", kernels=None):
        if self_interaction:
            raise NotImplementedError("self_interaction=True (the reference's dygraph_model.py passes False)")
        self.device = torch.device(device)
        self.k = kernels if kernels is not None else ops
        self.dense_feature_dim, self.sparse_feature_number = dense_feature_dim, sparse_feature_number
        self.sparse_feature_dim = D = sparse_feature_dim
        self.num_field = S = num_field
        self.bot_layer_sizes, self.top_layer_sizes = list(bot_layer_sizes), list(top_layer_sizes)
        if self.bot_layer_sizes[-1] != D:
            raise ValueError("bot_mlp must end in sparse_feature_dim (%d) outputs: x joins the embeddings in T" % D)
        self.concat_size = S * (S + 1) // 2                                        # net.py:58-60
        N = sparse_feature_number
        self.rec = torch.zeros(N, _round_up(D, 32), dtype=torch.float32, device=self.device)   # line-aligned rows
        self.embedding = self.rec[:, :D]
        torch.nn.init.trunc_normal_(self.embedding, 0.0, 1.0, -2.0, 2.0)           # net.py:70-77 TruncatedNormal()
        self.mlps = {"bot_mlp": [dense_feature_dim] + self.bot_layer_sizes,
                     "top_mlp": [self.concat_size + D] + self.top_layer_sizes}
        shapes = []
        for name, sizes in self.mlps.items():
            for i in range(len(sizes) - 1):
                shapes += [("%s.dense_%d.weight" % (name, i), (sizes[i], sizes[i + 1])),
                           ("%s.dense_%d.bias" % (name, i), (sizes[i + 1],)),
                           ("%s.norm_%d.weight" % (name, i), (sizes[i + 1],)),
                           ("%s.norm_%d.bias" % (name, i), (sizes[i + 1],))]
        self.dense = _FlatParams(shapes, self.device)
        self.buffers = {}
        for name, sizes in self.mlps.items():
            for i in range(len(sizes) - 1):
                std = 1.0 / math.sqrt(sizes[i])                                    # net.py:136-141
                torch.nn.init.trunc_normal_(self.dense.p["%s.dense_%d.weight" % (name, i)], 0.0, std, -2 * std, 2 * std)
                self.dense.p["%s.norm_%d.weight" % (name, i)].fill_(1.0)
                self.buffers["%s.norm_%d._mean" % (name, i)] = torch.zeros(sizes[i + 1], device=self.device)
                self.buffers["%s.norm_%d._variance" % (name, i)] = torch.ones(sizes[i + 1], device=self.device)
        self.sparse_state = None
        self.ws = self.k.Workspace(self.device)
        self.ws_group = self.k.Workspace(self.device)
        self.ws_bn = self.k.Workspace(self.device)
        self.status = self.k.new_status(self.device)
        self._diff = torch.tensor([[-1.0], [1.0]], dtype=torch.float32, device=self.device)     # raw -> raw1 - raw0
        self._diff_t = self._diff.t().contiguous()
        self.step_count = 0
        self.training = True
        self._side = None
        self._groups = None

    # -- parameters under the reference's state_dict keys ---------------------------------------
    def state_dict(self):
        sd = {"embedding.weight": self.embedding}
        sd.update(self.dense.p)
        sd.update(self.buffers)
        return sd

    def set_dict(self, sd):
        cur = self.state_dict()
        for k, v in sd.items():
            cur[k].copy_(torch.as_tensor(v).to(self.device).reshape(cur[k].shape))

    def parameters(self):
        return [self.embedding] + list(self.dense.p.values())

    def train(self):
        self.training = True

    def eval(self):
        self.training = False

    @staticmethod
    def _concat_ids(sparse_inputs):
        if isinstance(sparse_inputs, (list, tuple)):
            return torch.cat(list(sparse_inputs), dim=1).contiguous()
        return sparse_inputs

    # -- MLP of Linear -> ReLU -> BatchNorm layers ------------------------------------------------
    def _mlp_forward(self, name, x, training, out_last=None):
        k, p = self.k, self.dense.p
        cache = []
        n = len(self.mlps[name]) - 1
        for i in range(n):
            h = k.gemm(x, p["%s.dense_%d.weight" % (name, i)], self.ws, epilogue="bias_relu",
                       bias=p["%s.dense_%d.bias" % (name, i)])
            y, mu, invstd = k.batchnorm_fwd(h, p["%s.norm_%d.weight" % (name, i)], p["%s.norm_%d.bias" % (name, i)],
                                            self.buffers["%s.norm_%d._mean" % (name, i)],
                                            self.buffers["%s.norm_%d._variance" % (name, i)], self.ws_bn, training,
                                            BN_MOMENTUM, BN_EPS, out=out_last if i == n - 1 else None)
            cache.append((x, h, mu, invstd))
            x = y
        return x, cache

    def _mlp_backward(self, name, dy, cache):
        k, p, g = self.k, self.dense.p, self.dense.g
        for i in reversed(range(len(cache))):
            x, h, mu, invstd = cache[i]
            dpre, _, _ = k.batchnorm_bwd(h, dy, p["%s.norm_%d.weight" % (name, i)], mu, invstd, self.ws_bn,
                                         relu_mask=True, dgamma=g["%s.norm_%d.weight" % (name, i)],
                                         dbeta=g["%s.norm_%d.bias" % (name, i)])
            k.gemm(x, dpre, self.ws, trans_a=True, out=g["%s.dense_%d.weight" % (name, i)],
                   b_colsum=g["%s.dense_%d.bias" % (name, i)])
            dy = k.gemm(dpre, p["%s.dense_%d.weight" % (name, i)], self.ws, trans_b=True)
        return dy

    def _forward(self, ids, dense_inputs, training, keep=None):
        k, D = self.k, self.sparse_feature_dim
        B, S = ids.shape
        T = torch.empty(B, S + 1, D, dtype=torch.float32, device=self.device)
        flat = T.view(B, (S + 1) * D)
        k.emb_gather(ids.reshape(-1), self.embedding, None, self.status, out=flat, out_group=S,
                     out_group_stride=(S + 1) * D)                                            # net.py:87-92
        _, cb = self._mlp_forward("bot_mlp", dense_inputs, training, out_last=flat[:, S * D:])   # x -> T[:, S, :]
        R = k.dot_interact_fwd(T)                                                             # net.py:96-123
        raw, ct = self._mlp_forward("top_mlp", R, training)
        if keep is not None:
            keep.update(T=T, cb=cb, ct=ct)
        return raw

    def forward(self, sparse_inputs, dense_inputs):
        return self._forward(self._concat_ids(sparse_inputs), dense_inputs, self.training)

    __call__ = forward

    def predict(self, raw):
        """softmax(raw)[:, 1:2] (dygraph_model.py:78) = sigmoid(raw1 - raw0)."""
        return torch.sigmoid(self.k.gemm(raw, self._diff, self.ws))

    def _ensure_sparse_state(self):
        if self.sparse_state is None:
            D = self.sparse_feature_dim
            Dp = _round_up(D, 4)
            mv = torch.zeros(self.rec.shape[0], _round_up(2 * Dp, 32), dtype=torch.float32, device=self.device)
            self.sparse_state = dict(mv=mv, m=mv[:, :D], v=mv[:, Dp:Dp + D])

    def train_step(self, sparse_inputs, dense_inputs, label, lr=1e-3, auc_stats=None, acc_counts=None):
        """dlrm/dygraph_model.py:74-91 + tools/trainer.py:148-152.  label [B,1] int64.
        Returns (loss [1] device tensor, pred [B,1] = P(click))."""
        k, D = self.k, self.sparse_feature_dim
        ids = self._concat_ids(sparse_inputs)
        B, S = ids.shape
        self._ensure_sparse_state()
        self.step_count += 1
        on_gpu = self.device.type == "cuda"
        cur = torch.cuda.current_stream() if on_gpu else None
        if on_gpu and self._side is None:
            self._side = k.concurrent_stream(self.device)
        side = self._side if on_gpu else None
        if self._groups is None or self._groups.n != B * S:
            self._groups = k.IdGroups(B * S, self.device)
        groups = self._groups
        sv = {}
        raw = self._forward(ids, dense_inputs, True, keep=sv)
        with _OnSide(side, cur):                                   # merge keys depend on the ids only
            k.ids_group(ids, self.sparse_feature_number, None, self.ws_group, None, self.status, groups)
        d = k.gemm(raw, self._diff, self.ws)                                               # [B,1] = raw1 - raw0
        pred, dd, loss = k.bce_with_logits(d, label.to(torch.float32).reshape(B, 1).contiguous(), self.ws)
        if auc_stats is not None:
            k.auc_histogram(pred, label, auc_stats[0], auc_stats[1], NUM_THRESHOLDS)
        if acc_counts is not None:
            k.accuracy_count(pred, label.contiguous(), acc_counts)
        draw = k.gemm(dd, self._diff_t, self.ws)                                           # [B,2] = (-dd, +dd)
        dR = self._mlp_backward("top_mlp", draw, sv["ct"])
        dT = k.dot_interact_bwd(sv["T"], dR)                                               # [B,S+1,D]
        dflat = dT.view(B, (S + 1) * D)
        self._mlp_backward("bot_mlp", dflat[:, S * D:], sv["cb"])
        t, st = self.step_count, self.sparse_state
        with _OnSide(side, cur):   # non-lazy Adam: every row's moments move; touched rows get their merged gradient
            pp = self._pp = k.segment_partials(groups, dflat, D, grad_group=S, grad_group_stride=(S + 1) * D,
                                               out=getattr(self, "_pp", None))
            k.adam_rows_all(groups, dflat, 1, self.embedding, st["m"], st["v"], t, lr, grad_group=S,
                            grad_group_stride=(S + 1) * D, partials=pp)
        k.adam_dense(self.dense.data, self.dense.m, self.dense.v, self.dense.grad, t, lr)
        if on_gpu:
            cur.wait_stream(self._side)
        return loss, pred


class DygraphModel:
    """dlrm/dygraph_model.py:23-107 — same method names; tensors are torch device tensors."""

    def create_model(self, config, device="cuda", kernels=None):
        g = config.get
        return DLRMLayer(dense_feature_dim=g("hyper_parameters.dense_input_dim"),
                         bot_layer_sizes=g("hyper_parameters.bot_layer_sizes"),
                         sparse_feature_number=g("hyper_parameters.sparse_feature_number"),
                         sparse_feature_dim=g("hyper_parameters.sparse_feature_dim"),
                         top_layer_sizes=g("hyper_parameters.top_layer_sizes"),
                         num_field=g("hyper_parameters.num_field", g("hyper_parameters.sparse_inputs_slots") - 1),
                         self_interaction=False, device=device, kernels=kernels)

    def create_feeds(self, batch_data, config, device="cuda"):
        return slot_feeds(batch_data, config, device)

    def create_metrics(self, device="cuda"):
        """[Auc("ROC"), Accuracy()] (dygraph_model.py:58-63): the AUC bucket pair and a (correct, total) int64 pair."""
        auc, _ = auc_metrics(device)
        return auc + [torch.zeros(2, dtype=torch.int64, device=device)], ["auc", "accuracy"]

    @staticmethod
    def metric_value(name, m):
        if name == "accuracy":
            c = m.tolist() if torch.is_tensor(m) else [int(x) for x in m]
            return c[0] / c[1] if c[1] else 0.0
        return auc_from_buckets(m[0], m[1])

    def train_forward(self, dy_model, metrics_list, batch_data, config):
        label, sparse, dense = self.create_feeds(batch_data, config, dy_model.device)
        lr = config.get("hyper_parameters.optimizer.learning_rate", 0.001)
        dy_model.train()
        loss, _ = dy_model.train_step(sparse, dense, label, lr, metrics_list[0] if metrics_list else None,
                                      metrics_list[1] if metrics_list and len(metrics_list) > 1 else None)
        return loss, metrics_list, {"loss": loss}

    def infer_forward(self, dy_model, metrics_list, batch_data, config):
        label, sparse, dense = self.create_feeds(batch_data, config, dy_model.device)
        dy_model.eval()
        pred = dy_model.predict(dy_model.forward(sparse, dense)).contiguous()
        if metrics_list:
            dy_model.k.auc_histogram(pred, label.contiguous(), metrics_list[0][0], metrics_list[0][1], NUM_THRESHOLDS)
            if len(metrics_list) > 1:
                dy_model.k.accuracy_count(pred, label.contiguous(), metrics_list[1])
        return metrics_list, None
