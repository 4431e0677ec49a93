"""Host-side mirror of the reference's DCN-v2 plugin on the recengine HIP kernels.

Mirrors /root/reference/models/rank/dcn_v2/net.py:20-320 (DCN_V2Layer, DNNLayer, DeepCrossLayer,
CrossNetV2, CrossNetMix) and dcn_v2/dygraph_model.py (loss, Adam + ClipGradByGlobalNorm(10)) with the
reference's parameter names.  Kernels: rec_emb_gather (lookup straight into the
feature row), rec_gemm_f32 (dense_emb, CrossNet layers with the fused `x_l + x_0*(x_l W + b)`
epilogue, low-rank expert projections with fused tanh / gate mixing, MLP), rec_cross_bwd_prep,
rec_sparse_adam_rows with the clipping coefficient, rec_sumsq / rec_sparse_rows_sumsq.

Train mode (net.py:158,181-183; App. B-10): Dropout(dropout_rate) after EVERY element of the DNN tower's _mlp_layers —
after each Linear and again after its ReLU — with masks from rec_dropout's counter-based generator (Paddle's own
stream is not reproducible from the reference; the oracle restates the same generator), and L2Decay(l2_dnn) on the DNN
weights appended after the global-norm clip (net.py:164-170).  The bare layer defaults to dropout_rate 0 / l2_dnn 0 =
the reference's eval() arithmetic (what the golden fixtures hold); DygraphModel.create_model builds it with the
reference's 0.5 / 1e-7, so the trainer runs the reference's train-mode graph.
The sparse optimizer is lazy Adam (see deepfm.py).  Training covers both cross networks: CrossNetV2
(BASELINE config 3) and the shipped CrossNetMix (low-rank mixture of experts).
"""
import math
import os

import torch

from . import ops
from . import ops as _ops
from .deepfm import NUM_THRESHOLDS, _FlatParams, _round_up, _Timed, auc_metrics, slot_feeds

P = "DeepCrossLayer_.crossNet."


class DCN_V2Layer:
    """dcn_v2/net.py:20-137.  forward(sparse_inputs, dense_inputs) -> predict [B,1]."""

    def __init__(self, sparse_feature_number, sparse_feature_dim, dense_feature_dim, sparse_num_field,
                 layer_sizes, cross_num, is_Stacked=True, use_low_rank_mixture=False, low_rank=32,
                 num_experts=4, device="cuda", kernels=None, dropout_rate=0.0, dropout_seed=2025, l2_dnn=0.0):
        self.dropout_rate, self.dropout_seed, self.l2_dnn = float(dropout_rate), int(dropout_seed), float(l2_dnn)
        self.device = torch.device(device)
        self.k = kernels if kernels is not None else ops
        self.sparse_feature_number = N = sparse_feature_number
        self.sparse_feature_dim = D = sparse_feature_dim
        self.dense_feature_dim = Dn = dense_feature_dim
        self.sparse_num_field = S = sparse_num_field
        self.num_field = S + Dn
        self.layer_sizes = list(layer_sizes)
        self.cross_num, self.is_Stacked = cross_num, bool(is_Stacked)
        self.use_low_rank_mixture, self.low_rank, self.num_experts = bool(use_low_rank_mixture), low_rank, num_experts
        self.d = d = self.num_field * D
        f32 = dict(dtype=torch.float32, device=self.device)
        # embedding table: line-aligned records; padding_idx=0 (net.py:45-54)
        self.rec = torch.zeros(N, _round_up(D, 32), **f32)
        self.embedding = self.rec[:, :D]
        std = 0.1 / math.sqrt(float(D))
        torch.nn.init.trunc_normal_(self.embedding, 0.0, std, -2 * std, 2 * std)
        self.embedding[0].zero_()
        self.padding_idx = 0
        E, r = num_experts, low_rank
        shapes = [("dense_emb.weight", (Dn, D * Dn)), ("dense_emb.bias", (D * Dn,))]
        if self.use_low_rank_mixture:
            for i in range(cross_num):
                shapes += [(P + "U_list.%d" % i, (E, d, r)), (P + "V_list.%d" % i, (E, d, r)),
                           (P + "C_list.%d" % i, (E, r, r))]
            shapes += [(P + "gating.weight", (d, E)), (P + "gating.bias", (E,))]   # E Linear(d,1), stacked
            shapes += [(P + "bias.%d" % i, (d, 1)) for i in range(cross_num)]
        else:
            for i in range(cross_num):
                shapes += [(P + "cross_layers.%d.weight" % i, (d, d)), (P + "cross_layers.%d.bias" % i, (d,))]
        sizes = [d] + self.layer_sizes
        for i in range(len(self.layer_sizes)):
            shapes += [("DNN_.linear_%d.weight" % i, (sizes[i], sizes[i + 1])),
                       ("DNN_.linear_%d.bias" % i, (sizes[i + 1],))]
        fc_in = self.layer_sizes[-1] + (0 if self.is_Stacked else d)
        shapes += [("fc.weight", (fc_in, 1)), ("fc.bias", (1,))]
        self.dense = _FlatParams(shapes, self.device)
        p = self.dense.p
        # initialisers (net.py:56-57,164-170,70-87,241-276); Linear default = XavierUniform-like [EXT]
        for name, (fi, fo) in (("dense_emb.weight", (Dn, D * Dn)),):
            bound = math.sqrt(6.0 / (fi + fo))
            p[name].uniform_(-bound, bound)
        if self.use_low_rank_mixture:
            for i in range(cross_num):
                for nm in ("U_list", "V_list", "C_list"):
                    t = p[P + "%s.%d" % (nm, i)]
                    t.normal_(0.0, math.sqrt(2.0 / (t.shape[1] + t.shape[2])))
            p[P + "gating.weight"].uniform_(-math.sqrt(6.0 / (d + 1)), math.sqrt(6.0 / (d + 1)))
        else:
            for i in range(cross_num):
                p[P + "cross_layers.%d.weight" % i].uniform_(-math.sqrt(3.0 / d), math.sqrt(3.0 / d))
        for i in range(len(self.layer_sizes)):
            p["DNN_.linear_%d.weight" % i].normal_(0.0, 1.0 / math.sqrt(sizes[i]))
        p["fc.weight"].normal_(0.0, 1.0 / math.sqrt(self.layer_sizes[-1]))
        self.ws = self.k.Workspace(self.device)
        self.ws_group = self.k.Workspace(self.device)
        self.status = self.k.new_status(self.device)
        self.sparse_state = None
        self.step_count = 0
        self.timers = None
        self._plans, self._recording = {}, False          # recorded call lists of launch-bound steps (plan.py)
        self._groups = None
        self._side = None

    # ---------------------------------------------------------------- parameters (reference keys)
    def state_dict(self):
        sd = {"embedding.weight": self.embedding}
        for k, v in self.dense.p.items():
            if k == P + "gating.weight":
                for e in range(self.num_experts):
                    sd[P + "gating.%d.weight" % e] = v[:, e:e + 1]
            elif k == P + "gating.bias":
                for e in range(self.num_experts):
                    sd[P + "gating.%d.bias" % e] = v[e:e + 1]
            else:
                sd[k] = v
        return sd

    def set_dict(self, sd):
        cur = self.state_dict()
        for k, v in sd.items():
            dst = cur[k]
            dst.copy_(torch.as_tensor(v).to(dst.device).reshape(dst.shape))

    def grad_dict(self):
        """Dense gradients of the last train_step under the reference's parameter names."""
        out = {}
        for k, v in self.dense.g.items():
            if k == P + "gating.weight":
                for e in range(self.num_experts):
                    out[P + "gating.%d.weight" % e] = v[:, e:e + 1]
            elif k == P + "gating.bias":
                for e in range(self.num_experts):
                    out[P + "gating.%d.bias" % e] = v[e:e + 1]
            else:
                out[k] = v
        return out

    def _timed(self, name):
        return _Timed(self.timers, name)

    @staticmethod
    def _concat_ids(sparse_inputs):
        if isinstance(sparse_inputs, (list, tuple)):
            return torch.cat(list(sparse_inputs), dim=1).contiguous()       # net.py:93-94
        return sparse_inputs

    # ---------------------------------------------------------------- forward pieces
    def _feat(self, ids, dense_inputs):
        """net.py:93-108: lookup written straight into the head of the feature row, Linear(dense) into its tail."""
        B, S = ids.shape
        D, d = self.sparse_feature_dim, self.d
        feat = torch.empty(B, d, dtype=torch.float32, device=self.device)
        self.k.emb_gather(ids.reshape(-1), self.embedding, self.padding_idx, self.status, out=feat,
                          out_group=S, out_group_stride=d)
        self.k.gemm(dense_inputs, self.dense.p["dense_emb.weight"], self.ws, epilogue="bias",
                    bias=self.dense.p["dense_emb.bias"], out=feat[:, S * D:])
        return feat

    def _cross_v2(self, feat, out_last=None):
        """net.py:222-226: one rec_crossnet_v2_layer_fwd per layer.  Returns (x_L, xs, us)."""
        p, k = self.dense.p, self.k
        xs, us = [feat], []
        x = feat
        for i in range(self.cross_num):
            u = torch.empty_like(feat)
            last = i == self.cross_num - 1
            x = k.crossnet_v2_layer_fwd(feat, x, p[P + "cross_layers.%d.weight" % i],
                                        p[P + "cross_layers.%d.bias" % i], self.ws,
                                        out=out_last if (last and out_last is not None) else None, u=u)
            xs.append(x)
            us.append(u)
        return x, xs, us

    def _cross_mix(self, feat, out_last=None, saved=None):
        """net.py:278-320 (row-vector form): one rec_crossnet_mix_layer_fwd per layer.
        saved (list, optional): per layer (x_l, t1, t2, prob)."""
        p, k = self.dense.p, self.k
        x = feat
        for i in range(self.cross_num):
            last = i == self.cross_num - 1
            x_next, t1, t2, prob = k.crossnet_mix_layer_fwd(
                feat, x, p[P + "U_list.%d" % i], p[P + "V_list.%d" % i], p[P + "C_list.%d" % i],
                p[P + "bias.%d" % i].view(-1), p[P + "gating.weight"], p[P + "gating.bias"], self.ws,
                out=out_last if (last and out_last is not None) else None)
            if saved is not None:
                saved.append((x, t1, t2, prob))
            x = x_next
        return x

    def _dnn_params(self):
        n = len(self.layer_sizes)
        p, g = self.dense.p, self.dense.g
        W = [p["DNN_.linear_%d.weight" % i] for i in range(n)]
        b = [p["DNN_.linear_%d.bias" % i] for i in range(n)]
        dW = [g["DNN_.linear_%d.weight" % i] for i in range(n)]
        db = [g["DNN_.linear_%d.bias" % i] for i in range(n)]
        return W, b, dW, db

    def _drop(self):
        return self.dropout_rate > 0.0

    def _dnn_tower(self, x, W, b, out_last=None, step=None):
        """DNNLayer.forward (net.py:178-184): Linear(+bias) -> ReLU per layer.  Train mode (step given, dropout_rate > 0):
        both dropouts of a layer in ONE pass behind the GEMM's bias+ReLU epilogue (they commute with the ReLU); mask
        streams of layer i = (step * n + i) * 2 and + 1.  -> (output, acts): acts[i] = input of layer i, acts[n] = output."""
        k, n = self.k, len(W)
        if step is None or not self._drop():
            return k.mlp_forward(x, W, b, self.ws, relu_last=True, out_last=out_last)
        acts = []
        for i in range(n):
            acts.append(x)
            x = k.gemm(x, W[i], self.ws, epilogue="bias_relu", bias=b[i], out=out_last if i == n - 1 else None)
            st = (step * n + i) * 2
            k.dropout(x, self.dropout_rate, self.dropout_seed, st, st + 1, step_stride=2 * n)
        return x, acts + [x]

    def _tower_grad_through_dropout(self, g, layer, n, step):
        """d(loss)/d(output of DNN layer `layer`) after the dX GEMM's ReLU mask: the gradient also passes that layer's two
        dropouts — the same rec_dropout call (keep mask redundant with the mask of the dropped output, scale 1/(1-p)^2)."""
        st = (step * n + layer) * 2
        return self.k.dropout(g, self.dropout_rate, self.dropout_seed, st, st + 1, step_stride=2 * n)

    def _logit(self, ids, dense_inputs, keep=False, step=None):
        p, k = self.dense.p, self.k
        B = ids.shape[0]
        feat = self._feat(ids, dense_inputs)
        W, b, _, _ = self._dnn_params()
        n_out = self.layer_sizes[-1]
        saved = dict(feat=feat)
        if self.is_Stacked:
            if self.use_low_rank_mixture:
                saved["mix"] = []
                cross = self._cross_mix(feat, saved=saved["mix"] if keep else None)
            else:
                cross, saved["xs"], saved["us"] = self._cross_v2(feat)
            if step is not None and self._drop():
                h, acts = self._dnn_tower(cross, W, b, step=step)
                logit = k.gemm(h, p["fc.weight"], self.ws, epilogue="bias", bias=p["fc.bias"])
                acts = acts + [logit]
            else:
                logit, acts = k.mlp_forward(cross, W + [p["fc.weight"]], b + [p["fc.bias"]], self.ws)
            saved["acts"] = acts
        else:
            last = torch.empty(B, n_out + self.d, dtype=torch.float32, device=self.device)      # net.py:129
            if self.use_low_rank_mixture:
                saved["mix"] = []
                self._cross_mix(feat, out_last=last[:, n_out:], saved=saved["mix"] if keep else None)
            else:
                _, saved["xs"], saved["us"] = self._cross_v2(feat, out_last=last[:, n_out:])
            _, acts = self._dnn_tower(feat, W, b, out_last=last[:, :n_out], step=step)
            logit = k.gemm(last, p["fc.weight"], self.ws, epilogue="bias", bias=p["fc.bias"])
            saved["acts"], saved["last"] = acts, last
        return (logit, saved) if keep else (logit, None)

    def forward(self, sparse_inputs, dense_inputs):
        ids = self._concat_ids(sparse_inputs)
        logit, _ = self._logit(ids, dense_inputs)
        return torch.sigmoid(logit)                                                       # net.py:117,134

    __call__ = forward

    # ---------------------------------------------------------------- training step (CrossNetV2)
    def _ensure_sparse_state(self):
        if self.sparse_state is None:
            D = self.sparse_feature_dim
            Dp = _round_up(D, 4)
            mv = torch.zeros(self.rec.shape[0], _round_up(2 * Dp, 32), dtype=torch.float32, device=self.device)
            self.sparse_state = dict(mv=mv, m=mv[:, :D], v=mv[:, Dp:Dp + D])

    def train_step(self, sparse_inputs, dense_inputs, label, lr=1e-3, clip_norm=10.0, auc_stats=None,
                   dlogit=None):
        """dcn_v2/dygraph_model.py:103-127 train_forward + backward + Adam with ClipGradByGlobalNorm.
        dlogit ([B,1], optional): d loss / d logit supplied by the caller instead of the log-loss head
        (custom losses; the golden-gradient tests use d pred.sum()).  Returns (loss [1], pred [B,1])."""
        # launch-bound batches (the reference's own: dcn_v2 config_bigdata.yaml batch_size 512): from the third sight of
        # an input signature on the step is replayed from its recorded C-ABI call list (plan.py), the Adam step count and
        # the dropout mask streams re-derived per replay; bit-identical (tests/test_dcn_v2_gpu.py)
        if (self.device.type == "cuda" and self.k is _ops and torch.is_tensor(sparse_inputs)
                and torch.is_tensor(dense_inputs) and torch.is_tensor(label) and dlogit is None
                and self.timers is None and not self._recording and not self.use_low_rank_mixture
                and sparse_inputs.numel() <= int(os.environ.get("REC_STEP_PLAN_MAX", "65536"))
                and os.environ.get("REC_STEP_PLAN", "1") != "0"):
            from .plan import CallPlan
            inputs = [sparse_inputs, dense_inputs, label]
            key = (tuple(sparse_inputs.shape), float(clip_norm or 0.0),
                   None if auc_stats is None else (auc_stats[0].data_ptr(), auc_stats[1].data_ptr()),
                   self.dropout_rate, self.dropout_seed, self.l2_dnn)       # baked into the recorded calls
            entry = self._plans.get(key)
            if entry is None:
                self._plans[key] = "seen"
            elif entry == "seen" or not entry.matches(inputs):
                plan = CallPlan()
                self._recording = True
                try:
                    out = plan.record(lambda: self.train_step(sparse_inputs, dense_inputs, label, lr, clip_norm,
                                                              auc_stats), inputs, step0=self.step_count + 1)
                finally:
                    self._recording = False
                self._plans[key] = plan
                return out
            else:
                self.step_count += 1
                return entry.replay(inputs, self.step_count, float(lr))
        k, p, g = self.k, self.dense.p, self.dense.g
        ids = self._concat_ids(sparse_inputs)
        B, S = ids.shape
        D, d = self.sparse_feature_dim, self.d
        self._ensure_sparse_state()
        self.step_count += 1
        t = self.step_count
        if self._groups is None or self._groups.n != B * S:
            self._groups = k.IdGroups(B * S, self.device)
        groups = self._groups
        drop = self._drop()
        with self._timed("fwd"):
            logit, sv = self._logit(ids, dense_inputs, keep=True, step=t if drop else None)
        k.ids_group(ids, self.sparse_feature_number, self.padding_idx, self.ws_group, None, self.status, groups)
        pred, dz, loss = k.sigmoid_logloss(logit, None, None, label, self.ws)
        if dlogit is not None:
            dz = dlogit
        if auc_stats is not None:
            k.auc_histogram(pred, label, auc_stats[0], auc_stats[1], NUM_THRESHOLDS)
        W, b, dW, db = self._dnn_params()
        n_out = self.layer_sizes[-1]
        with self._timed("bwd"):
            n_dnn = len(W)

            def tower_backward(gy, acts):
                """d(input of the tower) from gy = d(output of its last layer) already masked by that layer's ReLU
                and passed through its dropouts; dW / db of every layer on the way (train mode: net.py:181-183)."""
                for i in reversed(range(n_dnn)):
                    k.gemm(acts[i], gy, self.ws, trans_a=True, out=dW[i], b_colsum=db[i])
                    if i > 0:
                        gy = k.gemm(gy, W[i], self.ws, trans_b=True, epilogue="relu_mask", aux0=acts[i])
                        gy = self._tower_grad_through_dropout(gy, i - 1, n_dnn, t)
                    else:
                        gy = k.gemm(gy, W[0], self.ws, trans_b=True)
                return gy
            if self.is_Stacked and not drop:
                dcross = k.mlp_backward(dz, sv["acts"], W + [p["fc.weight"]], dW + [g["fc.weight"]],
                                        db + [g["fc.bias"]], self.ws)
                dx0_acc, have_acc = torch.empty_like(sv["feat"]), False
            elif self.is_Stacked:
                acts = sv["acts"]                                  # [cross, y_0 .. y_{n-1}, logit]
                k.gemm(acts[n_dnn], dz, self.ws, trans_a=True, out=g["fc.weight"], b_colsum=g["fc.bias"])
                gy = k.gemm(dz, p["fc.weight"], self.ws, trans_b=True, epilogue="relu_mask", aux0=acts[n_dnn])
                gy = self._tower_grad_through_dropout(gy, n_dnn - 1, n_dnn, t)
                dcross = tower_backward(gy, acts)
                dx0_acc, have_acc = torch.empty_like(sv["feat"]), False
            else:
                last, fcw = sv["last"], p["fc.weight"]
                k.gemm(last, dz, self.ws, trans_a=True, out=g["fc.weight"], b_colsum=g["fc.bias"])
                ddnn = k.gemm(dz, fcw[:n_out], self.ws, trans_b=True, epilogue="relu_mask", aux0=last[:, :n_out])
                dcross = k.gemm(dz, fcw[n_out:], self.ws, trans_b=True)
                if drop:
                    ddnn = self._tower_grad_through_dropout(ddnn, n_dnn - 1, n_dnn, t)
                    dx0_acc, have_acc = tower_backward(ddnn, sv["acts"]), True
                else:
                    dx0_acc, have_acc = k.mlp_backward(ddnn, sv["acts"], W, dW, db, self.ws), True   # d feat via DNN
            feat = sv["feat"]
            if self.use_low_rank_mixture:
                dx = self._cross_mix_backward(dcross, feat, sv["mix"], dx0_acc, have_acc)
            else:
                dx = dcross
                xs, us = sv["xs"], sv["us"]
                for i in reversed(range(self.cross_num)):       # one rec_crossnet_v2_layer_bwd per layer
                    dx = k.crossnet_v2_layer_bwd(feat, xs[i], p[P + "cross_layers.%d.weight" % i], us[i], dx, dx0_acc,
                                                 have_acc, i == 0, g[P + "cross_layers.%d.weight" % i],
                                                 g[P + "cross_layers.%d.bias" % i], self.ws)
                    have_acc = True
            dfeat = dx                                           # d loss / d feat_embeddings  [B,d]
            k.gemm(dense_inputs, dfeat[:, S * D:], self.ws, trans_a=True, out=g["dense_emb.weight"],
                   b_colsum=g["dense_emb.bias"])
        with self._timed("optimizer"):
            scale = None
            pp = self._pp = k.segment_partials(groups, dfeat, D, grad_group=S, grad_group_stride=d,
                                               out=getattr(self, "_pp", None))   # hot rows, shared by both consumers
            if clip_norm:
                ss = self._scalar("sumsq")
                k.sumsq(self.dense.grad, ss, self.ws)
                k.sparse_rows_sumsq(groups, dfeat, D, ss, self.ws, accumulate=True, grad_group=S,
                                    grad_group_stride=d, partials=pp)
                scale = k.clip_scale(ss, clip_norm, self._scalar("scale"))
            if self.l2_dnn > 0.0:       # L2Decay on the DNN weights, appended AFTER the clip (net.py:164-170; [EXT] order)
                for i in range(len(W)):
                    k.l2_decay_grad(dW[i].reshape(-1), W[i].reshape(-1), self.l2_dnn, scale)
            k.adam_dense(self.dense.data, self.dense.m, self.dense.v, self.dense.grad, t, lr, grad_scale=scale)
            st = self.sparse_state
            k.sparse_adam_rows(groups, dfeat, 1, self.embedding, st["m"], st["v"], t, lr, grad_group=S,
                               grad_group_stride=d, grad_scale=scale, partials=pp)
        self._last_dfeat = dfeat
        return loss, pred

    def _cross_mix_backward(self, dout, feat, saved, dx0_acc, have_acc):
        """Backward of _cross_mix (oracle: oracle/dcn_v2_ref.py cross_mix_backward): one rec_crossnet_mix_layer_bwd
        per layer; the gating Linear layers are shared by all cross layers (net.py:267-268), so their gradients are
        written by the first layer processed and accumulated by the others.  Returns d feat."""
        p, g, k = self.dense.p, self.dense.g, self.k
        dx = dout
        for i in reversed(range(self.cross_num)):
            xl, t1, t2, prob = saved[i]
            dx = k.crossnet_mix_layer_bwd(
                feat, xl, p[P + "U_list.%d" % i], p[P + "V_list.%d" % i], p[P + "C_list.%d" % i],
                p[P + "bias.%d" % i].view(-1), p[P + "gating.weight"], t1, t2, prob, dx, dx0_acc, have_acc, i == 0,
                g[P + "U_list.%d" % i], g[P + "V_list.%d" % i], g[P + "C_list.%d" % i], g[P + "bias.%d" % i].view(-1),
                g[P + "gating.weight"], g[P + "gating.bias"], i != self.cross_num - 1, self.ws)
            have_acc = True
        return dx

    # ---------------------------------------------------------------- the step through ONE C-ABI call
    def c_net(self, clip_norm=10.0):
        """rec_dcn_v2_net over this layer's tensors (include/recengine.h)."""
        from . import _lib
        if len(self.layer_sizes) > _lib.DCN_MAX_LAYERS or self.cross_num > _lib.DCN_MAX_LAYERS:
            raise ValueError("rec_dcn_v2_train_step takes at most %d cross layers / DNN layers" % _lib.DCN_MAX_LAYERS)
        self._ensure_sparse_state()
        p, g, st = self.dense.p, self.dense.g, self.sparse_state
        net = _lib.DcnV2Net()
        net.num_slots, net.dim, net.dense_dim = self.sparse_num_field, self.sparse_feature_dim, self.dense_feature_dim
        net.num_rows, net.padding_idx = self.sparse_feature_number, self.padding_idx
        net.emb_stride, net.state_stride = self.embedding.stride(0), st["m"].stride(0)
        net.emb, net.emb_m, net.emb_v = self.embedding.data_ptr(), st["m"].data_ptr(), st["v"].data_ptr()
        net.cross_num, net.n_dnn = self.cross_num, len(self.layer_sizes)
        for i, w in enumerate(self.layer_sizes):
            net.widths[i] = w
        net.is_stacked, net.low_rank_mix = int(self.is_Stacked), int(self.use_low_rank_mixture)
        net.num_experts, net.low_rank = self.num_experts, self.low_rank
        net.dropout_rate, net.dropout_seed, net.l2_dnn = self.dropout_rate, self.dropout_seed, self.l2_dnn
        net.clip_norm = float(clip_norm or 0.0)

        def both(field, name, i=None):
            for prefix, src in (("", p), ("g_", g)):
                if i is None:
                    setattr(net, prefix + field, src[name].data_ptr())
                else:
                    getattr(net, prefix + field)[i] = src[name].data_ptr()
        both("dense_emb_w", "dense_emb.weight")
        both("dense_emb_b", "dense_emb.bias")
        for i in range(self.cross_num):
            if self.use_low_rank_mixture:
                for field, nm in (("mix_u", "U_list"), ("mix_v", "V_list"), ("mix_c", "C_list"), ("mix_bias", "bias")):
                    both(field, P + "%s.%d" % (nm, i), i)
            else:
                both("cross_w", P + "cross_layers.%d.weight" % i, i)
                both("cross_b", P + "cross_layers.%d.bias" % i, i)
        if self.use_low_rank_mixture:
            both("gate_w", P + "gating.weight")
            both("gate_b", P + "gating.bias")
        for i in range(len(self.layer_sizes)):
            both("dnn_w", "DNN_.linear_%d.weight" % i, i)
            both("dnn_b", "DNN_.linear_%d.bias" % i, i)
        both("fc_w", "fc.weight")
        both("fc_b", "fc.bias")
        net.flat_param, net.flat_grad = self.dense.data.data_ptr(), self.dense.grad.data_ptr()
        net.flat_m, net.flat_v, net.flat_numel = self.dense.m.data_ptr(), self.dense.v.data_ptr(), self.dense.data.numel()
        return net

    def train_step_c(self, sparse_inputs, dense_inputs, label, lr=1e-3, clip_norm=10.0, auc_stats=None):
        """train_step through rec_dcn_v2_train_step: ONE foreign call per step — what a non-Python binder of
        include/recengine.h gets (csrc/dcn_v2_step.hip).  Same entry points, arguments and order as train_step:
        bit-identical (tests/test_dcn_v2_step_c.py)."""
        ids = self._concat_ids(sparse_inputs)
        key = float(clip_norm or 0.0)
        cached = getattr(self, "_c_net", None)
        if cached is None or cached[0] != key:
            cached = self._c_net = (key, self.c_net(clip_norm))
        if getattr(self, "_ws_c", None) is None:
            self._ws_c = self.k.Workspace(self.device)
        self.step_count += 1
        return self.k.dcn_v2_train_step(cached[1], ids, dense_inputs, label.reshape(-1), self.step_count, lr, self._ws_c,
                                        auc_stats=auc_stats, num_thresholds=NUM_THRESHOLDS, status=self.status)

    def _scalar(self, name):
        b = getattr(self, "_s_" + name, None)
        if b is None:
            b = torch.zeros(1, dtype=torch.float32, device=self.device)
            setattr(self, "_s_" + name, b)
        return b


class DygraphModel:
    """dcn_v2/dygraph_model.py:24-140 — same method names; tensors are torch device tensors."""

    def create_model(self, config, device="cuda", kernels=None):
        g = config.get
        return DCN_V2Layer(g("hyper_parameters.sparse_feature_number"), g("hyper_parameters.sparse_feature_dim"),
                           g("hyper_parameters.dense_input_dim"), g("hyper_parameters.sparse_inputs_slots") - 1,
                           g("hyper_parameters.fc_sizes"), g("hyper_parameters.cross_num"),
                           g("hyper_parameters.is_Stacked", None), g("hyper_parameters.use_low_rank_mixture", None),
                           g("hyper_parameters.low_rank", 32), g("hyper_parameters.num_experts", 4), device=device,
                           kernels=kernels,
                           # the reference's train-mode graph: Dropout(0.5) in the DNN tower (net.py:146,158), L2Decay(1e-7)
                           # on its weights (net.py:165); forward() / infer stay eval mode
                           dropout_rate=g("hyper_parameters.dropout_rate", 0.5), l2_dnn=1e-7,
                           dropout_seed=g("runner.seed", 12345))

    def create_feeds(self, batch_data, config, device="cuda"):
        return slot_feeds(batch_data, config, device)

    def create_metrics(self, device="cuda"):
        return auc_metrics(device)

    def train_forward(self, dy_model, metrics_list, batch_data, config):
        label, sparse, dense = self.create_feeds(batch_data, config, dy_model.device)
        lr = config.get("hyper_parameters.optimizer.learning_rate", 0.001)
        clip = config.get("hyper_parameters.optimizer.clip_by_norm", 10.0)        # dygraph_model.py:83-85
        loss, _ = dy_model.train_step(sparse, dense, label, lr, clip, metrics_list[0] if metrics_list else None)
        return loss, metrics_list, {"log_loss": loss}

    def infer_forward(self, dy_model, metrics_list, batch_data, config):
        label, sparse, dense = self.create_feeds(batch_data, config, dy_model.device)
        pred = dy_model.forward(sparse, dense)
        if metrics_list:
            dy_model.k.auc_histogram(pred.contiguous(), label.contiguous(), metrics_list[0][0], metrics_list[0][1],
                                     NUM_THRESHOLDS)
        return metrics_list, None
