"""rank/xdeepfm on the engine (the CIN sibling of DeepFM).

Host mirror of /root/reference/models/rank/xdeepfm/net.py (`xDeepFMLayer` :23-55, `Linear` :58-124, `CIN` :127-202,
`DNN` :205-242) and xdeepfm/dygraph_model.py (`DygraphModel`).

    y_linear, feat = Linear(ids, dense)      net.py:104-124  == DeepFM's first-order term and feat_embeddings:
                                             rec_deepfm_fm_fwd with padding_idx = none (its y2 is ignored)
    y_cin  = CIN(feat)                       net.py:155-202  per layer  Z = X0 (x) Xk  (rec_cin_outer_fwd, batch chunks)
                                             XT_{k+1} = Z @ Wc^T (rec_gemm_f32, M = B*D)   pooled += sum_d (rec_cin_sumpool)
                                             y_cin = pooled @ cnn_fc.weight + cnn_fc.bias
    y_dnn  = DNN(feat.reshape(B, F*D))       net.py:235-242  Linear+ReLU ... Linear(1)  (rec_gemm_f32 epilogues)
    pred   = sigmoid(y_linear + bias + y_cin + y_dnn)        net.py:54
Backward is the explicit chain (CIN: dZ = dXT @ Wc, dWc = dXT^T @ Z with Z recomputed per chunk, rec_cin_outer_bwd into
d feat), L2Decay(1e-4) on the CIN / cnn_fc / DNN weights (net.py:139,150,219) added to their gradients, then the
reference's optimizer: Adam(lazy_mode=True) (dygraph_model.py:60-64) — lazy rows fused per record, dense flat Adam.
Parameter keys follow the reference's sublayer names; `cin.cin_linear.*` and `cin.cnn_fc.*` are the same tensors (the
reference registers that Linear twice, net.py:146-153).
"""
import math
import os

import torch

from .deepfm import NUM_THRESHOLDS, DeepFMLayer, _OnSide, _round_up, auc_metrics, slot_feeds

L2_COEFF = 1e-4            # net.py:139,150,219
# bytes of the outer-product scratch (Z / Y) per batch chunk.  Small enough to stay in the 256 MB Infinity Cache between
# the kernel that writes it and the GEMM that reads it — and to be overwritten there by the next chunk before it is ever
# written back: REC_CIN_CHUNK_MB (default measured in profiles/r03_xdeepfm_chunk.txt)
Z_CHUNK_BYTES = int(os.environ.get("REC_CIN_CHUNK_MB", "4096")) << 20


class xDeepFMLayer(DeepFMLayer):
    """xdeepfm/net.py:23-55.  forward(sparse_inputs, dense_inputs) -> predict [B,1]."""

    supports_padded_feat = False     # the CIN reads feat as [B, fields, D]: dense layout

    def __init__(self, sparse_feature_number, sparse_feature_dim, dense_feature_dim, sparse_num_field,
                 layer_sizes_cin, layer_sizes_dnn, device="cuda", kernels=None):
        F = dense_feature_dim + sparse_num_field
        self.layer_sizes_cin = list(layer_sizes_cin)
        extra, last = [], F
        for i, c in enumerate(self.layer_sizes_cin):
            extra.append(("cin.cnn_%d.weight" % i, (c, last * F, 1, 1)))          # Conv2D weight [out, in, 1, 1]
            last = c
        self.cin_total = sum(self.layer_sizes_cin)
        self._pad_bufs = {}
        extra += [("cin.cnn_fc.weight", (self.cin_total, 1)), ("cin.cnn_fc.bias", (1,))]
        super().__init__(sparse_feature_number, sparse_feature_dim, dense_feature_dim, sparse_num_field,
                         layer_sizes_dnn, device=device, zero_padding_row=False, kernels=kernels, extra_dense=extra)
        self.fm.padding_idx = None                      # net.py:75-93: the Embeddings have no padding_idx
        self.compact = False                            # the CIN reads every field of feat_embeddings, dense ones too
        self.fp = self.num_field
        p = self.dense.p
        p["fm.dense_w_one"].fill_(1.0)                  # net.py:96-104 Constant(1.0)
        p["fm.dense_w"].fill_(1.0)
        D = sparse_feature_dim
        sizes = [D * F] + self.layer_sizes + [1]
        for i in range(self.n_linear):                  # net.py:216-221 Normal(std = 0.1 / sqrt(in))
            p["dnn.linear_%d.weight" % i].normal_(0.0, 0.1 / math.sqrt(sizes[i]))
        last = F
        for i, c in enumerate(self.layer_sizes_cin):    # net.py:137-142 Normal(std = 1 / sqrt(last_s * num_field))
            p["cin.cnn_%d.weight" % i].normal_(0.0, 1.0 / math.sqrt(last * F))
            last = c
        p["cin.cnn_fc.weight"].normal_(0.0, 0.1 / math.sqrt(self.cin_total))      # net.py:147-153
        self.cin_w = [p["cin.cnn_%d.weight" % i].view(c, -1) for i, c in enumerate(self.layer_sizes_cin)]   # [C, F*S]
        self.cin_dw = [self.dense.g["cin.cnn_%d.weight" % i].view(c, -1) for i, c in enumerate(self.layer_sizes_cin)]
        self._decayed = ["cin.cnn_%d.weight" % i for i in range(len(self.layer_sizes_cin))] + ["cin.cnn_fc.weight"] + \
                        ["dnn.linear_%d.weight" % i for i in range(self.n_linear)]
        self._zbuf = None
        self._keep_bufs, self._kept = {}, {}
        self._bias_sum = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._zeros = None

    def state_dict(self):
        sd = super().state_dict()
        sd["cin.cin_linear.weight"], sd["cin.cin_linear.bias"] = sd["cin.cnn_fc.weight"], sd["cin.cnn_fc.bias"]
        return sd

    # -- CIN ----------------------------------------------------------------------------------------
    def _chunks(self, B, K):
        D = self.sparse_feature_dim
        per = max(1, Z_CHUNK_BYTES // (4 * D * K))
        return [(b0, min(B, b0 + per)) for b0 in range(0, B, per)]

    @staticmethod
    def _use_y(i, S, Cn):
        """Which association a CIN layer takes: Y = Xk W'^T (C*F columns) when it is smaller than Z (F*S columns); layer 0
        always takes Z (its Xk is feat_embeddings, not a d-major matrix)."""
        return i > 0 and Cn < S

    def _z(self, rows, K):
        """Grow-only scratch for the outer-product rows of one batch chunk (Z forward, dZ backward)."""
        if self._zbuf is None or self._zbuf.numel() < rows * K:
            self._zbuf = torch.empty(rows * K, dtype=torch.float32, device=self.device)
        return self._zbuf[: rows * K].view(rows, K)

    def _keep_buf(self, name, rows, K):
        """Outer-product rows (Z) / the Y intermediate of a layer, KEPT from the forward for the backward when the whole
        batch is one chunk: a few GB at B 65536 (Z 3.6 GB, Y 2.9 GB of 288) instead of recomputing them in the backward —
        0.8 ms of HBM passes and a 1.7 ms GEMM per step (REC_CIN_KEEP=0: recompute, the round-3 behaviour)."""
        b = self._keep_bufs.get(name)
        if b is None or b.numel() < rows * K:
            b = self._keep_bufs[name] = torch.empty(rows * K, dtype=torch.float32, device=self.device)
        return b[: rows * K].view(rows, K)

    def _scratch(self, name, shape):
        b = self._pad_bufs.get(name)
        if b is None or tuple(b.shape) != tuple(shape):
            b = self._pad_bufs[name] = torch.zeros(*shape, dtype=torch.float32, device=self.device)
        return b

    def _padded_weight(self, i, K):
        """(K rounded up to the GEMM's k-step, the conv weight of layer i with zero columns behind K): F*S = 39*39 = 1521
        is odd, which keeps the layer-1 GEMMs (K or N = 1521, M = B*D) off the aligned tile loaders — 72 TF at K 1521
        against 110+ at K 1536 (profiles/r03_xdeepfm.txt).  The zero columns meet zero columns of Z."""
        KP = (K + 15) // 16 * 16
        if KP == K or os.environ.get("REC_CIN_PAD", "1") == "0":
            return K, self.cin_w[i]
        w = self._scratch("cin_wz%d" % i, (self.cin_w[i].shape[0], KP))
        w[:, :K].copy_(self.cin_w[i])
        return KP, w

    def _z_padded(self, rows, K, KP):
        """-> (Z scratch [rows, KP], its [rows, K] view the outer-product kernels write); pad columns zeroed."""
        zf = self._z(rows, KP)
        if KP != K:
            zf[:, K:].zero_()
        return zf, zf[:, :K]

    def _layer_inputs(self, feat, xts, i):
        """(tensor, view maker) of X_i: feat_embeddings for layer 0, the previous layer's d-major output after."""
        k, D = self.k, self.sparse_feature_dim
        if i == 0:
            return feat, (lambda t: k.cin_view(t, "bfd"))
        return xts[i - 1], (lambda t: k.cin_view((t, D), "xt"))

    def _cin_forward(self, feat, keep=False):
        """-> (pooled [B, sum C], [XT_1 .. XT_L])   XT_k [B*D, C_k] d-major.  keep: a training forward — the Z / Y
        intermediates of one-chunk layers stay in their own buffers for _cin_backward (self._kept)."""
        k, D = self.k, self.sparse_feature_dim
        B, F, _ = feat.shape
        keep = keep and os.environ.get("REC_CIN_KEEP", "1") != "0"
        self._kept = {}
        pooled = torch.empty(B, self.cin_total, dtype=torch.float32, device=self.device)
        xts, off, S = [], 0, F
        for i, Cn in enumerate(self.layer_sizes_cin):
            xk, mk = self._layer_inputs(feat, xts, i)
            xt = torch.empty(B * D, Cn, dtype=torch.float32, device=self.device)
            if self._use_y(i, S, Cn):       # C < S: Y = Xk @ W'^T [., C*F] is smaller than Z [., F*S] and better shaped
                chunks = self._chunks(B, F * Cn)
                for b0, b1 in chunks:
                    n = b1 - b0
                    if keep and len(chunks) == 1:
                        y = self._keep_buf("y%d" % i, n * D, Cn * F)
                        self._kept[i] = ("y", B, y)
                    else:
                        y = self._z(n * D, Cn * F)
                    k.gemm(xk[b0 * D:b1 * D], self.cin_w[i].view(Cn * F, S), self.ws, trans_b=True, out=y)
                    k.cin_contract_fwd(n, D, F, y, feat[b0:b1], k.cin_view(feat, "bfd"), xt[b0 * D:b1 * D])
            KP, wz = self._padded_weight(i, F * S)
            zchunks = [] if self._use_y(i, S, Cn) else self._chunks(B, KP)
            for b0, b1 in zchunks:
                n = b1 - b0
                if keep and len(zchunks) == 1:
                    zf = self._keep_buf("z%d" % i, n * D, KP)
                    if KP != F * S:
                        zf[:, F * S:].zero_()
                    z = zf[:, : F * S]
                    self._kept[i] = ("z", B, zf)
                else:
                    zf, z = self._z_padded(n * D, F * S, KP)
                xk_c = xk[b0:b1] if i == 0 else xk[b0 * D:b1 * D]
                k.cin_outer_fwd(n, D, F, S, feat[b0:b1], k.cin_view(feat, "bfd"), xk_c, mk(xk_c), z)
                k.gemm(zf, wz, self.ws, trans_b=True, out=xt[b0 * D:b1 * D])                   # net.py:190 (1x1 conv)
            k.cin_sumpool(B, D, xt, pooled[:, off:off + Cn])                                   # net.py:195-198
            xts.append(xt)
            off, S = off + Cn, Cn
        return pooled, xts

    def _cin_backward(self, feat, xts, dpooled, dfeat):
        """Accumulates the CIN's gradient w.r.t. feat_embeddings into dfeat [B,F,D]; writes the conv weight gradients."""
        k, D = self.k, self.sparse_feature_dim
        B, F, _ = feat.shape
        L = len(self.layer_sizes_cin)
        offs = [sum(self.layer_sizes_cin[:i]) for i in range(L)]
        dxt = torch.empty(B * D, self.layer_sizes_cin[-1], dtype=torch.float32, device=self.device)
        k.cin_sumpool_bwd(B, D, dpooled[:, offs[-1]:offs[-1] + self.layer_sizes_cin[-1]], dxt)
        for i in reversed(range(L)):
            Cn = self.layer_sizes_cin[i]
            S = F if i == 0 else self.layer_sizes_cin[i - 1]
            xk, mk = self._layer_inputs(feat, xts, i)
            dxk = dfeat if i == 0 else torch.empty(B * D, S, dtype=torch.float32, device=self.device)
            use_y = self._use_y(i, S, Cn)
            if use_y:
                w2 = self.cin_w[i].view(Cn * F, S)
                dw2 = self.cin_dw[i].view(Cn * F, S)
                k.cin_sumpool_bwd(B, D, dpooled[:, offs[i - 1]:offs[i - 1] + S], dxk)       # the pooled-feature gradient
                kept = self._kept.get(i)
                kept = kept[2] if kept is not None and kept[0] == "y" and kept[1] == B else None
                for ci, (b0, b1) in enumerate(self._chunks(B, (1 if kept is not None else 2) * F * Cn)):
                    n = b1 - b0
                    xk_c, g = xk[b0 * D:b1 * D], dxt[b0 * D:b1 * D]
                    if kept is not None:                                                      # Y kept by the forward
                        y, dy = kept[b0 * D:b1 * D], self._z(n * D, Cn * F)
                    else:
                        both = self._z(2 * n * D, Cn * F)
                        y, dy = both[: n * D], both[n * D:]
                        k.gemm(xk_c, w2, self.ws, trans_b=True, out=y)                        # Y recomputed
                    k.cin_contract_bwd(n, D, F, y, g, feat[b0:b1], k.cin_view(feat, "bfd"), dy, dfeat[b0:b1],
                                       k.cin_view(dfeat, "bfd"), True)
                    k.gemm(dy, xk_c, self.ws, trans_a=True, out=dw2,
                           **(dict(epilogue="add", aux1=dw2) if ci > 0 else {}))              # dW' = dY^T Xk
                    dxk_c = dxk[b0 * D:b1 * D]
                    k.gemm(dy, w2, self.ws, epilogue="add", aux1=dxk_c, out=dxk_c)            # dXk = dY W' + pooled grad
            KP, wz = (F * S, None) if use_y else self._padded_weight(i, F * S)
            dwz = self.cin_dw[i] if KP == F * S else self._scratch("cin_dwz%d" % i, (Cn, KP))
            keptz = self._kept.get(i)
            keptz = keptz[2] if keptz is not None and keptz[0] == "z" and keptz[1] == B else None
            for ci, (b0, b1) in enumerate([] if use_y else self._chunks(B, KP)):
                n = b1 - b0
                xk_c = xk[b0:b1] if i == 0 else xk[b0 * D:b1 * D]
                dxk_c = dxk[b0:b1] if i == 0 else dxk[b0 * D:b1 * D]
                g = dxt[b0 * D:b1 * D]
                if keptz is not None:                                                                     # Z kept by the forward
                    zf, z = keptz, keptz[:, : F * S]
                else:
                    zf, z = self._z_padded(n * D, F * S, KP)
                    k.cin_outer_fwd(n, D, F, S, feat[b0:b1], k.cin_view(feat, "bfd"), xk_c, mk(xk_c), z)  # recomputed
                k.gemm(g, zf, self.ws, trans_a=True, out=dwz,
                       **(dict(epilogue="add", aux1=dwz) if ci > 0 else {}))                              # dWc
                k.gemm(g, wz, self.ws, out=zf)                                                            # dZ (in place of Z)
                dpool = None if i == 0 else dpooled[b0:b1, offs[i - 1]:offs[i - 1] + S]
                k.cin_outer_bwd(n, D, F, S, z, feat[b0:b1], k.cin_view(feat, "bfd"), xk_c, mk(xk_c),
                                dfeat[b0:b1], k.cin_view(dfeat, "bfd"), True, dxk_c, mk(dxk_c), i == 0, dpool)
            if not use_y and KP != F * S:
                self.cin_dw[i].copy_(dwz[:, : F * S])
            dxt = dxk

    def _logit_parts(self, ids, dense_inputs, keep=None):
        k = self.k
        y1, _, feat, sum_emb, _ = self._fm_fwd(ids, dense_inputs)
        B = feat.shape[0]
        pooled, xts = self._cin_forward(feat, keep=keep is not None)
        p = self.dense.p
        self._bias_sum.copy_(p["cin.cnn_fc.bias"])
        k.sgd_dense(self._bias_sum, p["bias"], -1.0)                      # cnn_fc.bias + bias (net.py:54)
        y_cin = k.gemm(pooled, p["cin.cnn_fc.weight"], self.ws, epilogue="bias", bias=self._bias_sum)
        y_dnn, acts = k.mlp_forward(feat.view(B, -1), self.mlp_w, self.mlp_b, self.ws_mlp)
        if keep is not None:
            keep.update(feat=feat, sum_emb=sum_emb, pooled=pooled, xts=xts, acts=acts)
        return y1, y_cin, y_dnn

    def forward(self, sparse_inputs, dense_inputs):
        y1, y_cin, y_dnn = self._logit_parts(self._concat_ids(sparse_inputs), dense_inputs)
        return torch.sigmoid(y1 + y_cin + y_dnn)

    __call__ = forward

    def train_step(self, sparse_inputs, dense_inputs, label, lr=1e-3, auc_stats=None, allreduce=None):
        """xdeepfm/dygraph_model.py:77-90 train_forward + tools/trainer.py:151-152 backward / step.
        label [B,1] int64.  Returns (loss [1] device tensor, pred [B,1])."""
        k = self.k
        ids = self._concat_ids(sparse_inputs)
        B, S = ids.shape
        D, Dn = self.sparse_feature_dim, self.dense_feature_dim
        self._ensure_sparse_state()
        self.step_count += 1
        on_gpu = self.device.type == "cuda"
        cur = torch.cuda.current_stream() if on_gpu else None
        if on_gpu and self._side is None:
            self._side = k.concurrent_stream(self.device)
        side = self._side if on_gpu else None
        groups = getattr(self, "_groups", None)
        if groups is None or groups.n != B * S:
            groups = self._groups = k.IdGroups(B * S, self.device)
        sv = {}
        y1, y_cin, y_dnn = self._logit_parts(ids, dense_inputs, keep=sv)
        with _OnSide(side, cur):                                   # merge keys depend on the ids only
            k.ids_group(ids, self.sparse_feature_number, None, self.ws_group, None, self.status, groups)
        pred, dz, loss = k.sigmoid_logloss(y1, y_cin, y_dnn, label, self.ws)
        if auc_stats is not None:
            k.auc_histogram(pred, label, auc_stats[0], auc_stats[1], NUM_THRESHOLDS)
        p, g = self.dense.p, self.dense.g
        # heads: y_cin = pooled @ w_fc + (b_fc + bias)
        k.gemm(sv["pooled"], dz, self.ws, trans_a=True, out=g["cin.cnn_fc.weight"], b_colsum=g["cin.cnn_fc.bias"])
        g["bias"].copy_(g["cin.cnn_fc.bias"])
        dpooled = k.gemm(dz, p["cin.cnn_fc.weight"], self.ws, trans_b=True)
        d_flat = k.mlp_backward(dz, sv["acts"], self.mlp_w, self.mlp_dw, self.mlp_db, self.ws_mlp)   # [B, F*D]
        dfeat = d_flat.view(B, self.num_field, D)
        self._cin_backward(sv["feat"], sv["xts"], dpooled, dfeat)
        if self._zeros is None or self._zeros.shape[0] != B:
            self._zeros = torch.zeros(B, 1, dtype=torch.float32, device=self.device)
        row_grad, _, _ = k.deepfm_fm_bwd(dense_inputs, sv["feat"], sv["sum_emb"], dfeat, dz, self._zeros, S, self.ws,
                                         out=(self._row_grad_buf(B * S), g["fm.dense_w"].view(Dn, -1),
                                              g["fm.dense_w_one"]),
                                         dense_w=p["fm.dense_w"], compact=False)
        for name in self._decayed:                                  # L2Decay: grad += coeff * w
            k.sgd_dense(g[name], p[name], -L2_COEFF)
        t, st = self.step_count, self.sparse_state
        with _OnSide(side, cur):
            pp = self._pp = k.segment_partials(groups, row_grad, D, out=getattr(self, "_pp", None))
            pp1 = self._pp1 = k.segment_partials(groups, dz, 1, grad_div=S, out=getattr(self, "_pp1", None))
            k.sparse_adam_record(groups, row_grad, dz, S, self.fm.rec, st["mv"], D, t, lr,
                                 v_offset=_round_up(D, 4), partials=pp, partials1=pp1)
        if allreduce is not None:
            allreduce(self.dense.grad)
        k.adam_dense(self.dense.data, self.dense.m, self.dense.v, self.dense.grad, t, lr)
        if side is not None:
            cur.wait_stream(self._side)
        return loss, pred


class DygraphModel:
    """xdeepfm/dygraph_model.py:23-104 — same method names; tensors are torch device tensors."""

    def create_model(self, config, device="cuda", kernels=None):
        g = config.get
        return xDeepFMLayer(g("hyper_parameters.sparse_feature_number"), g("hyper_parameters.sparse_feature_dim"),
                            g("hyper_parameters.dense_input_dim"), g("hyper_parameters.sparse_inputs_slots") - 1,
                            g("hyper_parameters.layer_sizes_cin"), g("hyper_parameters.layer_sizes_dnn"),
                            device=device, kernels=kernels)

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
