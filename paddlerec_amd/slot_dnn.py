"""rank/slot_dnn on the engine — the multi-slot sum-pool net the PS / gpubox benchmarks train.

Host mirror of /root/reference/models/rank/slot_dnn/net.py (`BenchmarkDNNLayer`) and static_model.py:
    for every slot: sparse_embedding(padding_idx=0, ONE shared table, entry=ShowClickEntry)   net.py:61-69
                    sequence_pool('sum')                                                       net.py:73
    y = concat(bows, axis=1)  ->  Linear+ReLU ... Linear(1)                                   net.py:77-82
    predict = sigmoid(clip(y, -15, 15))                                                        net.py:84
    cost = mean(log_loss(predict, click))                                                      static_model.py:104-108
All 408 slots of a batch go through ONE launch of rec_multislot_sumpool_fwd (slot-major CSR straight from the host
parser, uint64 feasigns hashed to table rows on the device); the backward reads every id's gradient row in place
from d y (rec_grad_layout.index) — no [nnz, D] tensors in either direction.  Sparse optimizer:
    "adam"  lazy Adam on the touched rows (static_model.py:110-112 `Adam(lazy_mode=True)`)
    "ps"    the table accessor of the PS / gpubox mode (config_online.yaml:57-89): AdaGrad rule, show/click counters
            fed from the show / click inputs (ShowClickEntry), embedx_threshold, lazy feature creation
Parameter keys: embedding (the shared table), linear_i.{weight,bias}.
"""
import math

import os

import torch

from . import ops
from .deepfm import NUM_THRESHOLDS, _FlatParams, _OnSide, _round_up, auc_metrics

CLIP = (-15.0, 15.0)      # net.py:84


class BenchmarkDNNLayer:
    """net.py:21-85.  forward(batch) -> predict [B,1]; batch = ops.MultislotBatch (values | lod [S,B+1] | slot_base)."""

    def __init__(self, dict_dim, emb_dim, slot_num, layer_sizes, device="cuda", kernels=None, sparse_optimizer="adam",
                 key_mode=1, accessor=None, scale_sparse_grad=True):
        """key_mode 1: batch values are uint64 feasign bit patterns (what queuedataset_reader.py feeds), hashed into
        dict_dim rows on the device; 0: values are rows.  accessor: kwargs of ops.PsTable for sparse_optimizer='ps'.
        scale_sparse_grad: the push carries the gradient of the SUMMED loss (mean-loss gradient x batch size), as
        Paddle's PS trainers do (scale_sparse_gradient_with_batch_size, default true [EXT])."""
        self.scale_sparse_grad = bool(scale_sparse_grad)
        self.device = torch.device(device)
        self.k = kernels if kernels is not None else ops
        self.dict_dim, self.emb_dim, self.slot_num = int(dict_dim), int(emb_dim), int(slot_num)
        self.layer_sizes = list(layer_sizes)
        self.key_mode = int(key_mode)
        self.sparse_optimizer = sparse_optimizer
        N, D = self.dict_dim, self.emb_dim
        if sparse_optimizer == "ps":
            self.table = self.k.PsTable(N, D, self.device, kind="slot", **(accessor or {}))
            self.rec = self.table.rec
            self.embedding = self.table.W
        elif sparse_optimizer == "adam":
            self.table = None
            self.rec = torch.zeros(N, _round_up(D, 16), dtype=torch.float32, device=self.device)  # 64-B aligned rows
            self.embedding = self.rec[:, :D]
            self.embedding.uniform_(-1e-4, 1e-4)
            self.embedding[0].zero_()                                    # padding row
        else:
            raise ValueError("sparse_optimizer must be 'adam' or 'ps'")
        sizes = [D * self.slot_num] + self.layer_sizes + [1]                        # net.py:36
        shapes = []
        for i in range(len(sizes) - 1):
            shapes += [("linear_%d.weight" % i, (sizes[i], sizes[i + 1])), ("linear_%d.bias" % i, (sizes[i + 1],))]
        # Layer 0's input width S x D (408 x 9 = 3672 on the gpubox benchmark) is no multiple of the GEMM tiles: dX_0, the
        # largest GEMM of the step, then runs on the edge-handling kernel.  The pooled features live at a padded sample stride
        # (rec_multislot_desc.out_stride; 3680 = 46 x 80: 8 % off dX_0, profiles/r04_gpubox_gemm_probe.txt) in a
        # zero-initialised buffer, the weight has ld0 - in0 zero rows behind it inside the flat parameter buffers (as in
        # deepfm.py: gradient, m and v stay exactly zero there), and the row-update kernels find a segment's gradient row
        # through rec_grad_layout{group = S, group_stride = ld0}
        self.in0 = sizes[0]
        self.ld0 = self._pad_width(self.in0) if (self.supports_padded_input and
                                                 getattr(self.k, "SUPPORTS_FEAT_LD", False)) else self.in0
        self.padded = self.ld0 != self.in0
        self.dense = _FlatParams(shapes, self.device,
                                 reserve={"linear_0.weight": self.ld0 * sizes[1]} if self.padded else None)
        self._x_bufs = {}
        self.n_linear = len(sizes) - 1
        for i in range(self.n_linear):                                              # net.py:38-46: Normal(std=0.2/sqrt(in))
            self.dense.p["linear_%d.weight" % i].normal_(0.0, 0.2 / math.sqrt(sizes[i]))
        p, g = self.dense.p, self.dense.g
        self.mlp_w = [p["linear_%d.weight" % i] for i in range(self.n_linear)]
        self.mlp_b = [p["linear_%d.bias" % i] for i in range(self.n_linear)]
        self.mlp_dw = [g["linear_%d.weight" % i] for i in range(self.n_linear)]
        self.mlp_db = [g["linear_%d.bias" % i] for i in range(self.n_linear)]
        self.gemm_w, self.gemm_dw = self.mlp_w, self.mlp_dw
        if self.padded:      # what the GEMMs see of layer 0: the weight / its gradient with the zero rows
            o, k_ = self.dense.offsets["linear_0.weight"], self.ld0 * sizes[1]
            self.gemm_w = [self.dense.data[o:o + k_].view(self.ld0, sizes[1])] + self.mlp_w[1:]
            self.gemm_dw = [self.dense.grad[o:o + k_].view(self.ld0, sizes[1])] + self.mlp_dw[1:]
        self.sparse_state = None
        self.ws = self.k.Workspace(self.device)
        self.ws_group = self.k.Workspace(self.device)
        self.ws_mlp = self.k.Workspace(self.device)
        self.status = self.k.new_status(self.device)
        self.step_count = 0
        self._side = None
        self._groups = None
        self.timers = None

    supports_padded_input = True      # (the row-sharded subclass pools into its own buffers: dense layout)

    @staticmethod
    def _pad_width(in0):
        """Smallest multiple of 80 >= in0 when that costs <= 2 % more layer-0 work (REC_SLOT_PAD0=0: dense layout)."""
        if os.environ.get("REC_SLOT_PAD0", "1") == "0" or in0 % 80 == 0:
            return in0
        c = -(-in0 // 80) * 80
        return c if (c - in0) * 50 <= in0 else in0

    # -- parameters ------------------------------------------------------------------------------
    def state_dict(self):
        sd = {"embedding": self.embedding}
        sd.update(self.dense.p)
        return sd

    def set_dict(self, sd):
        cur = self.state_dict()
        for k, v in sd.items():
            cur[k].copy_(torch.as_tensor(v).to(self.device).reshape(cur[k].shape))
        if self.table is not None and "embedding" in sd:        # explicitly set rows exist: state = embedx created
            self.rec[:, self.table.state_col] = 2.0

    def parameters(self):
        return list(self.state_dict().values())

    # -- forward ---------------------------------------------------------------------------------
    def _pool(self, mb, want_backward):
        lazy = self.table.lazy_init if self.table is not None else None
        out = None
        if self.padded:      # persistent: the padding columns are zero once and for all (the kernel never writes them)
            out = self._x_bufs.get(mb.batch)
            if out is None:
                if len(self._x_bufs) > 2:
                    self._x_bufs.clear()
                out = self._x_bufs[mb.batch] = torch.zeros(mb.batch, self.ld0, dtype=torch.float32, device=self.device)
        return self.k.multislot_sumpool(mb, self.embedding, self.dict_dim, 0, self.key_mode, self.status,
                                        want_backward=want_backward, lazy_init=lazy, out=out)

    def forward(self, mb):
        x, _, _, _, _ = self._pool(mb, False)
        y, _ = self.k.mlp_forward(x, self.gemm_w, self.mlp_b, self.ws_mlp)
        return torch.sigmoid(torch.clamp(y, CLIP[0], CLIP[1]))

    __call__ = forward

    def _ensure_sparse_state(self):
        if self.sparse_state is None and self.table is None:
            D = self.emb_dim
            Dp = _round_up(D, 4)
            mv = torch.zeros(self.rec.shape[0], _round_up(2 * Dp, 16), dtype=torch.float32, device=self.device)
            self.sparse_state = dict(mv=mv, m=mv[:, :D], v=mv[:, Dp:Dp + D])

    def _timed(self, name):
        from .deepfm import _Timed
        return _Timed(self.timers, name)

    # -- one full training step ----------------------------------------------------------------------
    def train_step(self, mb, label, lr=1e-3, auc_stats=None, show=None):
        """static_model.py:104-112 + the trainer's backward / optimizer step.  label = the click input [B,1] int64;
        show [B] int64 (None: 1 per sample, the reader's constant show slot).  Returns (loss [1], pred [B,1])."""
        k, D, S = self.k, self.emb_dim, self.slot_num
        if mb.num_slots != S:
            raise ops.RecError("batch has %d slots, the net %d" % (mb.num_slots, S))
        self._ensure_sparse_state()
        self.step_count += 1
        t = self.step_count
        on_gpu = self.device.type == "cuda"
        cur = torch.cuda.current_stream() if on_gpu else None
        if on_gpu and self._side is None:
            self._side = k.concurrent_stream(self.device)
        side = self._side if on_gpu else None
        if self._groups is None or self._groups.n < mb.nnz:
            self._groups = k.IdGroups(int(mb.nnz * 1.25) + 1, self.device)
        groups = self._groups
        with self._timed("pool_fwd"):
            x, counts, seg, rows, _ = self._pool(mb, True)
        self.last_counts = counts
        with _OnSide(side, cur):            # SelectedRows merge keys: the rows the pooling kernel resolved; every
            # value carries its (sample, slot) segment = its gradient row in dx through the sort (no index indirection)
            k.ids_group(rows[: mb.nnz], self.dict_dim, 0, self.ws_group, None, self.status, groups,
                        payload=seg[: mb.nnz])
        # the tower's tail (last Linear, clip, sigmoid, log_loss) and its backward in one pass (rec_ctr_head_fwd_bwd)
        fused_head = self.n_linear > 1 and hasattr(k, "ctr_head") and k.ctr_head_ok(self.mlp_w[-1], self.mlp_dw[-1])
        with self._timed("mlp_fwd"):
            if fused_head:
                h, acts = k.mlp_forward(x, self.gemm_w[:-1], self.mlp_b[:-1], self.ws_mlp, relu_last=True)
                pred, dz, loss, g_head = k.ctr_head(h, self.mlp_w[-1], self.mlp_b[-1], None, None, label, self.ws,
                                                    self.mlp_dw[-1], self.mlp_db[-1], clip=CLIP)
            else:
                y, acts = k.mlp_forward(x, self.gemm_w, self.mlp_b, self.ws_mlp)
        if not fused_head:
            pred, dz, loss = k.sigmoid_logloss(y, None, None, label, self.ws, clip=CLIP)
        if auc_stats is not None:
            k.auc_histogram(pred, label, auc_stats[0], auc_stats[1], NUM_THRESHOLDS)
        with self._timed("mlp_bwd"):      # dX of layer 0 BEFORE its dW: the HBM-bound sparse update below runs on the
            # side stream underneath that MFMA-bound GEMM (the largest of the step: S*D x 512 x B)
            # dW_0 runs beside the HBM-bound sparse update: half a resident round of blocks (K split 4 instead of the
            # planner's 8) leaves that kernel its wave slots — sparse_update 3.30 -> 2.72 ms, dW_0 2.63 -> 2.72, the step
            # 10.35 -> 9.75 ms on the benchmark shape (profiles/r03_schedule_ab.txt; REC_SLOT_DW0_SPLIT=0: planner's)
            kw = {}
            sk = int(os.environ.get("REC_SLOT_DW0_SPLIT", "4"))
            if on_gpu and sk > 0 and label.shape[0] >= 16384:
                kw = dict(defer_split=sk)
            if fused_head:
                dx, finish_dw0 = k.mlp_backward(g_head, acts, self.gemm_w[:-1], self.gemm_dw[:-1], self.mlp_db[:-1],
                                                self.ws_mlp, defer_first=True, **kw)              # [B, S*D]
            else:
                dx, finish_dw0 = k.mlp_backward(dz, acts, self.gemm_w, self.gemm_dw, self.mlp_db, self.ws_mlp,
                                                defer_first=True, **kw)                           # [B, S*D]
        # gradient row of segment b * S + s inside dx [B, ld0]
        lay = dict(grad_group=S, grad_group_stride=self.ld0) if self.padded else {}
        with _OnSide(side, cur):
            with self._timed("sparse_update"):
                if self.table is not None:
                    # the trainer pushes the gradient of the SUMMED loss (scale_sparse_gradient_with_batch_size [EXT];
                    # heter_ps PushCopy `* bs`): the rule then divides it by the key's pushed show
                    if self.scale_sparse_grad:
                        self.table.accessor.grad_scale = float(label.shape[0])
                    k.ps_push_rows(self.table, groups, dx, S, show=show, click=label.reshape(-1), **lay)
                else:
                    st = self.sparse_state
                    pp = self._pp = k.segment_partials(groups, dx, D, out=getattr(self, "_pp", None), **lay)
                    k.sparse_adam_rows(groups, dx, 1, self.embedding, st["m"], st["v"], t, lr, partials=pp, **lay)
        with self._timed("mlp_bwd_dw0"):
            finish_dw0()
        k.adam_dense(self.dense.data, self.dense.m, self.dense.v, self.dense.grad, t, lr)
        if on_gpu:
            cur.wait_stream(self._side)
        return loss, pred


def accessor_kwargs(tp):
    """table_parameters.embedding.accessor (slot_dnn/config_online.yaml:57-89) -> kwargs of ops.PsTable: both SGD rules
    (embed_sgd_param for embed_w, embedx_sgd_param for embedx), embedx_threshold, the score coefficients."""
    we = (tp.get("embed_sgd_param") or {}).get("adagrad", {})
    wx = (tp.get("embedx_sgd_param") or {}).get("adagrad", {}) or we
    we = we or wx
    ctr = tp.get("ctr_accessor_param", {})
    return dict(lr=we.get("learning_rate", 0.05), initial_g2sum=we.get("initial_g2sum", 3.0),
                bounds=tuple(we.get("weight_bounds", (-10.0, 10.0))), initial_range=we.get("initial_range", 1e-4),
                embedx_lr=wx.get("learning_rate", 0.05), embedx_initial_g2sum=wx.get("initial_g2sum", 3.0),
                embedx_bounds=tuple(wx.get("weight_bounds", (-10.0, 10.0))),
                embedx_initial_range=wx.get("initial_range", 1e-4), embedx_threshold=tp.get("embedx_threshold", 10),
                nonclk_coeff=ctr.get("nonclk_coeff", 0.1), click_coeff=ctr.get("click_coeff", 1.0))


class StaticModel:
    """slot_dnn/static_model.py:23-118 — the method names of the reference's model class over the engine layer."""

    def __init__(self, config):
        self.config = config
        g = config.get
        self.dict_dim = g("hyper_parameters.dict_dim")
        self.emb_dim = g("hyper_parameters.emb_dim")
        self.slot_num = g("hyper_parameters.slot_num")
        self.layer_sizes = g("hyper_parameters.layer_sizes")
        self.learning_rate = g("hyper_parameters.optimizer.learning_rate", 0.001)

    def create_model(self, device="cuda", kernels=None, sparse_optimizer=None):
        tp = self.config.get("table_parameters.embedding.accessor", None)
        acc = None
        if sparse_optimizer is None:
            sparse_optimizer = "ps" if tp else "adam"
        if tp and sparse_optimizer == "ps":          # config_online.yaml:57-89
            acc = accessor_kwargs(tp)
        return BenchmarkDNNLayer(self.dict_dim, self.emb_dim, self.slot_num, self.layer_sizes, device=device,
                                 kernels=kernels, sparse_optimizer=sparse_optimizer, accessor=acc)

    def create_metrics(self, device="cuda"):
        return auc_metrics(device)

    def train_forward(self, model, metrics_list, mb, label, show=None):
        loss, _ = model.train_step(mb, label, self.learning_rate, metrics_list[0] if metrics_list else None, show=show)
        return loss, metrics_list, {"cost": loss}

    def infer_forward(self, model, metrics_list, mb, label):
        pred = model.forward(mb)
        if metrics_list:
            model.k.auc_histogram(pred.contiguous(), label.contiguous(), metrics_list[0][0], metrics_list[0][1],
                                  NUM_THRESHOLDS)
        return metrics_list, None
