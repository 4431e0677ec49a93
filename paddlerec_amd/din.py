"""Host-side mirror of the reference's DIN plugin on the recengine HIP kernels (forward / inference).

Mirrors /root/reference/models/rank/din/net.py:20-184 (DINLayer) with the reference's parameter names
(SURVEY.md App. C): three independent item tables, three cat tables, item_b, the attention MLP (kept in
`attention_layer`, NOT in state_dict — App. B-9), linearCon and the top MLP `linear_{0,1,2}`.
The attention-pool (net.py:141-173) is ONE fused kernel (rec_din_attention_pool_fwd); linearCon and the
top MLP run on rec_gemm_f32 with the sigmoid in the epilogue; `logit + item_b` is the last GEMM's
epilogue.  train_step adds the explicit backward chain (sigmoid' fused in the dX GEMMs, bias gradients
from the dW GEMMs, rec_din_attention_pool_bwd for the rows) and the reference's optimizer: SGD with
PiecewiseDecay([410000], [base_lr, 0.2]) (din/dygraph_model.py:64-73) — embedding rows through the
merge-fused rec_sparse_sgd_rows (rows with zero gradient do not move under SGD, so this IS the dense
update Paddle performs for is_sparse=False).  As in the reference's dygraph mode the attention MLP is
not optimised (App. B-9).
"""
import collections
import math
import os

import torch

from . import ops
from . import ops as _ops


def _xavier_uniform_(t, fan_in, fan_out):
    b = math.sqrt(6.0 / (fan_in + fan_out))
    return t.uniform_(-b, b)


class _StepBuffers:
    """The reusable device buffers of one train step: GEMM / grouping workspaces, what the attention forward saves
    for its backward, the SelectedRows grouping outputs and hot-row partial sums per (lookup count, width).  The eager
    step owns one set; under train_step_graphed every input signature owns its own (a hipGraph holds addresses)."""

    def __init__(self, k, device):
        self.ws, self.ws_group = k.Workspace(device), k.Workspace(device)
        self.ws_att_bwd = k.Workspace(device)      # the attention backward's ticket counter (its own: ws is the GEMMs')
        self.ws_group_side = k.Workspace(device)   # the grouping workspace of the side stream (two-stream schedule)
        self.att_saved, self.groups, self.partials = {}, {}, {}


class DINLayer:
    """din/net.py:20-184.  forward(...) -> logit [B,1]."""

    def __init__(self, item_emb_size, cat_emb_size, act, is_sparse, use_DataLoader, item_count, cat_count,
                 device="cuda", kernels=None):
        self.device = torch.device(device)
        self.k = kernels if kernels is not None else ops     # tests may inject a stand-in backend (host logic only)
        self.item_emb_size, self.cat_emb_size = item_emb_size, cat_emb_size
        self.item_count, self.cat_count = item_count, cat_count
        f32 = dict(dtype=torch.float32, device=self.device)
        E = item_emb_size + cat_emb_size
        self.firInDim = E
        self.params = {}
        for name, rows, dim in (("hist_item_emb_attr", item_count, item_emb_size),
                                ("hist_cat_emb_attr", cat_count, cat_emb_size),
                                ("target_item_emb_attr", item_count, item_emb_size),
                                ("target_cat_emb_attr", cat_count, cat_emb_size),
                                ("target_item_seq_emb_attr", item_count, item_emb_size),
                                ("target_cat_seq_emb_attr", cat_count, cat_emb_size)):
            self.params[name + ".weight"] = _xavier_uniform_(torch.empty(rows, dim, **f32), rows, dim)
        self.params["item_b_attr.weight"] = torch.zeros(item_count, 1, **f32)          # net.py:77-82
        # attention MLP (net.py:84-102): forward uses it, state_dict does not list it (App. B-9)
        sizes = [4 * E, 80, 40, 1]
        self.attention_w = [_xavier_uniform_(torch.empty(sizes[i], sizes[i + 1], **f32), sizes[i], sizes[i + 1])
                            for i in range(3)]
        self.attention_b = [torch.zeros(sizes[i + 1], **f32) for i in range(3)]
        # the dense parameters (linearCon, linear_0..2) are views of ONE flat buffer, their gradients of another:
        # one SGD launch per step instead of eight
        con = [2 * E, 80, 40, 1]                                                         # net.py:119-137
        shapes = [("linearCon.weight", (E, E)), ("linearCon.bias", (E,))]               # net.py:109-117
        for i in range(3):
            shapes += [("linear_%d.weight" % i, (con[i], con[i + 1])), ("linear_%d.bias" % i, (con[i + 1],))]
        self._dense = torch.zeros(sum(math.prod(sh) for _, sh in shapes), **f32)
        self._dense_grad = torch.zeros_like(self._dense)
        self._gb, o = {}, 0
        for name, sh in shapes:
            k_ = math.prod(sh)
            self.params[name] = self._dense[o:o + k_].view(sh)
            self._gb[name] = self._dense_grad[o:o + k_].view(sh)
            o += k_
            if name.endswith(".weight"):
                _xavier_uniform_(self.params[name], sh[0], sh[1])
        self.ws = self.k.Workspace(self.device)
        self.status = self.k.new_status(self.device)
        self._att_saved = {}     # what the attention-pool forward keeps for its backward (buffers reused per step)
        self._bufs = None        # the eager train step's reusable buffers (_StepBuffers)
        self._side = None        # side stream of the two-stream merge schedule (_step)
        self._graph = None       # StepGraph of train_step_graphed
        self._plans, self._recording = collections.OrderedDict(), False    # recorded call lists (plan.py), per signature
        self.step_count = 0

    def state_dict(self):
        return dict(self.params)

    def set_dict(self, sd):
        for k, v in sd.items():
            self.params[k].copy_(torch.as_tensor(v).to(self.device).reshape(self.params[k].shape))

    def set_attention(self, weights, biases):
        self._plans.clear()          # a recorded step holds the address of the transposed copy of the old weights
        for dst, src in zip(self.attention_w + self.attention_b, list(weights) + list(biases)):
            dst.copy_(torch.as_tensor(src).to(self.device).reshape(dst.shape))

    def forward(self, hist_item_seq, hist_cat_seq, target_item, target_cat, label, mask, target_item_seq,
                target_cat_seq, _keep=None, _bufs=None):
        p, E = self.params, self.firInDim
        B, T = hist_item_seq.shape
        ws = _bufs.ws if _bufs is not None else self.ws
        att_saved = _bufs.att_saved if _bufs is not None else self._att_saved
        mask2 = mask.reshape(B, T).contiguous()
        pooled, attw, _ = self.k.din_attention_pool(
            hist_item_seq, hist_cat_seq, target_item_seq, target_cat_seq, mask2,
            p["hist_item_emb_attr.weight"], p["hist_cat_emb_attr.weight"],
            p["target_item_seq_emb_attr.weight"], p["target_cat_seq_emb_attr.weight"],
            self.attention_w, self.attention_b, self.status, want_weights=_keep is not None,
            saved=att_saved if _keep is not None else None,
            **(dict(ws=ws) if self.k is _ops else {}))                                 # net.py:141-173
        emb = torch.empty(B, 2 * E, dtype=torch.float32, device=self.device)           # net.py:178
        self.k.gemm(pooled, p["linearCon.weight"], ws, epilogue="bias", bias=p["linearCon.bias"],
                 out=emb[:, :E])                                                         # net.py:175-176
        ti, tc = target_item.reshape(-1).contiguous(), target_cat.reshape(-1).contiguous()
        self.k.emb_gather(ti, p["target_item_emb_attr.weight"], None, self.status, out=emb[:, E:],
                       out_group=1, out_group_stride=2 * E)                              # net.py:143,152
        self.k.emb_gather(tc, p["target_cat_emb_attr.weight"], None, self.status,
                       out=emb[:, E + self.item_emb_size:], out_group=1, out_group_stride=2 * E)
        item_b, _ = self.k.emb_gather(ti, p["item_b_attr.weight"], None, self.status)      # net.py:147
        x1 = self.k.gemm(emb, p["linear_0.weight"], ws, epilogue="bias_sigmoid", bias=p["linear_0.bias"])
        x2 = self.k.gemm(x1, p["linear_1.weight"], ws, epilogue="bias_sigmoid", bias=p["linear_1.bias"])
        logit = self.k.gemm(x2, p["linear_2.weight"], ws, epilogue="add", bias=p["linear_2.bias"],
                         aux1=item_b)                                                    # net.py:180-183
        if _keep is not None:
            _keep.update(attw=attw, pooled=pooled, emb=emb, x1=x1, x2=x2, ti=ti, tc=tc)
        return logit

    __call__ = forward

    # ---------------------------------------------------------------- training
    @staticmethod
    def learning_rate(step, base_lr):
        """paddle.optimizer.lr.PiecewiseDecay(boundaries=[410000], values=[base_lr, 0.2]) (dygraph_model.py:65-70)."""
        return base_lr if step < 410000 else 0.2

    def _sorts(self, n, table):
        """Does the row update of n lookups into `table` take the sort-based merge (else: one rec_sparse_sgd_small launch)."""
        return not (n <= getattr(self.k, "SMALL_MERGE_MAX", 0) and table.shape[1] <= 256)

    def _group(self, bufs, ids, table, slot=None, ws=None):
        """SelectedRows merge keys of `ids` into the grouping buffers `slot` (None: the one set the tables share when they
        are grouped one after the other)."""
        key = (ids.numel(), slot)
        grp = bufs.groups.get(key)
        if grp is None:
            grp = bufs.groups[key] = self.k.IdGroups(ids.numel(), self.device)
        self.k.ids_group(ids.reshape(-1), table.shape[0], None, ws if ws is not None else bufs.ws_group, None, self.status,
                         grp)
        return grp

    def _sgd_rows(self, bufs, ids, grad_view, table, lr, row_stride_floats, grp=None, slot=None):
        n = ids.numel()
        if not self._sorts(n, table):
            # the shipped batch size (32 x ~150 positions): merge + update in ONE launch instead of the 12-13 of
            # sort + partials + row update — the launches, not the work, are the step time there
            self.k.sparse_sgd_small(ids.reshape(-1), grad_view, table, lr, None, self.status, grad_group=1,
                                    grad_group_stride=row_stride_floats)
            return
        if grp is None:
            grp = self._group(bufs, ids, table)
        key = (n, slot)
        pp = bufs.partials[key, table.shape[1]] = self.k.segment_partials(grp, grad_view, table.shape[1], grad_group=1,
                                                        grad_group_stride=row_stride_floats,
                                                        out=bufs.partials.get((key, table.shape[1])))   # popular items: hot rows
        self.k.sparse_sgd_rows(grp, grad_view, table, lr, grad_group=1, grad_group_stride=row_stride_floats,
                            partials=pp)

    def train_step(self, hist_item_seq, hist_cat_seq, target_item, target_cat, label, mask, target_item_seq,
                   target_cat_seq, base_lr=0.85):
        """din/dygraph_model.py:85-100 train_forward + backward + SGD step.  label float32 [B,1].
        Returns (loss [1], pred [B,1])."""
        if (self.device.type == "cuda" and self.k is _ops and not self._recording
                and hist_item_seq.numel() <= getattr(self.k, "SMALL_MERGE_MAX", 0) and hasattr(self.k, "din_train_step")
                and os.environ.get("REC_STEP_PLAN", "1") != "0" and os.environ.get("REC_SMALL_C_STEP", "1") != "0"):
            # the shipped batch size: the step through rec_din_train_step, where the three target gathers are one launch
            # and the dense SGD rides in the merges' launch (csrc/tail_roles.h; bit-identical to the list below)
            return self.train_step_c(hist_item_seq, hist_cat_seq, target_item, target_cat, label, mask, target_item_seq,
                                     target_cat_seq, base_lr=base_lr)
        if self._bufs is None:
            self._bufs = _StepBuffers(self.k, self.device)
        lr = self.learning_rate(self.step_count, base_lr)
        self.step_count += 1
        inputs = [hist_item_seq, hist_cat_seq, target_item, target_cat, label, mask, target_item_seq, target_cat_seq]
        run = lambda: self._step(self._bufs, *inputs, lr=lr)
        # the shipped batch size (32 x ~150 positions) is launch-bound: from the third step of an input signature on the
        # step is replayed from its recorded C-ABI call list (plan.py; bit-identical, tests/test_din_gpu.py)
        small = getattr(self.k, "SMALL_MERGE_MAX", 0)
        if (self.device.type == "cuda" and self.k is _ops and not self._recording and hist_item_seq.numel() <= small
                and os.environ.get("REC_STEP_PLAN", "1") != "0"):
            from .plan import CallPlan
            # the recorded calls hold the transposed copy of the attention weights made at record time: an in-place edit of
            # attention_w (its torch version counter) gets a new plan, as set_attention does
            key = (tuple(hist_item_seq.shape), float(lr), tuple(w._version for w in self.attention_w))
            entry = self._plans.get(key)
            if entry is None:
                self._plans[key] = "seen"
                if len(self._plans) > 32:
                    self._plans.popitem(last=False)
            elif entry == "seen" or not entry.matches(inputs):
                plan = CallPlan()
                self._recording = True
                try:
                    out = plan.record(run, inputs)
                finally:
                    self._recording = False
                self._plans[key] = plan
                return out
            else:
                self._plans.move_to_end(key)
                return entry.replay(inputs, 0, float(lr))
        return run()

    def c_net(self):
        """rec_din_net over this layer's tensors (rebuilt when the attention weights changed: it holds their transpose)."""
        from . import _lib
        ver = tuple(w._version for w in self.attention_w)
        cached = getattr(self, "_c_net", None)
        if cached is not None and cached[0] == ver:
            return cached[1]
        p = self.params
        net = _lib.DinNet()
        net.item_dim, net.cat_dim = self.item_emb_size, self.cat_emb_size
        net.item_rows, net.cat_rows = self.item_count, self.cat_count
        net.att_hidden1, net.att_hidden2 = self.attention_w[0].shape[1], self.attention_w[1].shape[1]
        net.mlp_hidden1, net.mlp_hidden2 = p["linear_0.weight"].shape[1], p["linear_1.weight"].shape[1]
        for field, name in (("w_hist_item", "hist_item_emb_attr"), ("w_hist_cat", "hist_cat_emb_attr"),
                            ("w_tgt_item_seq", "target_item_seq_emb_attr"), ("w_tgt_cat_seq", "target_cat_seq_emb_attr"),
                            ("w_tgt_item", "target_item_emb_attr"), ("w_tgt_cat", "target_cat_emb_attr"),
                            ("w_item_b", "item_b_attr")):
            setattr(net, field, p[name + ".weight"].data_ptr())
        w1t = self.attention_w[0].t().contiguous()
        net.att_w1, net.att_w1_t = self.attention_w[0].data_ptr(), w1t.data_ptr()
        net.att_b1, net.att_w2, net.att_b2 = (self.attention_b[0].data_ptr(), self.attention_w[1].data_ptr(),
                                               self.attention_b[1].data_ptr())
        net.att_w3, net.att_b3 = self.attention_w[2].data_ptr(), self.attention_b[2].data_ptr()
        for tag, name in (("con", "linearCon"), ("l0", "linear_0"), ("l1", "linear_1"), ("l2", "linear_2")):
            setattr(net, "w_" + tag, p[name + ".weight"].data_ptr())
            setattr(net, "b_" + tag, p[name + ".bias"].data_ptr())
            setattr(net, "g_w_" + tag, self._gb[name + ".weight"].data_ptr())
            setattr(net, "g_b_" + tag, self._gb[name + ".bias"].data_ptr())
        net.flat_param, net.flat_grad, net.flat_numel = (self._dense.data_ptr(), self._dense_grad.data_ptr(),
                                                         self._dense.numel())
        self._c_net = (ver, net, w1t)              # w1t: kept alive with the struct that points at it
        return net

    def train_step_c(self, hist_item_seq, hist_cat_seq, target_item, target_cat, label, mask, target_item_seq,
                     target_cat_seq, base_lr=0.85):
        """train_step through rec_din_train_step: ONE foreign call per step — what a non-Python binder of
        include/recengine.h gets (csrc/din_step.hip).  Same entry points, arguments and order as _step: bit-identical."""
        lr = self.learning_rate(self.step_count, base_lr)
        self.step_count += 1
        if getattr(self, "_ws_c", None) is None:
            self._ws_c = self.k.Workspace(self.device)
        B, T = hist_item_seq.shape
        side = None
        if (self.device.type == "cuda" and self._sorts(B * T, self.params["hist_item_emb_attr.weight"])
                and os.environ.get("REC_DIN_SIDE", "1") != "0"):
            if self._side is None:
                self._side = self.k.concurrent_stream(self.device)
            side = self._side                   # the two-stream schedule of _step (large batches)
        return self.k.din_train_step(self.c_net(), hist_item_seq, hist_cat_seq, target_item.reshape(-1).contiguous(),
                                     target_cat.reshape(-1).contiguous(), label.reshape(-1).contiguous(),
                                     mask.reshape(B, T).contiguous(), target_item_seq, target_cat_seq, lr, self._ws_c,
                                     status=self.status, side_stream=side)

    def train_step_graphed(self, hist_item_seq, hist_cat_seq, target_item, target_cat, label, mask, target_item_seq,
                           target_cat_seq, base_lr=0.85):
        """train_step replayed from a hipGraph per input signature (paddlerec_amd/graph.py): at the shipped batch size
        (din/config.yaml: 32) the step is ~120 launches of a few microseconds and the host's launch path is the step
        time.  Same arithmetic, same kernels, same order; the returned (loss, pred) are the graph's static outputs,
        overwritten by the next step of the same signature."""
        if self._graph is None:
            import weakref
            from .graph import StepGraph
            me = weakref.ref(self)                          # no cycle model <-> graph: see graph.py
            self._graph = StepGraph(lambda st, *a, **kw: me()._step(st, *a, **kw),
                                    state_factory=lambda: _StepBuffers(me().k, me().device))
        lr = self.learning_rate(self.step_count, base_lr)     # host scalar baked into the launches: part of the key
        self.step_count += 1
        return self._graph(hist_item_seq, hist_cat_seq, target_item, target_cat, label, mask, target_item_seq,
                           target_cat_seq, lr=lr)

    def _step(self, bufs, hist_item_seq, hist_cat_seq, target_item, target_cat, label, mask, target_item_seq,
              target_cat_seq, lr):
        p, E, Ei = self.params, self.firInDim, self.item_emb_size
        B, T = hist_item_seq.shape
        sv = {}
        # Two-stream schedule of the sort-based merge (batches above the one-launch merge): the merge keys of the four
        # [B, T] tables depend on the ids only — grouped on a side stream from the start of the step —, and the four row
        # updates, independent tables, go two per stream behind the backward.  Same calls, same arguments: bit-identical
        # (B 4096: 1.72 -> 1.47 ms at T 100, 5.34 -> 5.04 ms at T 512; profiles/r05_din.txt)
        seq = [(target_item_seq, p["target_item_seq_emb_attr.weight"]), (target_cat_seq, p["target_cat_seq_emb_attr.weight"]),
               (hist_item_seq, p["hist_item_emb_attr.weight"]), (hist_cat_seq, p["hist_cat_emb_attr.weight"])]
        two = (self.k is _ops and self.device.type == "cuda" and os.environ.get("REC_DIN_SIDE", "1") != "0"
               and all(self._sorts(ids.numel(), tab) for ids, tab in seq) and not torch.cuda.is_current_stream_capturing())
        pre = [None] * 7
        if two:
            if self._side is None:
                self._side = self.k.concurrent_stream(self.device)
            cur, side = torch.cuda.current_stream(), self._side
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for i, (ids, tab) in enumerate(seq):
                    pre[i] = self._group(bufs, ids, tab, slot=i, ws=bufs.ws_group_side)
        logit = self.forward(hist_item_seq, hist_cat_seq, target_item, target_cat, label, mask,
                             target_item_seq, target_cat_seq, _keep=sv, _bufs=bufs)
        pred, dz, loss = self.k.bce_with_logits(logit, label.reshape(B, 1).contiguous(), bufs.ws)
        g = {}
        ws = bufs.ws

        def lin_bwd(name, x, dy, act=None):
            """dW, db of Linear `name` (input x, output-gradient dy); returns d x (sigmoid' of x fused when act)."""
            if hasattr(self.k, "linear_backward"):      # both GEMMs in one call: one launch at the shipped batch size
                g[name + ".weight"], g[name + ".bias"] = self._gb[name + ".weight"], self._gb[name + ".bias"]
                return self.k.linear_backward(x, dy, p[name + ".weight"], ws, g[name + ".weight"], g[name + ".bias"],
                                              **(dict(epilogue="dsigmoid", aux0=act) if act is not None else {}))
            g[name + ".weight"] = self.k.gemm(x, dy, ws, trans_a=True, out=self._gb[name + ".weight"],
                                              b_colsum=self._gb[name + ".bias"])
            g[name + ".bias"] = self._gb[name + ".bias"]
            if act is None:
                return self.k.gemm(dy, p[name + ".weight"], ws, trans_b=True)
            return self.k.gemm(dy, p[name + ".weight"], ws, trans_b=True, epilogue="dsigmoid", aux0=act)

        d2 = lin_bwd("linear_2", sv["x2"], dz, act=sv["x2"])
        d1 = lin_bwd("linear_1", sv["x1"], d2, act=sv["x1"])
        de0 = lin_bwd("linear_0", sv["emb"], d1)                       # [B, 2E] = [d linearCon out | d target_concat]
        dpooled = lin_bwd("linearCon", sv["pooled"], de0[:, :E])
        dh, dq = self.k.din_attention_pool_bwd(
            hist_item_seq, hist_cat_seq, target_item_seq, target_cat_seq,
            p["hist_item_emb_attr.weight"], p["hist_cat_emb_attr.weight"],
            p["target_item_seq_emb_attr.weight"], p["target_cat_seq_emb_attr.weight"],
            self.attention_w, self.attention_b, sv["attw"], dpooled, saved=bufs.att_saved,
            **(dict(ws=bufs.ws_att_bwd) if self.k is _ops else {}))
        self._last = dict(dh=dh, dq=dq, de0=de0, dz=dz, dense=g)
        # ---- SGD (dygraph_model.py:64-73).  Embedding tables: merged rows; dense: in place.
        # (the target-seq tables first: one row per sample collects the gradients of all its history positions — the longest
        # serial chains of the one-launch merge; their blocks should be the first to start, not the last)
        jobs = [(target_item_seq, dq, p["target_item_seq_emb_attr.weight"], E),
                (target_cat_seq, dq[:, :, Ei:], p["target_cat_seq_emb_attr.weight"], E),
                (hist_item_seq, dh, p["hist_item_emb_attr.weight"], E),
                (hist_cat_seq, dh[:, :, Ei:], p["hist_cat_emb_attr.weight"], E),
                (sv["ti"], de0[:, E:], p["target_item_emb_attr.weight"], 2 * E),
                (sv["tc"], de0[:, E + Ei:], p["target_cat_emb_attr.weight"], 2 * E),
                (sv["ti"], dz, p["item_b_attr.weight"], 1)]
        small = getattr(self.k, "SMALL_MERGE_MAX", 0)
        if (hasattr(self.k, "sparse_sgd_small_multi") and os.environ.get("REC_SMALL_MULTI", "1") != "0"
                and all(ids.numel() <= small and tab.shape[1] <= 256 for ids, _, tab, _ in jobs)):
            # the shipped batch size: the seven independent merges share ONE launch (rec_sparse_sgd_small_multi)
            self.k.sparse_sgd_small_multi([(ids.reshape(-1), gv, tab, 1, rs) for ids, gv, tab, rs in jobs], lr,
                                          self.status)
        elif two:
            cur.wait_stream(side)                               # the merge keys
            side.wait_stream(cur)                               # the gradients
            on_side = (0, 3)            # one item and one category table per stream
            with torch.cuda.stream(side):
                for i in on_side:
                    ids, gv, tab, rs = jobs[i]
                    self._sgd_rows(bufs, ids, gv, tab, lr, rs, grp=pre[i], slot=i)
            for i, (ids, gv, tab, rs) in enumerate(jobs):
                if i not in on_side:
                    self._sgd_rows(bufs, ids, gv, tab, lr, rs, grp=pre[i], slot=i)
        else:
            for ids, gv, tab, rs in jobs:
                self._sgd_rows(bufs, ids, gv, tab, lr, rs)
        self.k.sgd_dense(self._dense, self._dense_grad, lr)      # all four Linear layers (every gradient was written above)
        if two:
            cur.wait_stream(side)
        return loss, pred


NUM_THRESHOLDS = 4095  # paddle.metric.Auc default [EXT]


class DygraphModel:
    """din/dygraph_model.py:21-113 — same method names; tensors are torch device tensors."""

    def create_model(self, config, device="cuda", kernels=None):
        g = config.get
        return DINLayer(g("hyper_parameters.item_emb_size", 64), g("hyper_parameters.cat_emb_size", 64),
                        g("hyper_parameters.act", "sigmoid"), g("hyper_parameters.is_sparse", False),
                        g("hyper_parameters.use_DataLoader", False), g("hyper_parameters.item_count", 63001),
                        g("hyper_parameters.cat_count", 801), device=device, kernels=kernels)

    def create_feeds(self, batch, config, device="cuda"):
        t = [torch.as_tensor(x).to(device) for x in batch]
        label = t[4].to(torch.float32).reshape(-1, 1)                                  # dygraph_model.py:52
        return t[0], t[1], t[2], t[3], label, t[5], t[6], t[7]

    def create_metrics(self, device="cuda"):
        stats = (torch.zeros(NUM_THRESHOLDS + 1, dtype=torch.int64, device=device),
                 torch.zeros(NUM_THRESHOLDS + 1, dtype=torch.int64, device=device))
        return [stats], ["auc"]

    def _auc(self, dy_model, metrics_list, pred, label):
        if metrics_list:
            dy_model.k.auc_histogram(pred.contiguous(), label.to(torch.int64).contiguous(), metrics_list[0][0],
                              metrics_list[0][1], NUM_THRESHOLDS)

    def train_forward(self, dy_model, metrics_list, batch_data, config):
        feeds = self.create_feeds(batch_data, config, dy_model.device)
        base_lr = config.get("hyper_parameters.optimizer.learning_rate_base_lr")
        loss, pred = dy_model.train_step(*feeds, base_lr=base_lr)
        self._auc(dy_model, metrics_list, pred, feeds[4])
        return loss, metrics_list, {"loss": loss}

    def infer_forward(self, dy_model, metrics_list, batch_data, config):
        feeds = self.create_feeds(batch_data, config, dy_model.device)
        pred = torch.sigmoid(dy_model.forward(*feeds))
        self._auc(dy_model, metrics_list, pred, feeds[4])
        return metrics_list, None
