"""Host-side mirror of the reference's DeepFM plugin, running on the recengine HIP kernels.

Mirrors (same class names, constructor arguments, parameter names and forward signature):
    /root/reference/models/rank/deepfm/net.py:21-174        DeepFMLayer / FM / DNN
    /root/reference/models/rank/deepfm/dygraph_model.py     DygraphModel (create_model, create_loss,
                                                            create_optimizer, train_forward, ...)
The embedding lookups, FM block, loss head, SelectedRows merge, sparse/dense Adam and the AUC
histogram are hand-written HIP kernels behind the C-ABI (include/recengine.h); the top-MLP GEMMs run on the
MFMA units through rec_gemm_f32 (exact-f32 v_mfma_f32_16x16x4, bias/ReLU/ReLU' fused in the epilogue).
There is no autograd tape and no CPU fallback: backward is the explicit chain the reference's
`loss.backward()` (tools/trainer.py:151) implies.
"""
import math
import os
import time

import torch

from . import ops

NUM_THRESHOLDS = 4095  # paddle.metric.Auc default [EXT]


class _FlatParams:
    """All dense parameters in ONE flat f32 buffer (+ grad, Adam m/v): one optimizer launch and one
    all-reduce bucket per step instead of one per tensor.  Every tensor starts on a 256-byte boundary (the gaps stay
    zero in all four buffers): the GEMM loaders and the head kernels take 16-byte vector loads of the weights, and an
    odd-sized tensor in front (fm.dense_w [13], a bias [1]) would otherwise push every later one off alignment and
    silently onto the scalar-load variants.  `packed()` / `load_packed()` give the gap-free declaration-order image
    checkpoints hold."""
    ALIGN = 64          # floats

    def __init__(self, shapes, device, reserve=None):
        """reserve: {name: floats}: room kept behind that tensor (zero like every gap, and it stays zero under the
        optimizers: zero gradient on a zero parameter) — a zero-padded view of it then needs no copy."""
        self.names = [n for n, _ in shapes]
        self.shapes = dict(shapes)
        self.offsets, o = {}, 0
        for n, s in shapes:
            self.offsets[n] = o
            o += -(-max(math.prod(s), (reserve or {}).get(n, 0)) // self.ALIGN) * self.ALIGN
        self.data = torch.zeros(max(o, 1), dtype=torch.float32, device=device)
        self.grad = torch.zeros_like(self.data)
        self.m = torch.zeros_like(self.data)
        self.v = torch.zeros_like(self.data)
        self.p, self.g, self.pm, self.pv = {}, {}, {}, {}
        for n, s in shapes:
            o, k = self.offsets[n], math.prod(s)
            self.p[n] = self.data[o:o + k].view(s)
            self.g[n] = self.grad[o:o + k].view(s)
            self.pm[n] = self.m[o:o + k].view(s)
            self.pv[n] = self.v[o:o + k].view(s)

    def packed(self, buf):
        """Gap-free image of one of the four buffers, tensors in declaration order."""
        return torch.cat([buf[self.offsets[n]:self.offsets[n] + math.prod(self.shapes[n])] for n in self.names]) \
            if self.names else buf[:0]

    def load_packed(self, buf, flat):
        flat = torch.as_tensor(flat).reshape(-1)
        want = sum(math.prod(self.shapes[n]) for n in self.names)
        if flat.numel() != want:
            raise ValueError("packed dense image has %d floats, the parameters %d" % (flat.numel(), want))
        o = 0
        for n in self.names:
            k = math.prod(self.shapes[n])
            buf[self.offsets[n]:self.offsets[n] + k].copy_(flat[o:o + k].to(buf.device))
            o += k


class _Timed:
    """Brackets a region with HIP events on the current stream when a timers dict is installed."""

    def __init__(self, timers, name):
        self.timers, self.name = timers, name

    def __enter__(self):
        if self.timers is not None:
            self.h0 = time.perf_counter()
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *a):
        if self.timers is not None:
            self.e1.record()
            self.timers.setdefault(self.name, []).append((self.e0, self.e1))
            # host time spent ISSUING the region (launch overhead, host syncs) under "<name>@host"
            self.timers.setdefault(self.name + "@host", []).append(time.perf_counter() - self.h0)


class _OnSide:
    """`with _OnSide(side, cur):` — issue on the side stream, ordered after everything issued so far on `cur`.
    side None (a CPU device: orchestration tests with an injected operator backend) makes it a no-op."""

    def __init__(self, side, cur):
        self.side, self.cur = side, cur

    def __enter__(self):
        if self.side is not None:
            self.side.wait_stream(self.cur)
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.side is not None:
            self.ctx.__exit__(*a)


def _round_up(x, m):
    return (x + m - 1) // m * m


class FM:
    """net.py:52-139.  Holds embedding_one [N,1], embedding [N,D]; dense_w_one/dense_w live in the
    flat dense buffer of the owning DeepFMLayer."""

    def __init__(self, sparse_feature_number, sparse_feature_dim, dense_feature_dim,
                 sparse_num_field, device, slot_offset=None, zero_padding_row=True, rec=None):
        """rec: an existing record buffer [N, >= D+3] to use as the table WITHOUT initialising it (the PS table of
        the gpubox mode: rows are born lazily, a 160 GB shard is never swept by an init pass)."""
        self.sparse_feature_number = sparse_feature_number
        self.sparse_feature_dim = sparse_feature_dim
        self.dense_feature_dim = dense_feature_dim
        self.dense_emb_dim = sparse_feature_dim
        self.sparse_num_field = sparse_num_field
        self.init_value_ = 0.1
        self.padding_idx = 0                                   # net.py:69,81
        std = self.init_value_ / math.sqrt(float(sparse_feature_dim))
        N, D = sparse_feature_number, sparse_feature_dim
        # Table layout (DESIGN.md "table layout"): HBM is fetched in whole 128-B lines, so both
        # embeddings of a row live in ONE line-aligned record  [W(D) | W1 | m1 | v1 | pad]  and a
        # lookup costs one line instead of two; the second-order Adam moments sit in a second
        # record buffer [m(D) | v(D)] that only the optimizer touches.  The reference's two
        # parameters are views: embedding = rec[:, :D], embedding_one = rec[:, D:D+1].
        self.slot_offset = slot_offset
        if rec is not None:
            self.rec, self.rec_width = rec, rec.shape[1]
            self.embedding = self.rec[:, :D]
            self.embedding_one = self.rec[:, D:D + 1]
            return
        # D + 3 <= 16 (the reference's own D 9 / D 10): the record is ONE 64-byte half line — the fabric fetches 64-byte
        # requests for it (TCC_EA0_RDREQ_64B), so a lookup moves half the bytes of a 128-byte record; wider rows keep
        # whole lines.  REC_TABLE_RECORD_FLOATS overrides (a multiple of 4 >= D + 3).
        self.rec_width = 16 if D + 3 <= 16 else _round_up(D + 3, 32)
        forced = int(os.environ.get("REC_TABLE_RECORD_FLOATS", "0"))
        if forced:
            if forced % 4 or forced < D + 3:
                raise ValueError("REC_TABLE_RECORD_FLOATS=%d: need a multiple of 4 >= D + 3 = %d" % (forced, D + 3))
            self.rec_width = forced
        self.rec = torch.zeros(N, self.rec_width, dtype=torch.float32, device=device)
        self.embedding = self.rec[:, :D]
        self.embedding_one = self.rec[:, D:D + 1]
        for t in (self.embedding_one, self.embedding):        # TruncatedNormal(0,std) net.py:72-75
            torch.nn.init.trunc_normal_(t, 0.0, std, -2 * std, 2 * std)
            if slot_offset is None and zero_padding_row:
                t[self.padding_idx].zero_()                    # padding row zeroed at construction [EXT]


class DeepFMLayer:
    """net.py:21-49.  forward(sparse_inputs, dense_inputs) -> predict [B,1]."""

    def __init__(self, sparse_feature_number, sparse_feature_dim, dense_feature_dim,
                 sparse_num_field, layer_sizes, device="cuda", slot_offset=None, table_rows=None,
                 zero_padding_row=True, kernels=None, extra_dense=(), table_rec=None):
        """table_rows: rows actually allocated on this device (row-sharded subclass: ceil(N/G));
        kernels: operator backend (default: the HIP kernels of paddlerec_amd.ops — tests/ may inject
        a stand-in to exercise host orchestration without a GPU; the product never does);
        extra_dense: extra (name, shape) entries appended to the flat dense buffer."""
        self.device = torch.device(device)
        self.k = kernels if kernels is not None else ops
        self.sparse_feature_number = sparse_feature_number
        self.sparse_feature_dim = sparse_feature_dim
        self.dense_feature_dim = dense_feature_dim
        self.sparse_num_field = sparse_num_field
        self.layer_sizes = list(layer_sizes)
        # slot s owning rows [s*R, (s+1)*R) (BASELINE configs[1]: 26 tables x 1M rows as one table): the merge keys are
        # then sorted slot by slot (rec_ids_group_slots) and fm_bwd writes its row gradients in sorted order
        self.slot_rows = None
        if slot_offset is not None:
            so = [int(x) for x in torch.as_tensor(slot_offset).reshape(-1).tolist()]
            # ... and only when the table IS S equal spans: rec_ids_group_slots keeps a lookup iff 0 <= id < slot_rows,
            # fm_fwd iff id + slot_offset[s] < num_rows — with a longer last span (or extra rows behind the spans) the
            # forward would look a row up whose gradient the slot-local merge drops
            rows_total = table_rows if table_rows is not None else sparse_feature_number
            if (len(so) >= 2 and so[0] == 0 and so[1] > 0 and all(so[i] == i * so[1] for i in range(len(so)))
                    and rows_total == len(so) * so[1]):
                self.slot_rows = so[1]
            slot_offset = torch.as_tensor(slot_offset, dtype=torch.int64, device=self.device)
        self.fm = FM(table_rows if table_rows is not None else sparse_feature_number,
                     sparse_feature_dim, dense_feature_dim, sparse_num_field, self.device,
                     slot_offset, zero_padding_row, rec=table_rec)
        D, Dn = sparse_feature_dim, dense_feature_dim
        self.num_field = Dn + sparse_num_field
        sizes = [D * self.num_field] + self.layer_sizes + [1]               # net.py:150
        shapes = [("fm.dense_w_one", (Dn,)), ("fm.dense_w", (1, Dn, D))]
        for i in range(len(sizes) - 1):
            shapes += [("dnn.linear_%d.weight" % i, (sizes[i], sizes[i + 1])),
                       ("dnn.linear_%d.bias" % i, (sizes[i + 1],))]
        shapes.append(("bias", (1,)))       # created, never used in forward (net.py:36-39; App. B-14)
        shapes += list(extra_dense)
        # Dense "embeddings" x_j * dense_w[j,:] are never materialised (DESIGN.md §3 "compact feat"): feat keeps
        # the S embedding rows + one row of raw dense values, and layer 0 runs on folded weights
        #   W0' = [ W0[:S*D] ; M ; 0 ],  M[j,:] = dense_w[j,:] @ W0[(S+j)*D:(S+j+1)*D, :]   (rebuilt every step)
        self.compact = 0 < Dn <= D
        self.fp = (sparse_num_field + 1) if self.compact else self.num_field      # fields per sample in feat
        # Layer-0 width that is no multiple of the GEMM tiles (the reference's own layout: 39 fields x D 9 / 10 = 351 /
        # 390 columns put all three layer-0 GEMMs on the edge-handling kernels, 0.9 ms instead of 0.6 per step at B 65536):
        # feat lives in a zero-initialised [B, ld0] buffer (rec_deepfm_desc.feat_stride) and layer 0 runs on the weight
        # with ld0 - in0 zero rows behind it — whole tiles for forward, dX and dW.  The rows are the gap the flat
        # buffer keeps behind linear_0.weight (parameter AND gradient: the dW GEMM writes exact zeros there, feat's
        # padding columns being zero), so neither direction copies anything
        self.in0 = self.fp * D
        self.ld0 = self._pad_width(self.in0) if (not self.compact and self.supports_padded_feat
                                                 and getattr(self.k, "SUPPORTS_FEAT_LD", False)) else self.in0
        self.padded = self.ld0 != self.in0
        self.dense = _FlatParams(shapes, self.device,
                                 reserve={"dnn.linear_0.weight": self.ld0 * sizes[1]} if self.padded else None)
        std = 0.1 / math.sqrt(float(D))
        for n in ("fm.dense_w_one", "fm.dense_w"):                          # net.py:89-103
            torch.nn.init.trunc_normal_(self.dense.p[n], 0.0, std, -2 * std, 2 * std)
        self.n_linear = len(sizes) - 1
        for i in range(self.n_linear):                                      # net.py:155-160
            self.dense.p["dnn.linear_%d.weight" % i].normal_(0.0, 1.0 / math.sqrt(sizes[i]))
        self.mlp_w = [self.dense.p["dnn.linear_%d.weight" % i] for i in range(self.n_linear)]
        self.mlp_b = [self.dense.p["dnn.linear_%d.bias" % i] for i in range(self.n_linear)]
        self.mlp_dw = [self.dense.g["dnn.linear_%d.weight" % i] for i in range(self.n_linear)]
        self.mlp_db = [self.dense.g["dnn.linear_%d.bias" % i] for i in range(self.n_linear)]
        # sparse Adam state (lazy rows) + bookkeeping
        self.sparse_state = None
        if self.compact:
            self._w0p = torch.zeros(self.fp * D, sizes[1], dtype=torch.float32, device=self.device)
            self._dw0f = torch.zeros(self.fp * D, sizes[1], dtype=torch.float32, device=self.device)
            self._dm = torch.zeros(Dn, sizes[1], dtype=torch.float32, device=self.device)
        if self.padded:
            o, k = self.dense.offsets["dnn.linear_0.weight"], self.ld0 * sizes[1]
            self._w0p = self.dense.data[o:o + k].view(self.ld0, sizes[1])
            self._dw0p = self.dense.grad[o:o + k].view(self.ld0, sizes[1])
            self._fm_bufs = {}
        self.ws = self.k.Workspace(self.device)
        self.ws_group = self.k.Workspace(self.device)
        self.ws_mlp = self.k.Workspace(self.device)
        self.status = self.k.new_status(self.device)
        self.step_count = 0
        self._side = None
        # Pipelined steps (round 6; bench.py and the trainer loops switch it on): the step's critical chain
        # fm_bwd -> sparse update -> the NEXT step's lookup stays on ONE stream (no cross-queue event latency), the dense
        # tail (dW_0, fold, dense Adam, the next step's folded layer-0 weight and every GEMM weight image) runs on the side
        # stream and is joined only where the next step's first GEMM needs it.  Between steps the dense parameters /
        # gradients are therefore still being written: read them after sync().  Same kernels, same arguments: bit-identical.
        self.pipelined = os.environ.get("REC_DEEPFM_PIPELINED", "0") == "1"
        self._side_pending, self._pending_refs, self._gside, self._adam_done = False, None, None, None
        self._w_key, self._images = None, None        # (param version, step, images?) the folded weight / images belong to
        self._plans, self._recording = {}, False      # recorded call lists of launch-bound steps (plan.py)
        self.timers = None      # bench.py: dict name -> list of (start,end) torch.cuda.Event pairs

    supports_padded_feat = True      # (the row-sharded subclass keeps its own FM path: dense feat)

    @staticmethod
    def _pad_width(in0):
        """Smallest multiple of the whole-tile widths (80, 144) >= in0, if that costs <= 20 % more layer-0 work."""
        if os.environ.get("REC_DEEPFM_PAD0", "1") == "0" or in0 % 80 == 0 or in0 % 144 == 0:
            return in0
        c = min(-(-in0 // 80) * 80, -(-in0 // 144) * 144)
        return c if (c - in0) * 5 <= in0 else in0

    # -- parameters under the reference's state_dict keys (Appendix C) -------------------------
    def state_dict(self):
        self.sync()
        sd = {"fm.embedding_one.weight": self.fm.embedding_one, "fm.embedding.weight": self.fm.embedding}
        sd.update({k: v for k, v in self.dense.p.items() if not k.startswith("__")})
        return sd

    def set_dict(self, sd):
        for k, v in sd.items():
            dst = self.state_dict()[k]
            dst.copy_(torch.as_tensor(v).to(dst.device).reshape(dst.shape))

    def parameters(self):
        return list(self.state_dict().values())

    # -- forward (net.py:41-49) -----------------------------------------------------------------
    @staticmethod
    def _concat_ids(sparse_inputs):
        if isinstance(sparse_inputs, (list, tuple)):
            return torch.cat(list(sparse_inputs), dim=1).contiguous()       # net.py:107
        return sparse_inputs

    def _fm_fwd(self, ids, dense_inputs):
        if self.padded:
            B, dev = ids.shape[0], self.device
            bufs = self._fm_bufs.get(B)
            if bufs is None:
                if len(self._fm_bufs) > 4:
                    self._fm_bufs.clear()
                f32 = dict(dtype=torch.float32, device=dev)
                bufs = self._fm_bufs[B] = (torch.empty(B, 1, **f32), torch.empty(B, 1, **f32),
                                           torch.zeros(B, self.ld0, **f32), torch.empty(B, self.sparse_feature_dim, **f32))
            return self.k.deepfm_fm_fwd(ids, dense_inputs, self.fm.embedding, self.fm.embedding_one,
                                        self.dense.p["fm.dense_w"], self.dense.p["fm.dense_w_one"],
                                        self.fm.padding_idx, self.fm.slot_offset, self.status, out=bufs,
                                        compact=False, feat_ld=self.ld0)
        return self.k.deepfm_fm_fwd(ids, dense_inputs, self.fm.embedding, self.fm.embedding_one,
                                    self.dense.p["fm.dense_w"], self.dense.p["fm.dense_w_one"],
                                    self.fm.padding_idx, self.fm.slot_offset, self.status,
                                    compact=self.compact)

    # -- layer 0 on folded weights ------------------------------------------------------------------
    def _weight_lists(self):
        if self.padded:
            return [self._w0p] + self.mlp_w[1:], [self._dw0p] + self.mlp_dw[1:]
        if not self.compact:
            return self.mlp_w, self.mlp_dw
        if self._fold_full:
            return [self._w0p] + self.mlp_w[1:], [self._dw0f] + self.mlp_dw[1:]
        return [self._w0p] + self.mlp_w[1:], [self.mlp_dw[0][: self.fp * self.sparse_feature_dim]] + self.mlp_dw[1:]

    def _refresh_weights(self, images=False):
        """What the GEMMs of a step read besides the parameters themselves, on the current stream: layer 0's folded
        weight (compact feat) and — images=True — the bf16 x 3 image of every Linear's weight in both orientations
        (forward: W, dX: W^T), ALL in one launch (rec_gemm_b_images) instead of one split launch in front of each GEMM."""
        if self.compact and not self.padded:
            S, D, Dn = self.sparse_num_field, self.sparse_feature_dim, self.dense_feature_dim
            w0 = self.mlp_w[0]
            if self._fold_full:
                self.k.dense_fold_fwd_full(S, self.dense.p["fm.dense_w"].view(Dn, D), w0, self._w0p)
            else:
                self._copy(self._w0p[: S * D], w0[: S * D])
                self.k.dense_fold_fwd(S, self.dense.p["fm.dense_w"].view(Dn, D), w0, self._w0p[S * D: S * D + Dn])
        if images:
            if self._images is None:
                ws_ = self._weight_lists()[0]
                self._images = self.k.GemmImages([(w, False) for w in ws_] + [(w, True) for w in ws_], self.device)
            self._images.refresh()

    def _mlp_weights(self, images=False):
        """(weights, weight-grad views) of the top MLP as the GEMMs see them this step.  The folded weight / the images
        are rebuilt unless they were made for exactly these parameter values and this step (the pipelined step makes
        them at the end of the previous one, behind its dense Adam)."""
        key = (self.dense.data._version, self.step_count)
        if self._w_key is None or self._w_key[:2] != key or (images and not self._w_key[2]):
            self._refresh_weights(images)
            self._w_key = key + (bool(images),)
        return self._weight_lists()

    def _image_lists(self):
        """(images of the weights, images of their transposes) for mlp_forward / mlp_backward, or (None, None)."""
        if self._images is None or self._w_key is None or not self._w_key[2]:
            return None, None
        n = self.n_linear
        return [self._images.get(i) for i in range(n)], [self._images.get(n + i) for i in range(n)]

    def sync(self):
        """Joins the side stream of a pipelined step into the current stream: after it, parameters, gradients and
        optimizer state are ordered behind everything the last step issued (checkpoints, eval, tests)."""
        if self._side_pending:
            torch.cuda.current_stream().wait_event(self._side_pending)       # recorded behind the tail's last kernel
            self._side_pending, self._pending_refs, self._adam_done = False, None, None

    @property
    def _fold_full(self):
        """One launch per direction for the dense fold (rec_dense_fold_*_full) where the backend has it."""
        return hasattr(self.k, "dense_fold_fwd_full") and os.environ.get("REC_FOLD_FULL", "1") != "0"

    def _copy(self, dst, src):
        """Parameter-slice copy as a C-ABI call where the backend has one (a recorded step must not hide a torch kernel)."""
        f = getattr(self.k, "copy_f32", None)
        if f is not None and dst.is_cuda:
            f(dst, src)
        else:
            dst.copy_(src)

    def _fold_backward(self):
        """After dW0' = feat'^T dZ0 landed in the first (S+1)*D rows of the layer-0 gradient buffer: turn its
        dense rows (= dM) into the gradients of the real parameters (W0 dense rows, dense_w MLP part)."""
        if self.padded or not self.compact:
            return
        S, D, Dn = self.sparse_num_field, self.sparse_feature_dim, self.dense_feature_dim
        if self._fold_full:
            self.k.dense_fold_bwd_full(S, self.dense.p["fm.dense_w"].view(Dn, D), self.mlp_w[0], self._dw0f,
                                       self.mlp_dw[0], self.dense.g["fm.dense_w"].view(Dn, D), accumulate=True)
            return
        self._copy(self._dm, self.mlp_dw[0][S * D: S * D + Dn])
        self.k.dense_fold_bwd(S, self.dense.p["fm.dense_w"].view(Dn, D), self.mlp_w[0], self._dm, self.mlp_dw[0],
                              self.dense.g["fm.dense_w"].view(Dn, D), accumulate=True)

    def forward(self, sparse_inputs, dense_inputs):
        self.sync()
        ids = self._concat_ids(sparse_inputs)
        y1, y2, feat, _, _ = self._fm_fwd(ids, dense_inputs)
        y_dnn, _ = self.k.mlp_forward(feat.view(feat.shape[0], -1), self._mlp_weights()[0], self.mlp_b, self.ws_mlp)
        return torch.sigmoid(y1 + y2 + y_dnn)

    __call__ = forward

    # -- one full training step: train_forward + backward + optimizer.step ----------------------
    def _ensure_sparse_state(self):
        if self.sparse_state is None:
            D = self.sparse_feature_dim
            Dp = _round_up(D, 4)
            mv = torch.zeros(self.fm.rec.shape[0], _round_up(2 * Dp, 32), dtype=torch.float32,
                             device=self.device)
            self.sparse_state = dict(mv=mv, m=mv[:, :D], v=mv[:, Dp:Dp + D],
                                     m1=self.fm.rec[:, D + 1:D + 2], v1=self.fm.rec[:, D + 2:D + 3])

    lazy_mode = True   # False = the dygraph default (every row decays each step, 6*N*(D+1)*4 B of traffic)

    # -- launch-bound batches: the step as a recorded call list (plan.py) ------------------------------------------
    def _plan_eligible(self, sparse_inputs, dense_inputs, label, auc_stats, allreduce):
        return (self.device.type == "cuda" and self.k is ops and torch.is_tensor(sparse_inputs)
                and torch.is_tensor(dense_inputs) and torch.is_tensor(label) and allreduce is None and self.timers is None and self.lazy_mode and not self._recording
                and sparse_inputs.numel() <= int(os.environ.get("REC_STEP_PLAN_MAX", "65536"))
                and os.environ.get("REC_STEP_PLAN", "1") != "0")

    def _c_step_small(self, ids):
        from . import _lib
        return (ids.dim() == 2 and ids.numel() <= getattr(self.k, "SMALL_MERGE_MAX", 0)
                and self.n_linear <= _lib.DeepFMNet.MAX_LINEAR and hasattr(self.k, "deepfm_train_step")
                and os.environ.get("REC_SMALL_MERGE", "1") != "0" and os.environ.get("REC_SMALL_C_STEP", "1") != "0")

    def _train_step_planned(self, ids, dense_inputs, label, lr, auc_stats):
        from .plan import CallPlan
        inputs = [ids, dense_inputs, label]
        key = (tuple(ids.shape), None if auc_stats is None else (auc_stats[0].data_ptr(), auc_stats[1].data_ptr()))
        entry = self._plans.get(key)
        if entry is None:                       # first sight of a signature: plain eager step (sizes its buffers)
            self._plans[key] = "seen"
            return None
        if entry == "seen" or not entry.matches(inputs):
            plan = CallPlan()
            self._recording = True
            try:                                # recorded on ONE stream: nothing to join, nothing torch would issue
                out = plan.record(lambda: self.train_step(ids, dense_inputs, label, lr, auc_stats), inputs)
            finally:
                self._recording = False
            self._plans[key] = plan
            return out
        self.step_count += 1
        self._w_key = None
        return entry.replay(inputs, self.step_count, float(lr))

    # -- the step through the ONE-call C entry point (rec_deepfm_train_step) ---------------------------------------
    def c_net(self):
        """_lib.DeepFMNet over this layer's tensors (built once; the tensors live as long as the layer)."""
        net = getattr(self, "_c_net", None)
        if net is not None:
            return net
        from . import _lib
        if self.n_linear > _lib.DeepFMNet.MAX_LINEAR:
            raise ops.RecError("rec_deepfm_train_step takes at most %d Linear layers" % _lib.DeepFMNet.MAX_LINEAR)
        self._ensure_sparse_state()
        D, st = self.sparse_feature_dim, self.sparse_state
        p, g = self.dense.p, self.dense.g
        net = _lib.DeepFMNet()
        net.num_slots, net.dim, net.dense_dim, net.n_linear = self.sparse_num_field, D, self.dense_feature_dim, self.n_linear
        sizes = self.layer_sizes + [1]
        for i in range(self.n_linear):
            net.widths[i] = sizes[i]
            net.w[i], net.b[i] = self.mlp_w[i].data_ptr(), self.mlp_b[i].data_ptr()
            net.gw[i], net.gb[i] = self.mlp_dw[i].data_ptr(), self.mlp_db[i].data_ptr()
        net.table_rows, net.num_rows = self.fm.rec.shape[0], self.sparse_feature_number
        net.padding_idx = -1 if self.fm.padding_idx is None else self.fm.padding_idx
        net.slot_rows = self.slot_rows or 0
        net.slot_offset = None if self.fm.slot_offset is None else self.fm.slot_offset.data_ptr()
        net.rec, net.rec_stride = self.fm.rec.data_ptr(), self.fm.rec.stride(0)
        net.mv, net.mv_stride, net.v_offset = st["mv"].data_ptr(), st["mv"].stride(0), _round_up(D, 4)
        net.dense_w, net.dense_w_one = p["fm.dense_w"].data_ptr(), p["fm.dense_w_one"].data_ptr()
        net.g_dense_w, net.g_dense_w_one = g["fm.dense_w"].data_ptr(), g["fm.dense_w_one"].data_ptr()
        net.flat_param, net.flat_grad = self.dense.data.data_ptr(), self.dense.grad.data_ptr()
        net.flat_m, net.flat_v, net.flat_numel = self.dense.m.data_ptr(), self.dense.v.data_ptr(), self.dense.data.numel()
        net.w0_folded = self._w0p.data_ptr() if (self.compact or self.padded) else None
        net.layer0_width = self.ld0 if self.padded else 0
        self._c_net = net
        return net

    def train_step_c(self, sparse_inputs, dense_inputs, label, lr=1e-3, auc_stats=None):
        """train_step through rec_deepfm_train_step: ONE foreign call per step — what a non-Python binder of
        include/recengine.h gets.  Same kernels, arguments and order as train_step (with its side stream for the large
        batches): bit-identical."""
        self.sync()
        ids = self._concat_ids(sparse_inputs)
        self.step_count += 1
        if getattr(self, "_ws_c", None) is None:
            self._ws_c = self.k.Workspace(self.device)
        side = None
        if self.device.type == "cuda" and os.environ.get("REC_DEEPFM_OVERLAP", "1") != "0":
            if self._side is None:
                self._side = self.k.concurrent_stream(self.device)
            side = self._side
        self._w_key = None
        return self.k.deepfm_train_step(self.c_net(), ids, dense_inputs, label.reshape(-1), self.step_count, self._ws_c,
                                        lr=lr, auc_stats=auc_stats, num_thresholds=NUM_THRESHOLDS, status=self.status,
                                        side_stream=side)

    def train_step(self, sparse_inputs, dense_inputs, label, lr=1e-3, auc_stats=None,
                   allreduce=None):
        """dygraph_model.py:76-88 train_forward + tools/trainer.py:151-152 backward/step.
        label [B,1] int64.  Returns (loss [1] device tensor, pred [B,1])."""
        if self._plan_eligible(sparse_inputs, dense_inputs, label, auc_stats, allreduce):
            if self._c_step_small(sparse_inputs):
                # launch-bound sizes: the step through rec_deepfm_train_step, where the folds and the dense Adam ride
                # in the row update's launch and layer 0's weight fold in the lookup's (csrc/tail_roles.h: 16 -> 10
                # launches; bit-identical to the list below, tests/test_deepfm_step_c.py; REC_SMALL_C_STEP=0: the list)
                return self.train_step_c(sparse_inputs, dense_inputs, label, lr=lr, auc_stats=auc_stats)
            self.sync()
            out = self._train_step_planned(sparse_inputs, dense_inputs, label, lr, auc_stats)
            if out is not None:
                return out
        ids = self._concat_ids(sparse_inputs)
        B, S = ids.shape
        self._ensure_sparse_state()
        self.step_count += 1
        on_gpu = self.device.type == "cuda"
        cur = torch.cuda.current_stream() if on_gpu else None
        overlap = on_gpu and os.environ.get("REC_DEEPFM_OVERLAP", "1") != "0" and not self._recording
        if overlap and self._side is None:
            self._side = self.k.concurrent_stream(self.device)   # verified to overlap with the main stream
        side = self._side if overlap else None
        # one launch for every weight image of the step (bf16 x 3 GEMMs: M >= 8192 rows; REC_GEMM_IMAGES=0: each GEMM
        # splits its own weight as before)
        use_images = (on_gpu and B >= 8192 and hasattr(self.k, "GemmImages") and not self._recording
                      and os.environ.get("REC_GEMM_IMAGES", "1") != "0")
        groups = getattr(self, "_groups", None)          # persistent: the wait_stream below orders reuse
        if groups is None or groups.n != B * S:
            groups = self._groups = self.k.IdGroups(B * S, self.device)
        # SelectedRows merge keys only depend on ids: sort them on a side stream, hidden behind the MFMA-bound GEMMs.
        # REC_DEEPFM_GROUP_AT (measurement knob): "bwd" = under the dX chain (default: 2.83 ms/step), "fwd" = under
        # the forward GEMMs (2.875 ms; gpurun call 25) — the sort costs the GEMMs it runs beside about its own time
        # either way, the backward chain absorbs it slightly better
        # (round 3, weights aligned and the pipe kernels in the step: under the forward GEMMs is now the better place —
        # 2.311 against 2.32-2.34 ms in three A/B pairs on one box, profiles/r03_schedule_ab.txt)
        group_at = os.environ.get("REC_DEEPFM_GROUP_AT", "pre")

        # the reference's own batch sizes (config_bigdata.yaml: 512 x 26 = 13312 lookups): the merge happens inside the
        # ONE launch of the record update — no grouping sort, no hot-row partial passes (12 launches less per step)
        small = (self.lazy_mode and B * S <= getattr(self.k, "SMALL_MERGE_MAX", 0)
                 and hasattr(self.k, "sparse_adam_record_small")
                 and os.environ.get("REC_SMALL_MERGE", "1") != "0")

        # Slot-local grouping (round 4: rec_ids_group_slots, 7 launches instead of 13; REC_DEEPFM_GROUP=general: the one
        # 25-bit sort) — and, optionally, row gradients written in SORTED order through the rank it emits
        # (REC_DEEPFM_SORTED_GRAD=1: rec_deepfm_fm_bwd_sorted + rec_grad_layout.sorted).  Measured on MI355X
        # (profiles/r04_sorted_grad_ab.txt): the update kernel's traffic drops 1151 -> 1002 MB per launch (the floor of the
        # two-line record layout is ~985) and it runs 375 -> 340-358 us beside dW_0, but fm_bwd — on the critical path —
        # pays 62 -> 75 us for scattering 64-byte rows, and the step time is the same (2.25 ms both ways).  Default off:
        # the roofline kernels stay at their streaming speed.
        slot_sort = (not small and self.slot_rows is not None
                     and hasattr(self.k, "group_slots_eligible") and self.k.group_slots_eligible(B, S, self.slot_rows)
                     and os.environ.get("REC_DEEPFM_GROUP", "slots") != "general")
        sorted_rg = slot_sort and self.lazy_mode and os.environ.get("REC_DEEPFM_SORTED_GRAD", "0") == "1"
        # REC_DEEPFM_GROUP_CUS=<stride>[,range]: the grouping on its own stream confined to every stride-th CU ("k,r":
        # the first r CUs) — the 256x80 GEMM blocks fill the VGPR file (4 waves x 128 per SIMD), so a sort block can
        # only run where a GEMM block is not: confined to a few CUs it packs there instead of displacing GEMM blocks
        # all over the chip
        gspec = os.environ.get("REC_DEEPFM_GROUP_CUS", "")
        gside = side
        # pipelined: lazy Adam, sort-based merge, two streams, no collective in the tail
        pipe = (self.pipelined and side is not None and self.lazy_mode and not small and allreduce is None
                and B >= 16384 and os.environ.get("REC_DEEPFM_DEFER_ALL", "0") != "1")
        if pipe:                      # the side stream carries the previous step's dense tail: the grouping gets its own
            if self._gside is None:
                self._gside = self.k.concurrent_stream(self.device, index=1)
            gside = self._gside
        elif self._side_pending:
            self.sync()
        if gspec and side is not None and hasattr(self.k, "cu_stride_stream"):
            if "," in gspec:
                gside = self.k.cu_range_stream(self.device, 0, int(gspec.split(",")[1]))
            else:
                gside = self.k.cu_stride_stream(self.device, 0, int(gspec))

        def issue_group():
            if small:
                return
            if self.step_count > 3 and os.environ.get("REC_DEEPFM_SKIP_GROUP", "0") == "1":
                return      # MEASUREMENT ONLY: the step without its grouping sort (stale groups: wrong results)
            with _OnSide(gside, cur):
                if slot_sort:
                    self.k.ids_group_slots(ids, self.slot_rows, self.fm.padding_idx, self.ws_group, self.status, groups,
                                           want_rank=sorted_rg)
                else:
                    self.k.ids_group(ids, self.sparse_feature_number, self.fm.padding_idx, self.ws_group,
                                     self.fm.slot_offset, self.status, groups)
        if group_at == "pre":       # forked BEFORE fm_fwd: beside the HBM-bound lookup instead of under the forward GEMMs
            issue_group()
        if self._adam_done is not None and "adam" in os.environ.get("REC_PIPE_SKIP", ""):
            self._adam_done = None              # MEASUREMENT ONLY
        if self._adam_done is not None:     # pipelined: the lookup reads fm.dense_w / dense_w_one, which the previous step's
            cur.wait_event(self._adam_done)     # dense Adam (side stream) wrote — long done when the update in front of
            self._adam_done = None              # this point has finished, so the wait costs nothing
        if os.environ.get("REC_PIPE_JOIN", "early") == "early":
            self.sync()     # ONE cross-queue wait in front of the lookup instead of two (every such wait costs the main
            #                 stream ~20 us in the step, fired or not): the side chain ends when the update does
        with self._timed("fm_fwd"):
            y1, y2, feat, sum_emb, _ = self._fm_fwd(ids, dense_inputs)
        self.sync()     # pipelined: the previous step's dense tail (side stream) is needed from the first GEMM on
        if group_at == "fwd":
            issue_group()
        mlp_w, mlp_dw = self._mlp_weights(use_images)
        img_f, img_t = self._image_lists() if use_images else (None, None)
        nl = self.n_linear
        ikw_f = lambda n_: dict(images=img_f[:n_]) if img_f is not None else {}       # noqa: E731
        ikw_t = lambda n_: dict(images_t=img_t[:n_]) if img_t is not None else {}     # noqa: E731
        # the last Linear(-> 1), sigmoid + log_loss and the backward of both in ONE pass over the last hidden activation
        # (rec_ctr_head_fwd_bwd: five launches and two passes less on the critical path; REC_CTR_HEAD_FUSED=0: the parts)
        fused_head = self.n_linear > 1 and hasattr(self.k, "ctr_head") and self.k.ctr_head_ok(mlp_w[-1], mlp_dw[-1])
        with self._timed("mlp_fwd"):
            if fused_head:
                h, acts = self.k.mlp_forward(feat.view(B, -1), mlp_w[:-1], self.mlp_b[:-1], self.ws_mlp, relu_last=True,
                                             **ikw_f(nl - 1))
                pred, dz, loss, g_head = self.k.ctr_head(h, mlp_w[-1], self.mlp_b[-1], y1, y2, label, self.ws,
                                                         mlp_dw[-1], self.mlp_db[-1])
            else:
                y_dnn, acts = self.k.mlp_forward(feat.view(B, -1), mlp_w, self.mlp_b, self.ws_mlp, **ikw_f(nl))
        if not fused_head:
            pred, dz, loss = self.k.sigmoid_logloss(y1, y2, y_dnn, label, self.ws)
        if group_at == "bwd":
            issue_group()
        if auc_stats is not None:
            self.k.auc_histogram(pred, label, auc_stats[0], auc_stats[1], NUM_THRESHOLDS)
        defer_all = os.environ.get("REC_DEEPFM_DEFER_ALL", "0") == "1"   # measurement knob: every dW GEMM in the tail
        # small: one stream, nothing to run dW_0 beside — it goes out WITH dX_0 (ops.linear_backward: one launch at these sizes)
        kw = dict(defer_all=True) if defer_all else dict(defer_first=True) if not small else {}
        if overlap and not defer_all and not small and B >= 16384:
            # dW_0 runs beside the sparse update: half a resident round of blocks (K split 16 instead of 32) leaves the
            # HBM-bound kernel its wave slots — sparse_adam 404 -> 336 us, dW_0 unchanged (REC_DW0_SPLIT: 0 = planner's)
            kw.update(defer_split=int(os.environ.get("REC_DW0_SPLIT", "16")))
            # ... and, on the bf16 x 3 kernel (one 512-register block per CU), a grid for 192 of the 256 CUs: dW_0 248 ->
            # 292 us, the update 337 -> 324, the step 1.567-1.576 -> 1.548-1.557 ms (REC_DW0_CUS=0: every CU)
            kw.update(defer_cus=int(os.environ.get("REC_DW0_CUS", "192")))
        # dW_i on a third stream beside dX_i (both consume g_i, neither the other): the half-empty last round of
        # blocks of one GEMM is filled by the other — 2.77-2.84 -> 2.70-2.81 ms per step in five A/B pairs on two boxes
        # (profiles/r02f_dw_stream_ab.txt); REC_MLP_DW_STREAM=0 puts them back on one stream
        # Round 3: with the MLP weights aligned (every GEMM on its fast variant) the two streams LOSE — a dX / dW pair
        # takes 455 us side by side against 182 + 225 one after the other, 2.32-2.35 -> 2.27-2.32 ms per step — so one
        # stream is the default again and REC_MLP_DW_STREAM=1 the switch
        if on_gpu and not defer_all and not self._recording and os.environ.get("REC_MLP_DW_STREAM", "0") == "1":
            if getattr(self, "_dw_stream", None) is None:
                self._dw_stream, self._ws_dw = self.k.concurrent_stream(self.device), self.k.Workspace(self.device)
            kw.update(dw_stream=self._dw_stream, dw_ws=self._ws_dw)
        with self._timed("mlp_bwd"):
            if fused_head:
                res = self.k.mlp_backward(g_head, acts, mlp_w[:-1], mlp_dw[:-1], self.mlp_db[:-1], self.ws_mlp, **kw,
                                          **ikw_t(nl - 1))
            else:
                res = self.k.mlp_backward(dz, acts, mlp_w, mlp_dw, self.mlp_db, self.ws_mlp, **kw, **ikw_t(nl))
            d_flat, finish_dw0 = res if kw else (res, (lambda: None))
        if group_at == "tail":
            issue_group()
        if sorted_rg and side is not None:
            cur.wait_stream(gside)          # fm_bwd scatters through the rank the grouping (side stream) produced
        with self._timed("fm_bwd"):
            row_grad, _, _ = self.k.deepfm_fm_bwd(
                dense_inputs, feat, sum_emb, d_flat if self.padded else d_flat.view(B, self.fp, -1), dz, dz, S, self.ws,
                out=(self._row_grad_buf(B * S),
                     self.dense.g["fm.dense_w"].view(self.dense_feature_dim, -1),
                     self.dense.g["fm.dense_w_one"]),
                dense_w=self.dense.p["fm.dense_w"], compact=self.compact,
                **(dict(row_rank=groups.rank) if sorted_rg else {}),
                **(dict(feat_ld=self.ld0) if self.padded else {}))
        # The lazy sparse optimizer (HBM-bound) runs underneath the MFMA-bound dW_0 GEMM, one of the two on the side stream;
        # it needs row_grad / dz and the merge keys (sorted on the side stream earlier).
        t = self.step_count
        st = self.sparse_state
        skw = dict(grad_sorted=True) if sorted_rg else {}
        # Which of the two leaves the main stream: a kernel that waits for another QUEUE's event starts ~20 us after that
        # event (fm_bwd ends -> the update's first kernel: 22-26 us in the kernel traces).  REC_DEEPFM_TAIL_SWAP=1 puts
        # dW_0 + fold + dense Adam on the side stream instead, so that the wait sits in front of the shorter chain —
        # measured: 2.193-2.201 ms against 2.188-2.196 (profiles/r04_tail_swap.txt); the update stays on the side stream
        swap = (side is not None and allreduce is None and not small
                and os.environ.get("REC_DEEPFM_TAIL_SWAP", "0") == "1")
        if not sorted_rg and side is not None and (gside is not side or swap):
            cur.wait_stream(gside)

        def tail_sparse():
            with self._timed("sparse_adam"):
                upd = self.k.sparse_adam_rows if self.lazy_mode else self.k.adam_rows_all
                if small:
                    D = self.sparse_feature_dim
                    self.k.sparse_adam_record_small(ids, self.fm.slot_offset, self.fm.padding_idx, row_grad, dz, S,
                                                    self.fm.rec, st["mv"], D, t, lr, v_offset=_round_up(D, 4),
                                                    status=self.status)
                    return
                # hot rows (Zipf ids): long duplicate runs are pre-reduced per 64-position tile
                pp = self._pp = self.k.segment_partials(groups, row_grad, row_grad.shape[1],
                                                        out=getattr(self, "_pp", None), **skw)
                pp1 = self._pp1 = self.k.segment_partials(groups, dz, 1, grad_div=S, out=getattr(self, "_pp1", None))
                if self.lazy_mode:
                    # W, m, v and W1, m1, v1 of a row in ONE pass: W1 / m1 / v1 share the record line with W
                    D = self.sparse_feature_dim
                    self.k.sparse_adam_record(groups, row_grad, dz, S, self.fm.rec, st["mv"], D, t, lr,
                                              v_offset=_round_up(D, 4), partials=pp, partials1=pp1, **skw)
                elif hasattr(self.k, "adam_record_all") and os.environ.get("REC_NONLAZY_RECORD", "1") != "0":
                    # the dygraph default: every row moves — both embeddings and their moments in ONE sweep of the table
                    D = self.sparse_feature_dim
                    self.k.adam_record_all(groups, row_grad, dz, S, self.fm.rec, st["mv"], D, t, lr,
                                           v_offset=_round_up(D, 4), partials=pp, partials1=pp1)
                else:
                    upd(groups, row_grad, 1, self.fm.embedding, st["m"], st["v"], t, lr, partials=pp)
                    upd(groups, dz, S, self.fm.embedding_one, st["m1"], st["v1"], t, lr, partials=pp1)

        def tail_dense():
            with self._timed("mlp_bwd_dw0"):
                finish_dw0()
                self._fold_backward()
            if allreduce is not None:
                allreduce(self.dense.grad)
            self.k.adam_dense(self.dense.data, self.dense.m, self.dense.v, self.dense.grad, t, lr)
            self._w_key = None          # the folded layer-0 weight / the weight images belong to the old parameters

        if pipe:
            # main: merge + update, then — next call — the lookup; side: dW_0 -> fold -> dense Adam -> the NEXT step's
            # folded layer-0 weight and weight images, ordered behind fm_bwd by an event and ISSUED behind the update: the
            # kernel that reaches the chip first takes the wave slots — the HBM-bound update must be the resident one
            # that dW_0's blocks fill in beside (dW_0 first: 215 + 252 us one after the other; forked in front of
            # fm_bwd: the lookup's backward waits for the GEMM's blocks, 70 -> 248 us)
            after_bwd = torch.cuda.Event()
            after_bwd.record(cur)
            if "gwait" not in os.environ.get("REC_PIPE_SKIP", ""):       # MEASUREMENT ONLY (unordered: wrong results)
                cur.wait_stream(gside)
            tail_sparse()
            with torch.cuda.stream(side):
                side.wait_event(after_bwd)
                tail_dense()
                self._adam_done = torch.cuda.Event()
                self._adam_done.record()
                self._refresh_weights(use_images)
                self._w_key = (self.dense.data._version, self.step_count + 1, bool(use_images))
                self._side_pending = torch.cuda.Event()
                self._side_pending.record()
            self._pending_refs = (acts, finish_dw0, d_flat, feat)     # the side stream still reads them: freed at the join
            return loss, pred
        if swap:
            with _OnSide(side, cur):        # ordered behind fm_bwd, not behind the update issued below
                tail_dense()
            tail_sparse()
        else:
            with _OnSide(side, cur):
                tail_sparse()
            tail_dense()
            if side is not None and use_images and os.environ.get("REC_DEEPFM_REFRESH_TAIL", "1") != "0":
                # the NEXT step's folded layer-0 weight and weight images now, behind the dense Adam: the main stream
                # waits ~90 us for the side stream's table update at this point anyway (profiles/r06_step_timeline.txt),
                # and the next step's first GEMM then follows its lookup without the fold and split launches in between
                self._refresh_weights(use_images)
                self._w_key = (self.dense.data._version, self.step_count + 1, bool(use_images))
        if side is not None:
            cur.wait_stream(self._side)
        return loss, pred

    def _timed(self, name):
        return _Timed(self.timers, name)

    def time_fm_pair(self, batches, repeats=20, rounds=3):
        """MEASUREMENT ONLY (bench.py): the two FM kernels of a step, each launched `repeats` times back to back
        between ONE pair of HIP events on the current stream — the per-launch duration rocprofv3's kernel trace
        reports (plus the launch boundary), without the event markers and stream joins that bracket a kernel inside
        a step.  batches: list of (ids, dense); cycled so that a launch never re-reads the previous launch's table
        lines from cache.  Uses the step's own buffers (feat, d_feat, row_grad).  -> (fwd_us, bwd_us), medians."""
        ids0, dense0 = batches[0]
        ids0 = self._concat_ids(ids0)
        B, S = ids0.shape
        D = self.sparse_feature_dim
        y1, y2, feat, sum_emb, _ = self._fm_fwd(ids0, dense0)
        dfeat = torch.randn(*((B, self.ld0) if self.padded else (B, self.fp, D)), device=self.device) * 1e-3
        pkw = dict(feat_ld=self.ld0) if self.padded else {}
        dz = torch.randn(B, 1, device=self.device) * 1e-3
        out = (self._row_grad_buf(B * S), torch.empty(self.dense_feature_dim, D, device=self.device),
               torch.empty(self.dense_feature_dim, device=self.device))

        def fwd(i):
            ids, dense = batches[i % len(batches)]
            self.k.deepfm_fm_fwd(self._concat_ids(ids), dense, self.fm.embedding, self.fm.embedding_one,
                                 self.dense.p["fm.dense_w"], self.dense.p["fm.dense_w_one"], self.fm.padding_idx,
                                 self.fm.slot_offset, self.status, (y1, y2, feat, sum_emb), compact=self.compact, **pkw)

        # the backward as the STEP runs it: with the slot-local grouping its row gradients go out in sorted order through
        # the rank of the merge keys (rec_deepfm_fm_bwd_sorted) — the same variant is timed here
        rank = getattr(getattr(self, "_groups", None), "rank", None)
        rkw = dict(row_rank=rank) if rank is not None and rank.numel() >= B * S else {}

        def bwd(i):
            self.k.deepfm_fm_bwd(dense0, feat, sum_emb, dfeat, dz, dz, S, self.ws, out=out,
                                 dense_w=self.dense.p["fm.dense_w"], compact=self.compact, **rkw, **pkw)

        def run(fn):
            ts = []
            for _ in range(rounds):
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for i in range(repeats):
                    fn(i)
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) * 1e3 / repeats)
            return sorted(ts)[len(ts) // 2]
        return run(fwd), run(bwd)

    def _row_grad_buf(self, n):
        b = getattr(self, "_rg", None)
        if b is None or b.shape[0] != n:
            self._rg = torch.empty(n, self.sparse_feature_dim, dtype=torch.float32, device=self.device)
        return self._rg


def slot_feeds(batch_data, config, device):
    """create_feeds of the Criteo slot models (deepfm/dygraph_model.py:41-51, same code in fm / wide_deep / dcn_v2):
    -> (label [B,1] i64, sparse, dense [B,Dn] f32) on `device`.  batch_data is either the reference's 28 arrays
    [label, C1..C26, dense] (sparse = list of 26 [B,1] tensors) or the (label [B,1], ids [B,26], dense [B,13]) device
    tensors of paddlerec_amd.reader (sparse = the [B,26] tensor: no per-slot split and re-concat)."""
    if len(batch_data) == 3 and torch.is_tensor(batch_data[1]) and batch_data[1].dim() == 2 \
            and batch_data[1].shape[1] > 1:
        label, ids, dense = batch_data
        return label.to(device), ids.to(device), dense.to(device)
    dn = config.get("hyper_parameters.dense_input_dim")
    sparse = [torch.as_tensor(b).to(torch.int64).reshape(-1, 1).to(device) for b in batch_data[:-1]]
    dense = torch.as_tensor(batch_data[-1]).to(torch.float32).reshape(-1, dn).to(device)
    return sparse[0], sparse[1:], dense


def auc_metrics(device):
    """create_metrics: paddle.metric.Auc("ROC") = the two int64 bucket arrays of rec_auc_histogram, on the device."""
    stats = (torch.zeros(NUM_THRESHOLDS + 1, dtype=torch.int64, device=device),
             torch.zeros(NUM_THRESHOLDS + 1, dtype=torch.int64, device=device))
    return [stats], ["auc"]


class DygraphModel:
    """deepfm/dygraph_model.py:23-98 — same method names; tensors are torch device tensors."""

    def create_model(self, config, device="cuda", kernels=None, comm=None):
        """comm (paddlerec_amd.sharded.Comm, world > 1): the row-sharded layer of the collective mode."""
        args = (config.get("hyper_parameters.sparse_feature_number"), config.get("hyper_parameters.sparse_feature_dim"),
                config.get("hyper_parameters.dense_input_dim"), config.get("hyper_parameters.sparse_inputs_slots") - 1,
                config.get("hyper_parameters.fc_sizes"))
        if comm is not None and comm.world > 1:
            from .sharded import ShardedDeepFMLayer
            return ShardedDeepFMLayer(*args, device=device, comm=comm, kernels=kernels)
        return DeepFMLayer(*args, device=device, kernels=kernels)

    def create_feeds(self, batch_data, config, device="cuda"):
        return slot_feeds(batch_data, config, device)

    def create_metrics(self, device="cuda"):
        return auc_metrics(device)

    def train_forward(self, dy_model, metrics_list, batch_data, config, lr=None, next_batch=None):
        """next_batch (row-sharded layer only): the batch the next call will get — its ids are routed a step ahead."""
        label, sparse, dense = self.create_feeds(batch_data, config, dy_model.device)
        lr = lr if lr is not None else config.get("hyper_parameters.optimizer.learning_rate", 0.001)
        kw = {}
        if next_batch is not None and hasattr(dy_model, "comm"):
            kw["next_sparse_inputs"] = self.create_feeds(next_batch, config, dy_model.device)[1]
        loss, _ = dy_model.train_step(sparse, dense, label, lr, metrics_list[0] if metrics_list else None, **kw)
        return loss, metrics_list, {"loss": loss}

    def infer_forward(self, dy_model, metrics_list, batch_data, config):
        label, sparse, dense = self.create_feeds(batch_data, config, dy_model.device)
        pred = dy_model.forward(sparse, dense)
        if metrics_list:
            dy_model.k.auc_histogram(pred.contiguous(), label.contiguous(), metrics_list[0][0],
                                     metrics_list[0][1], NUM_THRESHOLDS)
        return metrics_list, None


def auc_from_buckets(stat_pos, stat_neg):
    """tools/utils/utils_single.py:183-204 — trapezoid sweep from the top bucket (host, fp64)."""
    pos_l = stat_pos.tolist()
    neg_l = stat_neg.tolist()
    area = pos = neg = 0.0
    total = 0
    for idx in range(len(pos_l) - 1, -1, -1):
        new_pos = pos + pos_l[idx]
        new_neg = neg + neg_l[idx]
        total += pos_l[idx] + neg_l[idx]
        area += (new_neg - neg) * (pos + new_pos) / 2
        pos, neg = new_pos, new_neg
    if pos * neg == 0 or total == 0:
        return 0.5
    return area / (pos * neg)
