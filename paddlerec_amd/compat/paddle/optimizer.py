"""paddle.optimizer — Adam / SGD over the engine's update kernels.

Dense parameters: rec_adam_dense / rec_sgd_dense per parameter.  Embedding parameters arrive as SelectedRows gradients
(ids + rows stashed by nn.Embedding's backward): merged with rec_ids_group and applied by rec_adam_rows_all (the dygraph
default lazy_mode=False: every row's moments decay, deepfm/dygraph_model.py:61-65) or
rec_sparse_adam_rows (lazy_mode=True) / rec_sparse_sgd_rows.

Order [EXT Optimizer._apply_optimize]: gradient clipping FIRST, then the regulariser, then the rule.
grad_clip = nn.ClipGradByGlobalNorm(clip_norm) (dcn_v2/dygraph_model.py:81-88): every gradient — the merged rows of
SelectedRows gradients included — is multiplied by clip_norm / max(global_norm, clip_norm): rec_sumsq /
rec_sparse_rows_sumsq / rec_clip_scale, the coefficient handed to the update kernels as a device scalar.
weight_decay / ParamAttr(regularizer=L2Decay(c)) (dcn_v2/net.py:164-170: L2Decay(1e-7) on the DNN weights; xdeepfm:
L2Decay(1e-4)): appended to the (clipped) GRADIENT [EXT append_regularization_ops] — g = scale * g + c * w on dense
parameters (rec_l2_decay_grad); a parameter's own regulariser wins over the optimizer's weight_decay."""
import torch as _t

from . import _backend


class _Base:
    def __init__(self, learning_rate=0.001, parameters=None, weight_decay=None, grad_clip=None):
        self._lr = learning_rate
        self._params = [p for p in (parameters or [])]
        self._weight_decay, self._grad_clip = weight_decay, grad_clip
        if grad_clip is not None and not hasattr(grad_clip, "clip_norm"):
            raise NotImplementedError("compat optimizer: grad_clip %r (only nn.ClipGradByGlobalNorm)" % (grad_clip,))
        self._step = 0
        self._state = {}
        self._ws = None
        self._groups = {}

    def minimize(self, loss, startup_program=None, parameters=None, no_grad_set=None):
        """Static graph (dnn/static_model.py:121-127): the loss variable and this optimizer become the main program's
        training target; Executor.train_from_dataset runs backward + step on every batch."""
        from . import static as _s
        if not isinstance(loss, _s.Var):
            raise NotImplementedError("optimizer.minimize(): dygraph code calls loss.backward(); optimizer.step()")
        _s.default_main_program().loss = loss
        _s.default_main_program().optimizer = self
        return None, None

    def _bind(self, params):
        if not self._params:
            self._params = list(params)

    def get_lr(self):
        lr = self._lr
        return float(lr() if callable(lr) else getattr(lr, "last_lr", lr))

    def clear_grad(self, set_to_zero=True):
        for p in self._params:
            p.grad = None
            if hasattr(p, "_sparse_grads"):
                p._sparse_grads = []

    clear_gradients = clear_grad

    def _merged_keys(self, p):
        """(groups, grad rows [n / div, D], div) of the SelectedRows gradients stashed on an embedding parameter: entries
        (ids, value, padding_idx, div) — one value row serves `div` consecutive ids (nn.Embedding: 1; the first-order
        table of the fused FM operator: one dy per sample for its S lookups)."""
        K = _backend.kernels()
        sg = p._sparse_grads
        if len(sg) == 1:
            ids, rows, div = sg[0][0], sg[0][1].contiguous(), sg[0][3]
        else:
            ids = _t.cat([g[0] for g in sg])
            rows = _t.cat([g[1] if g[3] == 1 else g[1].repeat_interleave(g[3], dim=0) for g in sg]).contiguous()
            div = 1
        pad = sg[0][2]
        if self._ws is None:
            self._ws = K.Workspace(p.device)
        grp = self._groups.get(id(p))
        if grp is None or grp.n != ids.numel():
            grp = self._groups[id(p)] = K.IdGroups(ids.numel(), p.device)
        status = getattr(p, "_rec_status", None)
        K.ids_group(ids.contiguous(), p.shape[0], pad, self._ws, None, status, grp)
        return grp, rows, div

    @staticmethod
    def _coeff(reg):
        if reg is None:
            return 0.0
        if isinstance(reg, (int, float)):
            return float(reg)
        if hasattr(reg, "coeff"):
            return float(reg.coeff)
        raise NotImplementedError("compat optimizer: regularizer %r (only L2Decay / a float)" % (reg,))

    def _prepare(self):
        """The global-norm clipping coefficient over the raw gradients, then the regulariser appended to the dense
        gradients (pre-divided by the coefficient, which the update kernels apply to the whole gradient).
        -> (sparse: {id(p): (groups, rows, div)}, scale: device float[1] or None)"""
        K = _backend.kernels()
        sparse = {}
        for p in self._params:
            if getattr(p, "_sparse_grads", None):
                if self._coeff(getattr(p, "_regularizer", None)):
                    raise NotImplementedError("compat optimizer: a regularizer on a sparse=True embedding")
                sparse[id(p)] = self._merged_keys(p)
        scale = None
        if self._grad_clip is not None:
            dev = self._params[0].device
            if self._ws is None:
                self._ws = K.Workspace(dev)
            ss = _t.zeros(1, dtype=_t.float32, device=dev)
            for p in self._params:
                if id(p) in sparse:
                    grp, rows, div = sparse[id(p)]
                    K.sparse_rows_sumsq(grp, rows, rows.shape[1], ss, self._ws, accumulate=True, grad_div=div)
                elif p.grad is not None:
                    K.sumsq(p.grad.contiguous().view(-1), ss, self._ws, accumulate=True)
            scale = K.clip_scale(ss, float(self._grad_clip.clip_norm), _t.empty_like(ss))
        for p in self._params:
            if id(p) not in sparse and p.grad is not None:
                c = self._coeff(getattr(p, "_regularizer", None) or self._weight_decay)
                if c:
                    g = p.grad.contiguous()
                    K.l2_decay_grad(g.view(-1), p.detach().contiguous().view(-1), c, scale)
                    p.grad = g
        return sparse, scale

    def state_dict(self):
        out = {"step": self._step}
        for i, p in enumerate(self._params):
            for k, v in self._state.get(id(p), {}).items():
                out["param%d.%s" % (i, k)] = v
        return out


class Adam(_Base):
    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8, parameters=None, weight_decay=None,
                 grad_clip=None, lazy_mode=False, name=None):
        super().__init__(learning_rate, parameters, weight_decay, grad_clip)
        self._b1, self._b2, self._eps, self._lazy = beta1, beta2, epsilon, lazy_mode

    def step(self):
        K = _backend.kernels()
        self._step += 1
        t, lr = self._step, self.get_lr()
        kw = dict(lr=lr, beta1=self._b1, beta2=self._b2, eps=self._eps)
        with _t.no_grad():
            merged, scale = self._prepare()
            kw["grad_scale"] = scale
            for p in self._params:
                sparse = id(p) in merged
                if not sparse and p.grad is None:
                    continue
                st = self._state.setdefault(id(p), {"m": _t.zeros_like(p), "v": _t.zeros_like(p)})
                if sparse:
                    grp, rows, div = merged[id(p)]
                    upd = K.sparse_adam_rows if self._lazy else K.adam_rows_all
                    upd(grp, rows, div, p.data, st["m"], st["v"], t, **kw)
                else:
                    K.adam_dense(p.data.view(-1), st["m"].view(-1), st["v"].view(-1), p.grad.contiguous().view(-1), t, **kw)


class SGD(_Base):
    def step(self):
        K = _backend.kernels()
        self._step += 1
        lr = self.get_lr()
        with _t.no_grad():
            merged, scale = self._prepare()
            for p in self._params:
                if id(p) in merged:
                    grp, rows, div = merged[id(p)]
                    K.sparse_sgd_rows(grp, rows if scale is None else rows * scale, p.data, lr, grad_div=div)
                elif p.grad is not None:
                    g = p.grad.contiguous().view(-1)
                    K.sgd_dense(p.data.view(-1), g if scale is None else g * scale, lr)


class _LRScheduler:
    """paddle.optimizer.lr.LRScheduler [EXT]: `last_lr` is what the optimizer reads; `step()` advances `last_epoch`.
    (The reference's tools/trainer.py never calls scheduler.step(): DIN trains at values[0], din/dygraph_model.py:64-73.)"""

    def __init__(self):
        self.last_epoch = 0
        self.last_lr = self.get_lr()

    def __call__(self):
        return self.last_lr

    def step(self, epoch=None):
        self.last_epoch = self.last_epoch + 1 if epoch is None else int(epoch)
        self.last_lr = self.get_lr()


class _PiecewiseDecay(_LRScheduler):
    """lr = values[i] while boundaries[i-1] <= last_epoch < boundaries[i] [EXT]."""

    def __init__(self, boundaries, values, last_epoch=-1, verbose=False):
        if len(values) != len(boundaries) + 1:
            raise ValueError("PiecewiseDecay: len(values) must be len(boundaries) + 1")
        self.boundaries, self.values = list(boundaries), [float(v) for v in values]
        super().__init__()

    def get_lr(self):
        for b, v in zip(self.boundaries, self.values):
            if self.last_epoch < b:
                return v
        return self.values[-1]


class lr:  # noqa: N801  (paddle.optimizer.lr namespace)
    LRScheduler = _LRScheduler
    PiecewiseDecay = _PiecewiseDecay
