"""paddle.optimizer — Adam / SGD over the engine's update kernels.

Dense parameters: rec_adam_dense / rec_sgd_dense per parameter.  Embedding parameters arrive as SelectedRows gradients
(ids + rows stashed by nn.Embedding's backward): merged with rec_ids_group and applied by rec_adam_rows_all (the dygraph
default lazy_mode=False: every row's moments decay, deepfm/dygraph_model.py:61-65) or
rec_sparse_adam_rows (lazy_mode=True) / rec_sparse_sgd_rows."""
import torch as _t

from . import _backend


class _Base:
    def __init__(self, learning_rate=0.001, parameters=None, weight_decay=None, grad_clip=None):
        self._lr = learning_rate
        self._params = [p for p in (parameters or [])]
        self._step = 0
        self._state = {}
        self._ws = None
        self._groups = {}

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
        """(groups, grad rows [n, D]) of the SelectedRows gradients stashed on an embedding parameter."""
        K = _backend.kernels()
        ids = _t.cat([g[0] for g in p._sparse_grads])
        rows = _t.cat([g[1] for g in p._sparse_grads]).contiguous()
        pad = p._sparse_grads[0][2]
        if self._ws is None:
            self._ws = K.Workspace(p.device)
        grp = self._groups.get(id(p))
        if grp is None or grp.n != ids.numel():
            grp = self._groups[id(p)] = K.IdGroups(ids.numel(), p.device)
        status = getattr(p, "_rec_status", None)
        K.ids_group(ids.contiguous(), p.shape[0], pad, self._ws, None, status, grp)
        return grp, rows

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
            for p in self._params:
                sparse = getattr(p, "_sparse_grads", None)
                if not sparse and p.grad is None:
                    continue
                st = self._state.setdefault(id(p), {"m": _t.zeros_like(p), "v": _t.zeros_like(p)})
                if sparse:
                    grp, rows = self._merged_keys(p)
                    upd = K.sparse_adam_rows if self._lazy else K.adam_rows_all
                    upd(grp, rows, 1, p.data, st["m"], st["v"], t, **kw)
                else:
                    K.adam_dense(p.data.view(-1), st["m"].view(-1), st["v"].view(-1), p.grad.contiguous().view(-1), t, **kw)


class SGD(_Base):
    def step(self):
        K = _backend.kernels()
        self._step += 1
        lr = self.get_lr()
        with _t.no_grad():
            for p in self._params:
                sparse = getattr(p, "_sparse_grads", None)
                if sparse:
                    grp, rows = self._merged_keys(p)
                    K.sparse_sgd_rows(grp, rows, p.data, lr)
                elif p.grad is not None:
                    K.sgd_dense(p.data.view(-1), p.grad.contiguous().view(-1), lr)
