"""paddle.nn — Layer containers and the two layers with a recengine kernel behind them:
    Embedding  -> rec_emb_gather (forward), SelectedRows gradient (ids + rows, merged by rec_ids_group in the optimizer)
    Linear     -> rec_gemm_f32 (bias in the epilogue; backward: dX GEMM, dW GEMM with the bias gradient fused)
(deepfm/net.py:62-86,142-174 and the same classes in the sibling nets).  The rest is torch glue."""
import torch as _t

from . import functional, initializer  # noqa: F401
from .. import _backend


class Layer(_t.nn.Module):
    def __call__(self, *args, **kwargs):
        # static graph: a layer applied to placeholder Vars is ONE recorded op (replayed on every batch)
        if _backend.static_mode():
            from ..static import has_var, record
            if has_var(args, kwargs):
                return record(super().__call__, args, kwargs)
        return super().__call__(*args, **kwargs)

    def add_sublayer(self, name, layer):
        # paddle semantics: same-name registration replaces the earlier entry
        self._modules[name] = layer
        return layer

    def sublayers(self):
        return list(self.modules())[1:]

    def set_dict(self, sd):
        own = self.state_dict()
        with _t.no_grad():
            for k, v in sd.items():
                if k in own:
                    own[k].copy_(_t.as_tensor(v).to(own[k].device).reshape(own[k].shape))

    set_state_dict = set_dict


class LayerList(_t.nn.ModuleList):
    def __init__(self, layers=None):
        super().__init__(list(layers) if layers is not None else None)


class ParameterList(_t.nn.ParameterList):
    def __init__(self, params=None):
        super().__init__(list(params) if params is not None else None)


def _init_from(attr, p, default):
    init = getattr(attr, "initializer", None) if attr is not None else None
    with _t.no_grad():
        (init or default)(p)
    reg = getattr(attr, "regularizer", None) if attr is not None else None
    if reg is not None:           # ParamAttr(regularizer=L2Decay(c)): applied by the optimizer (dcn_v2/net.py:164-170)
        p._regularizer = reg


class _EmbeddingFn(_t.autograd.Function):
    """out = 0 where id == padding_idx, else weight[id].  Backward = the SelectedRows gradient of
    lookup_table_v2_grad [EXT]: rows = the flattened ids (unmerged), value = d out — stashed on the parameter for the
    optimizer (rec_ids_group + sparse Adam), never scattered into a dense [N,D] gradient."""

    @staticmethod
    def forward(ctx, weight, ids, padding_idx):
        K = _backend.kernels()
        flat = ids.reshape(-1).contiguous()
        status = getattr(weight, "_rec_status", None)
        if status is None:
            status = weight._rec_status = K.new_status(weight.device)
        out, _ = K.emb_gather(flat, weight.detach(), padding_idx, status)
        ctx.save_for_backward(flat)
        ctx.weight, ctx.padding_idx = weight, padding_idx
        return out.reshape(*ids.shape, weight.shape[1])

    @staticmethod
    def backward(ctx, grad_out):
        (flat,) = ctx.saved_tensors
        w = ctx.weight
        if not hasattr(w, "_sparse_grads"):
            w._sparse_grads = []
        w._sparse_grads.append((flat, grad_out.reshape(flat.numel(), -1).contiguous(), ctx.padding_idx, 1))
        return None, None, None


class Embedding(Layer):
    def __init__(self, num_embeddings, embedding_dim, padding_idx=None, sparse=False, weight_attr=None, name=None):
        super().__init__()
        self.padding_idx = padding_idx
        self.weight = _t.nn.Parameter(_t.empty(num_embeddings, embedding_dim, device=_backend.device()))
        _init_from(weight_attr, self.weight, initializer.XavierUniform())
        if padding_idx is not None:
            with _t.no_grad():
                self.weight[padding_idx].zero_()
        self.weight._is_embedding = True

    def forward(self, ids):
        return _EmbeddingFn.apply(self.weight, ids, self.padding_idx)


class _LinearFn(_t.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        K = _backend.kernels()
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        y = K.gemm(x2, weight.detach(), _backend.workspace(), epilogue="bias", bias=bias.detach())
        ctx.save_for_backward(x2, weight)
        ctx.lead = x.shape[:-1]
        return y.reshape(*x.shape[:-1], weight.shape[1])

    @staticmethod
    def backward(ctx, gy):
        K = _backend.kernels()
        x2, weight = ctx.saved_tensors
        g2 = gy.reshape(-1, gy.shape[-1]).contiguous()
        ws = _backend.workspace()
        dx = K.gemm(g2, weight.detach(), ws, trans_b=True)
        dw = _t.empty_like(weight)
        db = _t.empty(weight.shape[1], dtype=weight.dtype, device=weight.device)
        K.gemm(x2, g2, ws, trans_a=True, out=dw, b_colsum=db)      # dW = X^T G, db = column sums of G (same pass)
        return dx.reshape(*ctx.lead, weight.shape[0]), dw, db


class Linear(Layer):
    """y = x @ W + b with W [in, out]."""

    def __init__(self, in_features, out_features, weight_attr=None, bias_attr=None, name=None):
        super().__init__()
        dev = _backend.device()
        self.weight = _t.nn.Parameter(_t.empty(in_features, out_features, device=dev))
        self.bias = _t.nn.Parameter(_t.zeros(out_features, device=dev))
        _init_from(weight_attr, self.weight, initializer.XavierUniform())
        _init_from(bias_attr, self.bias, initializer.Constant(0.0))

    def forward(self, x):
        return _LinearFn.apply(x, self.weight, self.bias)


class ReLU(Layer):
    def forward(self, x):
        return _t.relu(x)


class Sigmoid(Layer):
    def forward(self, x):
        return _t.sigmoid(x)


class Dropout(Layer):
    def __init__(self, p=0.5):
        super().__init__()
        self.p = p

    def forward(self, x):
        return _t.nn.functional.dropout(x, self.p, self.training)


class Conv1D(Layer):  # imported by din/net.py:13, never used
    pass


class ClipGradByGlobalNorm:
    """paddle.nn.ClipGradByGlobalNorm(clip_norm) (dcn_v2/dygraph_model.py:81-88): passed to an optimizer as grad_clip;
    the optimizer multiplies every gradient by clip_norm / max(global_norm, clip_norm) (paddle/optimizer.py here)."""

    def __init__(self, clip_norm, group_name="default_group"):
        self.clip_norm = float(clip_norm)
