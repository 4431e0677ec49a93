import math as _m
import torch as _t
from . import functional, initializer  # noqa: F401


class Layer(_t.nn.Module):
    def add_sublayer(self, name, layer):
        # paddle semantics: same-name registration replaces the earlier entry (Appendix B-9)
        self._modules[name] = layer
        return layer

    def sublayers(self):
        return list(self.modules())[1:]


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


class Embedding(Layer):
    """out = 0 where id == padding_idx, else weight[id]; padding row gets no grad (App. B-1)."""

    def __init__(self, num_embeddings, embedding_dim, padding_idx=None, sparse=False,
                 weight_attr=None, name=None):
        super().__init__()
        self.padding_idx = padding_idx
        self.weight = _t.nn.Parameter(_t.empty(num_embeddings, embedding_dim))
        _init_from(weight_attr, self.weight, initializer.XavierUniform())
        if padding_idx is not None:
            with _t.no_grad():
                self.weight[padding_idx].zero_()

    def forward(self, ids):
        out = self.weight[ids]
        if self.padding_idx is not None:
            out = out * (ids != self.padding_idx).unsqueeze(-1).to(out.dtype)
        return out


class Linear(Layer):
    """y = x @ W + b with W [in, out] (App. B-2)."""

    def __init__(self, in_features, out_features, weight_attr=None, bias_attr=None, name=None):
        super().__init__()
        self.weight = _t.nn.Parameter(_t.empty(in_features, out_features))
        self.bias = _t.nn.Parameter(_t.zeros(out_features))
        _init_from(weight_attr, self.weight, initializer.XavierUniform())
        _init_from(bias_attr, self.bias, initializer.Constant(0.0))

    def forward(self, x):
        return _t.matmul(x, self.weight) + self.bias


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


class Conv2D(Layer):
    """NCHW convolution, weight [out, in, kh, kw] (xdeepfm/net.py:133-143 uses 1x1, bias_attr=False)."""

    def __init__(self, in_channels, out_channels, kernel_size, weight_attr=None, bias_attr=None, name=None):
        super().__init__()
        kh, kw = (kernel_size, kernel_size) if isinstance(kernel_size, int) else kernel_size
        self.weight = _t.nn.Parameter(_t.empty(out_channels, in_channels, kh, kw))
        _init_from(weight_attr, self.weight, initializer.XavierUniform())
        self.bias = None
        if bias_attr is not False:
            self.bias = _t.nn.Parameter(_t.zeros(out_channels))
            _init_from(bias_attr, self.bias, initializer.Constant(0.0))

    def forward(self, x):
        return _t.nn.functional.conv2d(x, self.weight, self.bias)


class BatchNorm1D(Layer):
    """[EXT Paddle batch_norm] momentum 0.9, epsilon 1e-5; training: batch mean / BIASED batch variance, running stats
    moved by (1 - momentum) towards them (biased variance there too); parameters weight, bias, _mean, _variance."""

    def __init__(self, num_features, momentum=0.9, epsilon=1e-05, weight_attr=None, bias_attr=None, name=None):
        super().__init__()
        self.momentum, self.epsilon = momentum, epsilon
        self.weight = _t.nn.Parameter(_t.ones(num_features))
        self.bias = _t.nn.Parameter(_t.zeros(num_features))
        self.register_buffer("_mean", _t.zeros(num_features))
        self.register_buffer("_variance", _t.ones(num_features))

    def forward(self, x):
        if self.training:
            mean = x.mean(dim=0)
            var = ((x - mean) ** 2).mean(dim=0)
            with _t.no_grad():
                self._mean.mul_(self.momentum).add_((1 - self.momentum) * mean)
                self._variance.mul_(self.momentum).add_((1 - self.momentum) * var)
        else:
            mean, var = self._mean, self._variance
        return (x - mean) / _t.sqrt(var + self.epsilon) * self.weight + self.bias
