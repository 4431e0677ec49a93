"""Minimal torch-CPU stand-in for the `paddle` symbols touched by the reference's
models/rank/{deepfm,dcn_v2,din}/net.py.

ORACLE-SIDE TEST INFRASTRUCTURE: it exists so that oracle/make_golden.py can execute the
reference's *unmodified* net.py files in this container (PaddlePaddle is not installable) and
record golden tensors.  Op semantics follow SURVEY.md Appendix B ([EXT] = public Paddle
behaviour).  It is never imported by paddlerec_amd/.
"""
import numpy as _np
import torch as _t

from . import nn, framework, regularizer, static, distributed  # noqa: F401
from .framework import ParamAttr  # noqa: F401

Tensor = _t.Tensor
_DT = {"float32": _t.float32, "float64": _t.float64, "int64": _t.int64, "int32": _t.int32,
       "bool": _t.bool}


def _dtype(d):
    if isinstance(d, _t.dtype):
        return d
    return _DT[str(d).replace("paddle.", "")]


# Tensor method spellings that differ from torch
_t.Tensor.astype = lambda self, d: self.to(_dtype(d))


def seed(s):
    _t.manual_seed(int(s))


def is_compiled_with_custom_device(name):
    return False


def to_tensor(x, dtype=None):
    t = _t.as_tensor(_np.asarray(x))
    return t.to(_dtype(dtype)) if dtype is not None else t


class LoDTensor:
    """lod_level=1 feed: values [nnz, ...] + offsets [B+1] (what paddle.static.data(..., lod_level=1) carries)."""
    _n = 0

    def __init__(self, values, lod, name=None):
        self.values, self.lod = values, lod
        LoDTensor._n += 1
        self.name = name or "lod_%d" % LoDTensor._n


def cast(x, dtype):
    if isinstance(x, LoDTensor):
        return LoDTensor(x.values.to(_dtype(dtype)), x.lod, x.name + ".cast")
    return x.to(_dtype(dtype))


def clip(x, min=None, max=None):  # noqa: A002
    return _t.clamp(x, min=min, max=max)


def concat(x, axis=0):
    return _t.cat(list(x), dim=int(axis))


def stack(x, axis=0):
    return _t.stack(list(x), dim=int(axis))


def reshape(x, shape):
    shape = [x.shape[i] if s == 0 else s for i, s in enumerate(shape)]  # 0 = copy input dim
    return x.reshape(shape)


def sum(x, axis=None, dtype=None, keepdim=False):  # noqa: A001
    if axis is None:
        return x.sum()
    return x.sum(dim=axis, keepdim=keepdim)


def mean(x, axis=None, keepdim=False):
    return x.mean() if axis is None else x.mean(dim=axis, keepdim=keepdim)


def square(x):
    return x * x


def multiply(x, y):
    return x * y


def unsqueeze(x, axis):
    return x.unsqueeze(axis)


def matmul(x, y, transpose_x=False, transpose_y=False):
    if transpose_x:
        x = x.transpose(-1, -2)
    if transpose_y:
        y = y.transpose(-1, -2)
    return _t.matmul(x, y)


def transpose(x, perm):
    return x.permute(*perm)


def scale(x, scale=1.0, bias=0.0):  # noqa: A001
    return x * scale + bias


def tanh(x):
    return _t.tanh(x)


def create_parameter(shape, dtype="float32", default_initializer=None, attr=None, is_bias=False):
    p = _t.nn.Parameter(_t.zeros(list(shape), dtype=_dtype(dtype)))
    init = default_initializer or (attr.initializer if attr is not None else None)
    if init is not None:
        with _t.no_grad():
            init(p)
    return p


def add(x, y):
    return x + y


def bmm(x, y):
    return _t.bmm(x, y)


def triu(x, diagonal=0):
    return _t.triu(x, diagonal)


def tril(x, diagonal=0):
    return _t.tril(x, diagonal)


def ones_like(x, dtype=None):
    return _t.ones_like(x) if dtype is None else _t.ones_like(x, dtype=_dtype(dtype))


def greater_than(x, y):
    return x > y


def masked_select(x, mask):
    return _t.masked_select(x, mask)
