"""paddle.* — see paddlerec_amd/compat/__init__.py.  Symbol list: SURVEY.md Appendix A.1; semantics: Appendix B [EXT]."""
import pickle as _pickle

import numpy as _np
import torch as _t

from . import _backend
from . import nn, optimizer, metric, io, distributed, static, regularizer, framework, jit  # noqa: F401
from .framework import ParamAttr  # noqa: F401

Tensor = _t.Tensor
_DT = {"float32": _t.float32, "float64": _t.float64, "int64": _t.int64, "int32": _t.int32, "bool": _t.bool}


def _dtype(d):
    if isinstance(d, _t.dtype):
        return d
    return _DT[str(d).replace("paddle.", "")]


# Tensor method spellings that differ from torch (the reference calls them on batch fields and predictions)
_t.Tensor.astype = lambda self, d: self.to(_dtype(d))
_torch_numpy = _t.Tensor.numpy
_t.Tensor.numpy = lambda self, *a, **k: _torch_numpy(self.detach().cpu(), *a, **k)


def seed(s):
    _t.manual_seed(int(s))


def set_device(name):
    return _backend.set_device(name)


def get_device():
    d = _backend.device()
    return "gpu:%d" % (d.index or 0) if d.type == "cuda" else "cpu"


def is_compiled_with_custom_device(name):
    return False


def is_compiled_with_cuda():
    return _t.cuda.is_available()


class CPUPlace:
    pass


class CUDAPlace:
    def __init__(self, idx=0):
        self.idx = idx


def to_tensor(x, dtype=None, place=None, stop_gradient=True):
    if isinstance(x, _t.Tensor):
        t = x
    else:
        t = _t.as_tensor(_np.asarray(x))
    t = t.to(_backend.device())
    return t.to(_dtype(dtype)) if dtype is not None else t


class LoDTensor:
    """A lod_level=1 feed (`static.data(..., lod_level=1)`; Paddle's LoDTensor [EXT]): values [nnz, ...] of all segments
    of the batch back to back + the offsets lod [B+1] (device int64) — values and offsets as two dense tensors, which is
    also how the custom operator rec_multislot_sumpool takes them."""
    _n = 0

    def __init__(self, values, lod, name=None):
        dev = _backend.device()
        self.values = values.to(dev) if isinstance(values, _t.Tensor) else _t.as_tensor(_np.asarray(values)).to(dev)
        self.lod = (lod if isinstance(lod, _t.Tensor) else _t.as_tensor(_np.asarray(lod, _np.int64))).to(dev).to(_t.int64)
        LoDTensor._n += 1
        self.name = name or "lod_%d" % LoDTensor._n
        self.lod_level = 1

    @property
    def shape(self):
        return list(self.values.shape)

    @property
    def dtype(self):
        return self.values.dtype


def cast(x, dtype):
    if isinstance(x, LoDTensor):
        return LoDTensor(x.values.to(_dtype(dtype)), x.lod, x.name + ".cast")
    return x.to(_dtype(dtype))


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


def add(x, y):
    return x + y


def unsqueeze(x, axis):
    return x.unsqueeze(axis)


def squeeze(x, axis=None):
    return x.squeeze() if axis is None else x.squeeze(axis)


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


def clip(x, min=None, max=None):  # noqa: A002
    return _t.clamp(x, min=min, max=max)


def zeros(shape, dtype="float32"):
    return _t.zeros(list(shape), dtype=_dtype(dtype), device=_backend.device())


def ones(shape, dtype="float32"):
    return _t.ones(list(shape), dtype=_dtype(dtype), device=_backend.device())


def create_parameter(shape, dtype="float32", default_initializer=None, attr=None, is_bias=False):
    p = _t.nn.Parameter(_t.zeros(list(shape), dtype=_dtype(dtype), device=_backend.device()))
    init = default_initializer or (attr.initializer if attr is not None else None)
    if init is not None:
        with _t.no_grad():
            init(p)
    if attr is not None and getattr(attr, "regularizer", None) is not None:
        p._regularizer = attr.regularizer
    return p


def full_like(x, fill_value, dtype=None):
    return _t.full_like(x, fill_value, dtype=_dtype(dtype) if dtype is not None else None)


def enable_static():
    """paddle.enable_static(): from here on `static.data` placeholders are recorded on the main program's tape
    (compat/paddle/static/__init__.py)."""
    _backend._state["static"] = True


def disable_static():
    _backend._state["static"] = False


def in_dynamic_mode():
    return not _backend.static_mode()


def _ps_save(dirname, mode=0):
    """fleet.save_inference_model in gpubox mode: dense parameters + the EXISTING values of every sparse table
    (index + whole accessor record; Save(param = mode) decides which, UpdateStatAfterSave applied)."""
    import os as _os
    from . import static as _s
    _os.makedirs(dirname, exist_ok=True)
    prog = _s.default_main_program()
    K = _backend.kernels()
    out = {"dense.%d" % i: p.detach().cpu().numpy() for i, p in enumerate(prog.parameters())}
    for name, tab in prog.tables.items():
        sel = K.ps_save_select(tab.table, int(mode))
        rows = _t.nonzero(sel).reshape(-1)
        out["table.%s.rows" % name] = rows.cpu().numpy()
        out["table.%s.records" % name] = tab.table.rec[rows].cpu().numpy()
    _np.savez(_os.path.join(dirname, "rec_gpubox.npz"), **out)


def _ps_save_shards():
    """The pass checkpoint of an N-rank gpubox launch: the first worker's fleet.save_inference_model left a request
    (directory, mode); in fleet.barrier_worker() — which every rank reaches — each rank writes the existing values of
    ITS shard (global row ids + whole accessor records) and rank 0 merges the shards into the single rec_gpubox.npz a
    one-GPU run writes (same keys, rows ascending)."""
    import os as _os
    import torch.distributed as dist
    from . import _dist
    from . import static as _s
    c = _dist.comm()
    req = [_dist._state["pending_save"] if c.rank == 0 else None]
    dist.broadcast_object_list(req, src=dist.get_global_rank(c.group, 0), group=c.group)
    _dist._state["pending_save"] = None
    if req[0] is None:
        return
    dirname, mode = req[0]
    _os.makedirs(dirname, exist_ok=True)
    prog = _s.default_main_program()
    K = _backend.kernels()
    out = {}
    for name, tab in prog.tables.items():
        sel = K.ps_save_select(tab.table, int(mode))
        rows = _t.nonzero(sel).reshape(-1)
        out["table.%s.rows" % name] = (rows * c.world + c.rank).cpu().numpy()          # global row = local * G + rank
        out["table.%s.records" % name] = tab.table.rec[rows].cpu().numpy()
    _np.savez(_os.path.join(dirname, "rec_gpubox.shard%dof%d.npz" % (c.rank, c.world)), **out)
    dist.barrier(group=c.group)
    if c.rank == 0:
        merged = {"dense.%d" % i: p.detach().cpu().numpy() for i, p in enumerate(prog.parameters())}
        parts = [_np.load(_os.path.join(dirname, "rec_gpubox.shard%dof%d.npz" % (r, c.world))) for r in range(c.world)]
        for name in prog.tables:
            rows = _np.concatenate([p["table.%s.rows" % name] for p in parts])
            recs = _np.concatenate([p["table.%s.records" % name] for p in parts])
            order = _np.argsort(rows, kind="stable")
            merged["table.%s.rows" % name], merged["table.%s.records" % name] = rows[order], recs[order]
        _np.savez(_os.path.join(dirname, "rec_gpubox.npz"), **merged)


def save(obj, path):
    """paddle.save(state_dict, path): a pickled {name -> ndarray} dict (nested dicts of tensors are converted)."""
    def conv(o):
        if isinstance(o, _t.Tensor):
            return o.detach().cpu().numpy()
        if isinstance(o, dict):
            return {k: conv(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [conv(v) for v in o]
        return o
    with open(path, "wb") as f:
        _pickle.dump(conv(obj), f, protocol=4)


def load(path):
    with open(path, "rb") as f:
        return _pickle.load(f)


# Every tensor function of the namespace becomes static-aware: a call that sees a static Var is recorded on the main
# program's tape instead of executed (compat/paddle/static/__init__.py).
def _make_static_aware():
    import types
    from .static import static_aware
    from .nn import functional as _F
    skip = {"seed", "set_device", "get_device", "save", "load", "enable_static", "disable_static", "in_dynamic_mode",
            "is_compiled_with_custom_device", "is_compiled_with_cuda", "create_parameter", "to_tensor"}
    for mod in (globals(), vars(_F)):
        for name, fn in list(mod.items()):
            if isinstance(fn, types.FunctionType) and not name.startswith("_") and name not in skip:
                mod[name] = static_aware(fn)


_make_static_aware()
from . import incubate, utils  # noqa: E402,F401
