"""paddle.utils.cpp_extension — the loader of Paddle custom C++ operators.

    rec_ops = paddle.utils.cpp_extension.load(name="rec_ops", sources=[".../rec_paddle_ops.cc"], ...)
    y1, y2, feat, _, _ = rec_ops.rec_deepfm_fm(ids, dense, W, W1, dense_w, dense_w_one, padding_idx=0)

In PaddlePaddle `load` JIT-compiles the sources against paddle/extension.h and returns a module with one Python function
per PD_BUILD_OP (inputs positionally in the declared order, then the attributes; differentiable when a PD_BUILD_GRAD_OP
exists).  Here the same sources were compiled by paddlerec_amd/build.py against the executable stand-in header
(paddlerec_amd/paddle_ops/mock/paddle/extension.h) into paddle_ops/librec_paddle_ops.so; this module binds its `pd_mock_*`
entry points with ctypes and gives every registered operator
  * a forward that hands the op's kernel function device pointers of the torch tensors (the stream is torch's current
    stream; `paddle::empty` inside the kernel allocates through a callback into torch's caching allocator),
  * a torch.autograd.Function built from the grad op's declared Inputs / Outputs (forward inputs and outputs by name,
    `X@GRAD` = incoming / outgoing gradients),
  * SelectedRows for table inputs: a grad output the shim marks `selected_rows=<table>:<ids>` is the rows-form value; it
    is stashed on the parameter together with the ids, exactly as nn.Embedding's backward does, and the compat optimizer
    merges + applies it (rec_ids_group + sparse Adam / SGD kernels),
  * a shape / dtype check of every forward and gradient call against the op's own InferShape / InferDtype functions.
There is no CPU implementation behind it.  Tests that run the host logic without a GPU select an operator stand-in
module with REC_COMPAT_KERNELS (as for every other compat operator); it must provide CUSTOM_OPS[name] = (fwd, bwd)
with the kernels' calling convention, and the shim is then still loaded for the registry and the infer functions."""
import ctypes as C
import os
import threading

import torch

from .. import _backend

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.dirname(os.path.dirname(os.path.dirname(_HERE)))                    # paddlerec_amd/
SHIM_LIB = os.path.join(_PKG, "paddle_ops", "librec_paddle_ops.so")
SHIM_SRC = os.path.join(_PKG, "paddle_ops", "rec_paddle_ops.cc")

# paddle::DataType of the stand-in header <-> torch
_DT = {1: torch.bool, 2: torch.uint8, 3: torch.int8, 4: torch.int16, 5: torch.int32, 6: torch.int64, 7: torch.float16,
       8: torch.bfloat16, 9: torch.float32, 10: torch.float64}
_DT_REV = {v: k for k, v in _DT.items()}
_MAX_OUT = 16


class _Tensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("shape", C.c_int64 * 8), ("ndim", C.c_int32), ("dtype", C.c_int32),
                ("device", C.c_int32), ("reserved", C.c_int32), ("stream", C.c_void_p), ("handle", C.c_int64)]


class _Attr(C.Structure):
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("i", C.c_int64), ("f", C.c_double), ("s", C.c_char_p)]


_ALLOC = C.CFUNCTYPE(C.c_int64, C.POINTER(C.c_int64), C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double),
                     C.POINTER(C.c_void_p))


class _OpDef:
    def __init__(self, index, text):
        f = text.split("|")
        split = lambda s: [x for x in s.split(",") if x]
        self.index, self.name, self.is_grad = index, f[0], f[1] == "grad"
        self.inputs, self.outputs = split(f[2]), split(f[3])
        self.attrs = [tuple(p.strip() for p in a.split(":", 1)) for a in split(f[4])]        # (name, c++ type)
        self.has_kernel, self.has_shape, self.has_dtype = (c == "1" for c in f[5])
        self.notes = [n for n in f[6].split(";") if n]
        self.selected_rows = {}                                  # table input -> ids (a forward input or output)
        self.ps_tables = []                                      # inputs that are PS record tables updated in place
        for n in self.notes:
            if n.startswith("selected_rows="):
                t, i = n[len("selected_rows="):].split(":")
                self.selected_rows[t] = i
            elif n.startswith("ps_table="):
                self.ps_tables.append(n[len("ps_table="):])


class _Shim:
    """librec_paddle_ops.so: registry + run / infer entry points of the stand-in header."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise RuntimeError("custom-op shim %s is not built (python -m paddlerec_amd.build)" % path)
        if _backend.device().type == "cuda":
            from paddlerec_amd import _lib
            _lib.lib()                  # the engine (and torch's HIP runtime) first: one librecengine.so in the process
        self.lib = lib = C.CDLL(path)
        lib.pd_mock_op_describe.restype = C.c_char_p
        lib.pd_mock_last_error.restype = C.c_char_p
        self._live = threading.local()
        self._cb = _ALLOC(self._alloc)                                                      # keep the callback alive
        lib.pd_mock_set_allocator(self._cb)
        self.fwd, self.grad = {}, {}
        for i in range(lib.pd_mock_op_count()):
            d = _OpDef(i, lib.pd_mock_op_describe(i).decode())
            (self.grad if d.is_grad else self.fwd)[d.name] = d

    # -- paddle::empty / full inside a kernel -> torch's caching allocator (stream-ordered: safe for workspaces)
    def _alloc(self, shape, ndim, dtype, device, fill, data):
        try:
            shp = [int(shape[i]) for i in range(ndim)]
            dev = torch.device("cpu") if device < 0 else torch.device("cuda", device)
            if fill:
                t = torch.full(shp, fill[0], dtype=_DT[dtype], device=dev)
            else:
                t = torch.empty(shp, dtype=_DT[dtype], device=dev)
            live = self._live.tensors
            live.append(t)
            data[0] = t.data_ptr()
            return len(live)
        except Exception as e:                              # never let an exception cross the C frames
            self._live.error = e
            return 0

    @staticmethod
    def _desc(t, stream):
        d = _Tensor()
        d.data, d.ndim, d.dtype = t.data_ptr(), t.dim(), _DT_REV[t.dtype]
        for i, s in enumerate(t.shape):
            d.shape[i] = s
        d.device = t.device.index if t.device.type == "cuda" else -1
        d.stream, d.handle = stream, 0
        return d

    def _pack(self, op, inputs, attrs):
        """inputs: one tensor (or list of tensors for an X@VECTOR input) per declared input."""
        if len(inputs) != len(op.inputs):
            raise TypeError("%s takes %d inputs (%s), got %d" % (op.name, len(op.inputs), ", ".join(op.inputs), len(inputs)))
        flat, counts = [], []
        for name, x in zip(op.inputs, inputs):
            xs = list(x) if name.endswith("@VECTOR") else [x]
            for t in xs:
                if not isinstance(t, torch.Tensor):
                    raise TypeError("%s: input %s must be a Tensor" % (op.name, name))
                if not t.is_contiguous():
                    raise ValueError("%s: input %s is not contiguous" % (op.name, name))
            flat += xs
            counts.append(len(xs))
        dev = next((t.device for t in flat if t.device.type == "cuda"), None)
        stream = torch.cuda.current_stream(dev).cuda_stream if dev is not None else None
        tin = (_Tensor * max(1, len(flat)))(*[self._desc(t, stream) for t in flat])
        cnt = (C.c_int32 * max(1, len(counts)))(*counts)
        at = (_Attr * max(1, len(op.attrs)))()
        keep = []
        for k, (an, ty) in enumerate(op.attrs):
            v = attrs[an]
            if ty in ("float", "double"):
                at[k].kind, at[k].f = 1, float(v)
            elif ty == "std::string":
                keep.append(str(v).encode())
                at[k].kind, at[k].s = 2, keep[-1]
            elif ty == "bool":
                at[k].kind, at[k].i = 3, int(bool(v))
            elif ty == "std::vector<float>":
                arr = (C.c_float * max(1, len(v)))(*[float(x) for x in v])
                keep.append(arr)
                at[k].kind, at[k].i, at[k].s = 4, len(v), C.cast(arr, C.c_char_p)
            else:
                at[k].kind, at[k].i = 0, int(v)
        return flat, tin, cnt, at, stream, keep

    def infer(self, op, inputs, attrs):
        """-> [(shape | None, dtype | None)] per output from the op's InferShape / InferDtype, or None when it has none."""
        _, tin, cnt, at, _, _keep = self._pack(op, inputs, attrs)
        out = (_Tensor * _MAX_OUT)()
        n = C.c_int32(0)
        rc = self.lib.pd_mock_op_infer(op.index, tin, cnt, len(op.inputs), at, len(op.attrs), out, _MAX_OUT, C.byref(n))
        if rc == -2:
            return None
        if rc != 0:
            raise RuntimeError("%s infer: %s" % (op.name, self.lib.pd_mock_last_error().decode()))
        return [(None if out[i].ndim < 0 else tuple(out[i].shape[j] for j in range(out[i].ndim)), _DT.get(out[i].dtype))
                for i in range(n.value)]

    def run(self, op, inputs, attrs):
        flat, tin, cnt, at, stream, _keep = self._pack(op, inputs, attrs)
        self._live.tensors, self._live.error = [], None
        out = (_Tensor * _MAX_OUT)()
        n = C.c_int32(0)
        rc = self.lib.pd_mock_op_run(op.index, tin, cnt, len(op.inputs), at, len(op.attrs), stream, out, _MAX_OUT,
                                     C.byref(n))
        live, err = self._live.tensors, self._live.error
        self._live.tensors = []
        if err is not None:
            raise err
        if rc != 0:
            raise RuntimeError("%s: %s" % (op.name, self.lib.pd_mock_last_error().decode()))
        res = []
        for i in range(n.value):
            h = out[i].handle
            if h > 0:
                res.append(live[h - 1])
            else:                                            # a kernel returned one of its inputs
                res.append(next(t for t in flat if t.data_ptr() == out[i].data))
        return res


_shim = None
_lock = threading.Lock()


def shim():
    global _shim
    with _lock:
        if _shim is None:
            _shim = _Shim(os.environ.get("REC_PADDLE_OPS_LIB", SHIM_LIB))
    return _shim


def _standin(name):
    """The operator stand-in of a GPU-less test run (REC_COMPAT_KERNELS), or None: the product path is the shim."""
    if not os.environ.get("REC_COMPAT_KERNELS"):
        return None
    table = getattr(_backend.kernels(), "CUSTOM_OPS", None)
    if table is None or name not in table:
        raise NotImplementedError("REC_COMPAT_KERNELS module has no CUSTOM_OPS[%r]" % name)
    return table[name]


def _check_infer(s, op, inputs, attrs, outs):
    want = s.infer(op, inputs, attrs)
    if want is None:
        return
    for name, (shape, dtype), t in zip(op.outputs, want, outs):
        if shape is not None and tuple(t.shape) != shape:
            raise RuntimeError("%s: output %s has shape %s, InferShape says %s" % (op.name, name, tuple(t.shape), shape))
        if dtype is not None and op.has_dtype and t.dtype != dtype:
            raise RuntimeError("%s: output %s has dtype %s, InferDtype says %s" % (op.name, name, t.dtype, dtype))


class _CustomOp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fwd, grad, attrs, *inputs):
        s = shim()
        alt = _standin(fwd.name)
        det = [t.detach() for t in inputs]      # X@VECTOR (list) inputs never reach here: op() runs them without a tape
        outs = list(alt[0](det, attrs)) if alt is not None else s.run(fwd, det, attrs)
        _check_infer(s, fwd, det, attrs, outs)
        ctx.fwd, ctx.grad, ctx.attrs, ctx.n_in = fwd, grad, attrs, len(inputs)
        ctx.params = inputs                                   # the Parameter objects (SelectedRows are stashed on them)
        ctx.save_for_backward(*det, *outs)
        ctx.mark_non_differentiable(*[o for o in outs if not o.is_floating_point()])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        fwd, grad, attrs = ctx.fwd, ctx.grad, ctx.attrs
        saved = ctx.saved_tensors
        ins, outs = saved[:ctx.n_in], saved[ctx.n_in:]
        by_name = dict(zip(fwd.inputs, ins))
        by_name.update(zip(fwd.outputs, outs))
        for name, o, g in zip(fwd.outputs, outs, gouts):
            by_name[name + "@GRAD"] = torch.zeros_like(o) if g is None else g.contiguous()
        gin = [by_name[n] for n in grad.inputs]
        s = shim()
        alt = _standin(fwd.name)
        gattrs = {an: attrs[an] for an, _ in grad.attrs}
        res = list(alt[1](gin, gattrs)) if alt is not None else s.run(grad, gin, gattrs)
        _check_infer(s, grad, gin, gattrs, res)
        out = [None] * ctx.n_in
        for name, g in zip(grad.outputs, res):
            assert name.endswith("@GRAD"), name
            k = fwd.inputs.index(name[:-5])
            if name[:-5] in grad.selected_rows:               # rows-form: SelectedRows(rows = ids, value = g)
                p = ctx.params[k]
                if not ctx.needs_input_grad[3 + k]:           # a frozen table (stop_gradient) gets no update
                    continue
                ids = by_name[grad.selected_rows[name[:-5]]].reshape(-1)
                if g.shape[0] == 0 or ids.numel() % g.shape[0] != 0:
                    if ids.numel() or g.shape[0]:
                        raise RuntimeError("%s: %d ids cannot share %d gradient rows" % (fwd.name, ids.numel(), g.shape[0]))
                    continue
                if not hasattr(p, "_sparse_grads"):
                    p._sparse_grads = []
                p._sparse_grads.append((ids, g, attrs.get("padding_idx"), ids.numel() // g.shape[0]))
            elif ctx.needs_input_grad[3 + k]:
                out[k] = g.reshape(ins[k].shape)
        return (None, None, None) + tuple(out)


def _adopt_ps_table(op, rec, attrs):
    """An operator input marked `ps_table` (rec_ps_pull's Rec) is a GPU-PS record table the gradient operator updates in
    place.  The first call lists the tensor — a persistable variable made with paddle.static.create_global_var — among
    the program's sparse tables under the variable's name, so that the pass checkpoint (fleet.save_*), shrink and
    core.PSGPU see it exactly like a table static.nn.sparse_embedding created."""
    from .. import static as S
    name = getattr(rec, "_rec_var_name", None)
    if name is None or name in S._main.tables:
        return
    from ..static import nn as SN
    a = [float(x) for x in attrs["accessor"]]
    acc = dict(lr=a[0], initial_g2sum=a[1], bounds=(a[2], a[3]), initial_range=a[4], embedx_lr=a[5],
               embedx_initial_g2sum=a[6], embedx_bounds=(a[7], a[8]), embedx_initial_range=a[9], embedx_threshold=a[10],
               nonclk_coeff=a[11], click_coeff=a[12], seed=int(a[13]))
    S._main.tables[name] = SN.SparseTable(name, rec.shape[0], int(attrs["emb_dim"]), accessor=acc, rec=rec)
    print("[compat] custom operator %s: variable %r [%d, %d] listed as a GPU-PS table of the program"
          % (op.name, name, rec.shape[0], rec.shape[1]), flush=True)


class _OpModule:
    """What `load` returns: one function per registered forward operator."""

    def __init__(self, name):
        self.__name__ = name
        from ..static import static_aware
        s = shim()
        for op in s.fwd.values():
            setattr(self, op.name, static_aware(self._make(op, s.grad.get(op.name))))

    @staticmethod
    def _make(fwd, grad):
        def op(*args, **kwargs):
            n = len(fwd.inputs)
            inputs, extra = list(args[:n]), list(args[n:])
            attrs = {}
            for (an, _), v in zip(fwd.attrs, extra):
                attrs[an] = v
            for an, _ in fwd.attrs:
                if an in kwargs:
                    attrs[an] = kwargs.pop(an)
                if an not in attrs:
                    raise TypeError("%s: missing attribute %s" % (fwd.name, an))
            if kwargs:
                raise TypeError("%s: unexpected arguments %s" % (fwd.name, sorted(kwargs)))
            inputs = [x if isinstance(x, torch.Tensor) else list(x) for x in inputs]
            for tname in fwd.ps_tables:                      # the record table joins the program's sparse tables
                _adopt_ps_table(fwd, inputs[fwd.inputs.index(tname)], attrs)
            has_list = any(not isinstance(x, torch.Tensor) for x in inputs)
            if has_list and grad is not None and torch.is_grad_enabled() and any(
                    t.requires_grad for x in inputs for t in (x if isinstance(x, list) else [x])):
                raise NotImplementedError("%s: X@VECTOR inputs are not differentiable through the compat loader" % fwd.name)
            if grad is None or not torch.is_grad_enabled() or has_list:
                alt = _standin(fwd.name)
                det = [t.detach() if isinstance(t, torch.Tensor) else [u.detach() for u in t] for t in inputs]
                outs = list(alt[0](det, attrs)) if alt is not None else shim().run(fwd, det, attrs)
            else:
                outs = list(_CustomOp.apply(fwd, grad, attrs, *inputs))
            return outs[0] if len(outs) == 1 else outs
        op.__name__ = fwd.name
        op.__doc__ = "custom op %s(%s%s) -> %s" % (fwd.name, ", ".join(fwd.inputs),
                                                  "".join(", %s: %s" % a for a in fwd.attrs), ", ".join(fwd.outputs))
        return op


_modules = {}


def load(name, sources=None, extra_cxx_cflags=None, extra_cuda_cflags=None, extra_ldflags=None, extra_include_paths=None,
         build_directory=None, verbose=False, **kwargs):
    """paddle.utils.cpp_extension.load.  Only the engine's operator library is loadable here (no Paddle toolchain to
    JIT other sources with): `sources`, when given, must name paddle_ops/rec_paddle_ops.cc — the file the prebuilt shim
    was compiled from."""
    for src in sources or []:
        if os.path.basename(str(src)) != os.path.basename(SHIM_SRC):
            raise NotImplementedError("compat cpp_extension.load: cannot JIT %r (only %s, prebuilt by "
                                      "paddlerec_amd.build)" % (src, os.path.basename(SHIM_SRC)))
    if name not in _modules:
        _modules[name] = _OpModule(name)
    return _modules[name]


def get_build_directory():
    return os.path.dirname(SHIM_LIB)
