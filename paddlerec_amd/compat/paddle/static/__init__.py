"""paddle.static over the engine — the surface /root/reference/tools/static_gpubox_trainer.py and the static_model.py
files of its models touch, so that the reference's gpubox entry point runs UNMODIFIED:

    python -m paddlerec_amd.run_reference tools/static_gpubox_trainer.py -m models/rank/dnn/config_gpubox.yaml

Paddle builds a Program once and an Executor runs it per batch.  Here a Program is a TAPE: under paddle.enable_static()
`static.data` returns a placeholder `Var` (it carries a tiny example batch), every `paddle.*` / `paddle.nn.functional.*`
/ `paddle.static.nn.*` function and every `nn.Layer` call that sees a Var is executed once on the example values and
RECORDED (function, arguments, outputs); `Executor.train_from_dataset` replays the tape on each batch of the dataset
under torch autograd — the Linear layers run rec_gemm_f32, `static.nn.sparse_embedding` pulls from an ops.PsTable
(rec_feasign_rows + rec_emb_gather), the backward's SelectedRows go through rec_ids_group + rec_ps_push_rows (show = 1,
click = label: dnn/static_model.py:86-94), the dense parameters through the compat Adam, `static.auc` through
rec_auc_histogram.  Layers are constructed once (at record time), so parameters persist across replays like Paddle's
scope variables.  This is host glue around the same kernels as paddlerec_amd.gpubox, not a graph compiler."""
import os as _os

import numpy as _np
import torch as _t

from .. import _backend
from . import nn  # noqa: F401  (paddle.static.nn)


class Var:
    """A static-graph variable: name + an example value (what shape / dtype inference needs) + its value during replay."""

    _count = 0

    def __init__(self, example, name=None, persistable=False):
        Var._count += 1
        self.name = name or "tmp_%d" % Var._count
        self.example = example
        self.persistable = persistable
        self.stop_gradient = False
        self.value = example if persistable else None

    @property
    def shape(self):
        return list(self.example.shape)

    @property
    def dtype(self):
        return self.example.dtype

    def __repr__(self):
        return "var %s : %s%s" % (self.name, str(self.dtype).replace("torch.", ""), self.shape)

    # arithmetic on variables (wide_deep/static_model.py:99 `1 - pred`): one recorded elementwise op each
    def _binary(self, other, fn, name):
        fn.__name__ = fn.__qualname__ = name
        return record(fn, (self, other))

    def __add__(self, o):
        return self._binary(o, lambda a, b: a + b, "elementwise_add")

    __radd__ = __add__

    def __sub__(self, o):
        return self._binary(o, lambda a, b: a - b, "elementwise_sub")

    def __rsub__(self, o):
        return self._binary(o, lambda a, b: b - a, "elementwise_sub")

    def __mul__(self, o):
        return self._binary(o, lambda a, b: a * b, "elementwise_mul")

    __rmul__ = __mul__

    def __truediv__(self, o):
        return self._binary(o, lambda a, b: a / b, "elementwise_div")

    def __neg__(self):
        return self._binary(-1.0, lambda a, b: a * b, "scale")


class Program:
    def __init__(self):
        self.ops, self.feeds = [], []
        self.loss, self.optimizer = None, None
        self.tables = {}            # sparse_embedding tables by parameter name
        self.stateful = []          # nodes with persistent state (auc)
        self._fleet_opt = {}

    def __str__(self):
        lines = ["program (recengine tape) {"] + ["  feed " + repr(v) for v in self.feeds]
        for fn, args, kwargs, outs in self.ops:
            lines.append("  %s = %s(...)" % (", ".join(o.name for o in _flat_vars(outs)) or "-",
                                             getattr(fn, "__qualname__", getattr(fn, "__name__", type(fn).__name__))))
        return "\n".join(lines + ["}"])

    def parameters(self):
        seen, out = set(), []
        for fn, _, _, _ in self.ops:
            mod = getattr(fn, "__self__", None)
            if isinstance(mod, _t.nn.Module):
                for p in mod.parameters():
                    if id(p) not in seen:
                        seen.add(id(p))
                        out.append(p)
        return out


_main, _startup = Program(), Program()


def default_main_program():
    return _main


def default_startup_program():
    return _startup


def _flat_vars(x):
    if isinstance(x, Var):
        return [x]
    if isinstance(x, (list, tuple)):
        return [v for e in x for v in _flat_vars(e)]
    if isinstance(x, dict):
        return [v for e in x.values() for v in _flat_vars(e)]
    return []


def _subst(x, key):
    if isinstance(x, Var):
        return getattr(x, key)
    if isinstance(x, list):
        return [_subst(e, key) for e in x]
    if isinstance(x, tuple):
        return tuple(_subst(e, key) for e in x)
    if isinstance(x, dict):
        return {k: _subst(e, key) for k, e in x.items()}
    return x


def _wrap(out):
    if isinstance(out, _t.Tensor):
        return Var(out.detach())
    from .. import LoDTensor
    if isinstance(out, (LoDTensor, nn.LodEmbedding)):       # LoD values travel through the tape as they are
        return Var(out, name=out.name)
    if isinstance(out, (list, tuple)):
        return type(out)(_wrap(o) for o in out)
    return out


def has_var(args, kwargs=None):
    return bool(_flat_vars(list(args))) or bool(kwargs and _flat_vars(list(kwargs.values())))


def record(fn, args, kwargs=None):
    """Run fn once on the example values, put (fn, args, kwargs, outputs) on the main program's tape."""
    kwargs = kwargs or {}
    with _t.no_grad():
        out = fn(*_subst(list(args), "example"), **_subst(kwargs, "example"))
    outs = _wrap(out)
    _main.ops.append((fn, list(args), kwargs, outs))
    return outs


def static_aware(fn):
    """Decorator of the namespace's functions: a call that sees a Var is recorded instead of executed."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        if _backend.static_mode() and has_var(args, kwargs):
            return record(fn, args, kwargs)
        return fn(*args, **kwargs)
    return wrapper


def data(name, shape, dtype="float32", lod_level=0):
    """A feed variable: the example batch has 2 rows (None / -1 dims)."""
    from .. import _dtype
    shp = [2 if (s is None or s < 0) else int(s) for s in shape]
    ex = _t.zeros(shp, dtype=_dtype(dtype), device=_backend.device())
    if lod_level:                  # a multi-value feed: the example is two segments of one value each
        from .. import LoDTensor
        ex = LoDTensor(ex, list(range(shp[0] + 1)), name)
    v = Var(ex, name=name)
    v.lod_level = lod_level
    _main.feeds.append(v)
    return v


def create_global_var(shape, value, dtype, persistable=False, force_cpu=False, name=None):
    """paddle.static.create_global_var [EXT]: a non-trainable variable of the program filled with `value` — what a
    custom operator that keeps state across steps is handed (rec_ps_pull's record table).  One tensor per name."""
    from .. import _dtype
    name = name or "global_var_%d" % len(_persistables)
    for v in _persistables:
        if v.name == name:
            return v
    t = _t.full([int(s) for s in shape], value, dtype=_dtype(dtype),
                device="cpu" if force_cpu else _backend.device())
    t._rec_var_name = name
    v = Var(t, name=name, persistable=True)
    v.stop_gradient = True
    _persistables.append(v)
    return v


_parameters = {}


def create_parameter(shape, dtype, name=None, attr=None, is_bias=False, default_initializer=None):
    """paddle.static.create_parameter [EXT]: a trainable parameter of the program (zeros unless an initializer is given;
    one per name).  Returned as a persistable Var whose value is the torch Parameter."""
    from .. import create_parameter as _cp
    name = name or getattr(attr, "name", None) or "parameter_%d" % len(_parameters)
    if name not in _parameters:
        p = _cp(shape, dtype, default_initializer=default_initializer, attr=attr, is_bias=is_bias)
        v = Var(p, name=name, persistable=True)
        _persistables.append(v)
        _parameters[name] = v
    return _parameters[name]


def cpu_places(n=1):
    return ["cpu"] * n


class _ScopeVar:
    def __init__(self, var):
        self._v = var

    def get_tensor(self):
        t = self._v.value if self._v.value is not None else self._v.example
        return _TensorHandle(self._v, t)


class _TensorHandle:
    def __init__(self, var, t):
        self._var, self._t = var, t

    def __array__(self, dtype=None, copy=None):
        a = self._t.detach().cpu().numpy()
        return a.astype(dtype) if dtype is not None else a

    def _get_dims(self):
        return list(self._t.shape)

    def set(self, array, place=None):
        self._t.copy_(_t.as_tensor(_np.asarray(array)).to(self._t.device).reshape(self._t.shape))


class _Scope:
    def find_var(self, name):
        for v in _persistables:
            if v.name == name:
                return _ScopeVar(v)
        return None


_persistables = []
_scope = _Scope()


def global_scope():
    return _scope


def auc(input, label, curve="ROC", num_thresholds=2 ** 12 - 1, topk=1, slide_steps=1):  # noqa: A002
    """paddle.static.auc [EXT] (dnn/static_model.py:101-108): bucket statistics [1, num_thresholds + 1] int64 kept in
    persistable variables (the trainer reads stat_pos / stat_neg by name through global_scope and zeroes them per
    epoch); -> (auc, batch_auc, [batch_stat_pos, batch_stat_neg, stat_pos, stat_neg])."""
    dev = _backend.device()
    mk = lambda n: Var(_t.zeros(1, num_thresholds + 1, dtype=_t.int64, device=dev), name=n, persistable=True)
    k = len(_persistables)
    stats = [mk("_generated_var_%d" % (k + i)) for i in range(4)]      # batch_pos, batch_neg, pos, neg
    _persistables.extend(stats)

    def auc_op(pred, lab):
        K = _backend.kernels()
        p1 = pred[:, 1:2].contiguous().detach()
        lab = lab.reshape(-1, 1).contiguous()
        for pos, neg in ((stats[0], stats[1]), (stats[2], stats[3])):
            K.auc_histogram(p1, lab, pos.value.view(-1), neg.value.view(-1), num_thresholds)
        return _auc_value(stats[2].value, stats[3].value), _auc_value(stats[0].value, stats[1].value)
    auc_op.__qualname__ = "static.auc"
    a, b = record(auc_op, [input, label])
    return a, b, stats


def _auc_value(pos, neg):
    p, n = pos.view(-1).double().flip(0), neg.view(-1).double().flip(0)      # from the top bucket down
    cp, cn = _t.cumsum(p, 0), _t.cumsum(n, 0)
    area = ((cn - (cn - n)) * ((cp - p) + cp) / 2).sum()
    tot = cp[-1] * cn[-1]
    return (area / tot if float(tot) > 0 else _t.tensor(0.5, dtype=_t.float64)).reshape(1)


class Executor:
    """paddle.static.Executor: run(startup) is a no-op (parameters are initialised when their layer is constructed);
    train_from_dataset replays the main program's tape on every batch of the dataset."""

    def __init__(self, place=None):
        self.place = place

    def run(self, program=None, feed=None, fetch_list=None):
        if program is _startup or program is None or not program.ops:
            return []
        vals = self._forward(program, feed or {})
        return [(_subst(v, "value")).detach().cpu().numpy() for v in (fetch_list or [])] if vals is None else []

    def _forward(self, program, feed):
        for v in program.feeds:
            if v.name not in feed:
                raise KeyError("feed variable %r is missing from the batch" % v.name)
            v.value = feed[v.name]
        for fn, args, kwargs, outs in program.ops:
            out = fn(*_subst(args, "value"), **_subst(kwargs, "value"))
            _assign(outs, out)

    def train_from_dataset(self, program=None, dataset=None, scope=None, thread=0, debug=False, fetch_list=None,
                           fetch_info=None, print_period=100):
        program = program or _main
        if program.loss is None or program.optimizer is None:
            raise RuntimeError("train_from_dataset: optimizer.minimize(loss) was not called on this program")
        opt = program.optimizer
        params = program.parameters()
        opt._bind(params)
        from .. import _dist
        comm = _dist.comm()
        G = comm.world if comm is not None else 1
        plan = self._lookup_plan(program) if G > 1 else {}
        init_dump = _os.environ.get("REC_COMPAT_DUMP_INIT")
        if init_dump and not getattr(program, "_init_dumped", False):     # tests: the parameters an oracle replay starts from
            program._init_dumped = True
            if _dist.rank() == 0:
                _np.savez(init_dump, **{"dense.%d" % i: p.detach().cpu().numpy() for i, p in enumerate(params)})
        n = 0
        for feed in dataset._batches(_backend.device()):
            global_batch = feed.pop("__global_batch__", None)
            local_batch = int(next(iter(feed.values())).shape[0])
            if G > 1:
                # N ranks: ONE pull of the step's keys through the row-sharded tables (collective, also for a rank whose
                # share of a short last batch is empty), the tape on the local share, the loss weighted so that the
                # summed gradients are those of the mean over the GLOBAL batch
                for name, tab in program.tables.items():
                    tab.prefetch(_t.cat([feed[k].reshape(local_batch, -1)[:, :1] for k in plan[name]], dim=1))
                opt.clear_grad()
                if local_batch:
                    self._forward(program, feed)
                    loss = program.loss.value * (float(local_batch) / float(global_batch))
                    loss.backward()
                else:
                    loss = _t.zeros((), device=_backend.device())
                label = feed.get("label", feed.get("click"))
                if label is None and program.tables:
                    raise RuntimeError("train_from_dataset: the program has a sparse table but feeds neither 'label' "
                                       "nor 'click' (the click counter of the pushed features)")
                for tab in program.tables.values():
                    tab.push(label, global_batch)
                flat = _t.cat([(p.grad if p.grad is not None else _t.zeros_like(p)).reshape(-1) for p in params])
                comm.all_reduce_sum(flat)                                  # ONE all-reduce of the dense gradients
                o = 0
                for p in params:
                    k = p.numel()
                    p.grad = flat[o:o + k].view_as(p).clone()
                    o += k
                opt.step()
                n += 1
                continue
            self._forward(program, feed)
            loss = program.loss.value
            opt.clear_grad()
            loss.backward()
            # the click column of the pushed show / click pair: the label feed (dnn/static_model.py:86-94 casts `label`;
            # slot_dnn names it `click`) — never a silent "0 clicks" when the program feeds neither
            label = feed.get("label", feed.get("click"))
            if label is None and program.tables:
                raise RuntimeError("train_from_dataset: the program has a sparse table but feeds neither 'label' nor "
                                   "'click' (the click counter of the pushed features)")
            for tab in program.tables.values():
                tab.push(label)
            opt.step()
            n += 1
            if debug and n % print_period == 0:
                print("batch %d loss %.6f" % (n, float(loss)))
        self.last_loss = float(loss.detach()) if n else None
        return n

    @staticmethod
    def _lookup_plan(program):
        """{table name: feed variable names of its sparse_embedding calls, in tape order}: what lets N ranks pull all
        keys of a step in ONE exchange.  Every lookup of a sharded table must read a feed variable directly."""
        feeds = {id(v): v.name for v in program.feeds}
        plan = {name: [] for name in program.tables}
        for fn, args, _, _ in program.ops:
            name = getattr(fn, "_rec_table", None)
            if name is None:
                continue
            src = args[0] if args else None
            if id(src) not in feeds:
                raise NotImplementedError("multi-GPU gpubox: sparse_embedding[%s] reads a computed variable; the "
                                          "row-sharded pull needs the feasigns of a step up front (feed variables)" % name)
            plan[name].append(feeds[id(src)])
        return plan

    def infer_from_dataset(self, program=None, dataset=None, **kw):
        program = program or _main
        with _t.no_grad():
            for feed in dataset._batches(_backend.device()):
                self._forward(program, feed)


def _assign(outs, out):
    if isinstance(outs, Var):
        outs.value = out
    elif isinstance(outs, (list, tuple)):
        for o, x in zip(outs, out):
            _assign(o, x)


def save(program, path):
    import pickle
    with open(path + ".pdparams", "wb") as f:
        pickle.dump({"param%d" % i: p.detach().cpu().numpy() for i, p in enumerate(program.parameters())}, f)


def load(*a, **k):
    raise NotImplementedError("paddle.static.load: not on the gpubox training path")


def load_program_state(*a, **k):
    raise NotImplementedError("paddle.static.load_program_state: not on the gpubox training path")


def set_program_state(*a, **k):
    raise NotImplementedError("paddle.static.set_program_state: not on the gpubox training path")


def save_inference_model(path_prefix, feed_vars, fetch_vars, executor, program=None):
    save(program or _main, path_prefix)
