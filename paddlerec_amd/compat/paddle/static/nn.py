"""paddle.static.nn: the PS / gpubox lookups of dnn/net.py:67-82 and slot_dnn/net.py:61-75 [EXT semantics: SURVEY App. B-7,8].

sparse_embedding(input, size=[N, D+2], param_attr=ParamAttr(name=...)) -> [*, D+2] = [show, click, embed_w, embedx...] of
every key from the GPU-PS table (an ops.PsTable shared by all calls that name the same parameter: one table for the 26
slots); continuous_value_model(emb, show_click, use_cvm=False) strips the two CVM columns."""
import torch as _t

from .. import _backend


class _PsLookup(_t.autograd.Function):
    @staticmethod
    def forward(ctx, keys, anchor, table):
        K = _backend.kernels()
        t = table.table
        flat = keys.reshape(-1).contiguous()
        rows = K.feasign_rows(flat, t.num_rows)
        D = t.emb_dim
        w, _ = K.emb_gather(rows, t.W, None, table.status)
        st, _ = K.emb_gather(rows, t.rec[:, t.stat.start:t.stat.start + 2], None, table.status)   # show, click
        ctx.table, ctx.rows, ctx.D = table, rows, D
        return _t.cat([st, w], dim=1).reshape(*keys.shape[:-1], D + 2)

    @staticmethod
    def backward(ctx, g):
        g2 = g.reshape(-1, ctx.D + 2)[:, 2:].contiguous()
        ctx.table.pending.append((ctx.rows, g2))
        return None, None, None


class SparseTable:
    """The GPU-PS table behind static.nn.sparse_embedding: an ops.PsTable (uint64 feasigns hashed to its rows on the
    device) + the SelectedRows gradients of the lookups of the current step."""

    def __init__(self, name, num_rows, emb_dim, accessor=None):
        K = _backend.kernels()
        self.name = name
        self.table = K.PsTable(int(num_rows), int(emb_dim), _backend.device(), kind="slot", **(accessor or {}))
        self.status = K.new_status(_backend.device())
        self.anchor = _t.zeros(1, device=_backend.device(), requires_grad=True)   # gives the lookup a grad_fn
        self.pending, self.groups, self.ws = [], None, K.Workspace(_backend.device())

    def push(self, label):
        """After loss.backward(): merge the step's SelectedRows and apply the accessor's push (show = 1 per
        occurrence, click = the sample's label, gradient of the SUMMED loss)."""
        if not self.pending:
            return
        K = _backend.kernels()
        S, B = len(self.pending), self.pending[0][0].numel()
        rows = _t.stack([r for r, _ in self.pending], dim=1).reshape(-1).contiguous()            # position = b * S + s
        grad = _t.stack([g for _, g in self.pending], dim=1).reshape(B * S, -1).contiguous()
        self.pending = []
        if self.groups is None or self.groups.n != rows.numel():
            self.groups = K.IdGroups(rows.numel(), rows.device)
        # padding_idx 0 is on the ROW: rec_feasign_rows maps feasign 0 — and only feasign 0 — to row 0 (every other key
        # lands in [1, N)), so this drops exactly the padding key and never a real feature
        K.ids_group(rows, self.table.num_rows, 0, self.ws, None, self.status, self.groups)
        self.table.accessor.grad_scale = float(B)
        click = label.reshape(-1).to(_t.int64).contiguous() if label is not None else None
        K.ps_push_rows(self.table, self.groups, grad, S, click=click)


def sparse_embedding(input, size, padding_idx=None, is_test=False, entry=None, table_class="MemorySparseTable",  # noqa: A002
                     param_attr=None, dtype="float32", slot=None):
    from . import Var, _main, record
    name = getattr(param_attr, "name", None) or "embedding"
    tab = _main.tables.get(name)
    if tab is None:
        import os
        # the reference's size[0] is ignored by the GPU-PS (a hash map keyed by feasign [EXT]); the engine's table is a
        # hashed array: REC_GPUBOX_TABLE_ROWS rows (default 1 000 003), the looked-up vector is size[1] - 2 floats
        rows = int(os.environ.get("REC_GPUBOX_TABLE_ROWS", "1000003"))
        tab = _main.tables[name] = SparseTable(name, rows, int(size[1]) - 2,
                                               accessor=getattr(_main, "accessor_kwargs", None))

    def lookup(keys):
        return _PsLookup.apply(keys, tab.anchor, tab)
    lookup.__qualname__ = "static.nn.sparse_embedding[%s]" % name
    if isinstance(input, Var):
        return record(lookup, [input])
    return lookup(input)


def continuous_value_model(input, cvm, use_cvm=True):  # noqa: A002
    from . import has_var, record

    def cvm_op(x, c):
        if use_cvm:
            return _t.cat([_t.log(x[..., 0:1] + 1), _t.log(x[..., 1:2] + 1) - _t.log(x[..., 0:1] + 1), x[..., 2:]], -1)
        return x[..., 2:]
    cvm_op.__qualname__ = "static.nn.continuous_value_model"
    if has_var([input, cvm]):
        return record(cvm_op, [input, cvm])
    return cvm_op(input, cvm)


def sequence_pool(input, pool_type, is_test=False, pad_value=0.0):  # noqa: A002
    raise NotImplementedError("static.nn.sequence_pool over LoD feeds: use paddlerec_amd.gpubox (BenchmarkDNNLayer) for "
                              "the multi-value slot_dnn model; the tape executor serves dnn/config_gpubox.yaml")


def embedding(input, size, is_sparse=False, padding_idx=None, param_attr=None, dtype="float32"):  # noqa: A002
    return sparse_embedding(input, [size[0], size[1] + 2], param_attr=param_attr)


def fc(x, size, activation=None, name=None):
    raise NotImplementedError("static.nn.fc: the reference's rank models build their MLPs from paddle.nn.Linear")
