"""paddle.static.nn.sparse_embedding / sequence_pool over LoD inputs (SURVEY App. B-7 [EXT]).

A lod_level=1 feed is a paddle.LoDTensor(values [nnz,1] int64, lod [B+1]).  sparse_embedding with
param_attr=ParamAttr(name=...) resolves to ONE parameter per name (slot_dnn/net.py:64-69: all 408 slots share
"embedding"); the PS table of the reference is an exact hash map keyed by the id, which a [size[0], size[1]] dense
parameter indexed by the id reproduces for ids < size[0].  padding_idx rows come back as zeros and get no gradient.
"""
import torch as _t

PARAMS = {}          # name -> Parameter  (the golden generator reads / resets this)


def sparse_embedding(input, size, padding_idx=None, is_test=False, entry=None, table_class="MemorySparseTable",  # noqa: A002
                     param_attr=None, dtype="float32", slot=None):
    from .. import LoDTensor
    name = getattr(param_attr, "name", None) or "sparse_embedding_%d" % len(PARAMS)
    if name not in PARAMS:
        p = _t.nn.Parameter(_t.zeros(int(size[0]), int(size[1])))
        init = getattr(param_attr, "initializer", None)
        if init is not None:
            with _t.no_grad():
                init(p)
        PARAMS[name] = p
    w = PARAMS[name]
    ids = input.values.reshape(-1)
    out = w[ids]
    if padding_idx is not None:
        out = out * (ids != padding_idx).unsqueeze(-1).to(out.dtype)
    return LoDTensor(out, input.lod)


def sequence_pool(input, pool_type, is_test=False, pad_value=0.0):  # noqa: A002
    assert pool_type.lower() == "sum"
    lod = [int(x) for x in input.lod]
    rows = [input.values[lod[b]:lod[b + 1]].sum(dim=0) if lod[b + 1] > lod[b]
            else _t.full((input.values.shape[1],), float(pad_value)) for b in range(len(lod) - 1)]
    return _t.stack(rows, dim=0)
