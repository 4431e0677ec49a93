"""Checkpoint / resume — mirrors tools/utils/save_load.py:25-47.

    save_model(net, optimizer_state, model_path, epoch_id, prefix='rec')   -> {model_path}/{epoch_id}/rec.pdparams, rec.pdopt
    load_model(model_path, net, prefix='rec')

`rec.pdparams` is a pickled {parameter name -> float32 ndarray} dict under the reference's state_dict keys
 — the container `paddle.save(state_dict)` writes for a dygraph Layer [EXT: cannot be verified
against Paddle here; a maintainer can `paddle.load` it or feed the dict to `set_dict`].  The engine's internal
record layout never leaks: tables are exported as the reference's dense [N,D] / [N,1] parameters.
`rec.pdopt` holds the optimizer state the host mirrors keep (step count, dense moments, sparse moments).
Row-sharded models (paddlerec_amd/sharded.py) write one file per rank, `rec.shard{r}of{G}.pdparams`, holding the
rows r, r+G, r+2G, ... of each table plus the replicated dense parameters; `load_model` re-shards on load, so a
checkpoint written by G ranks can be resumed by any G'.
"""
import os
import pickle

import numpy as np
import torch

TABLE_KEYS = ("fm.embedding.weight", "fm.embedding_one.weight", "embedding.weight")


def _np(t):
    return t.detach().cpu().numpy().copy() if isinstance(t, torch.Tensor) else np.asarray(t)


def _mkdir_if_not_exist(path):
    os.makedirs(path, exist_ok=True)         # every rank of a sharded run creates the same epoch directory


def optimizer_state(net):
    """What `optimizer.state_dict()` is for the host mirrors: step + Adam moments (dense flat buffer, sparse rows)."""
    st = {"step": int(getattr(net, "step_count", 0))}
    dense = getattr(net, "dense", None)
    if dense is not None:
        st["dense.m"], st["dense.v"] = _np(dense.m), _np(dense.v)
    sp = getattr(net, "sparse_state", None)
    if sp:
        for k in ("m", "v", "m1", "v1"):
            if k in sp:
                st["sparse." + k] = _np(sp[k])
    return st


def set_optimizer_state(net, st):
    net.step_count = int(st.get("step", 0))
    dense = getattr(net, "dense", None)
    if dense is not None and "dense.m" in st:
        dense.m.copy_(torch.as_tensor(st["dense.m"]).to(dense.m.device))
        dense.v.copy_(torch.as_tensor(st["dense.v"]).to(dense.v.device))
    if any(k.startswith("sparse.") for k in st):
        net._ensure_sparse_state()
        for k in ("m", "v", "m1", "v1"):
            if "sparse." + k in st and k in net.sparse_state:
                dst = net.sparse_state[k]
                dst.copy_(torch.as_tensor(st["sparse." + k]).to(dst.device).reshape(dst.shape))


def save_model(net, optimizer, model_path, epoch_id, prefix="rec"):
    """tools/utils/save_load.py:25-31.  `optimizer` may be None (state is read from the net) or a dict."""
    model_path = os.path.join(model_path, str(epoch_id))
    _mkdir_if_not_exist(model_path)
    model_prefix = os.path.join(model_path, prefix)
    comm = getattr(net, "comm", None)
    sd = {k: _np(v) for k, v in net.state_dict().items()}
    opt = optimizer if isinstance(optimizer, dict) else optimizer_state(net)
    if comm is not None and comm.world > 1:
        tag = ".shard%dof%d" % (comm.rank, comm.world)
        sd["__shard__"] = np.asarray([comm.rank, comm.world, net.global_rows], np.int64)
        with open(model_prefix + tag + ".pdparams", "wb") as f:
            pickle.dump(sd, f, protocol=4)
        with open(model_prefix + tag + ".pdopt", "wb") as f:
            pickle.dump(opt, f, protocol=4)
    else:
        with open(model_prefix + ".pdparams", "wb") as f:
            pickle.dump(sd, f, protocol=4)
        with open(model_prefix + ".pdopt", "wb") as f:
            pickle.dump(opt, f, protocol=4)
    return model_path


def _load_global(model_prefix):
    """Reads either the single file or all rank shards and returns the GLOBAL state dict."""
    single = model_prefix + ".pdparams"
    if os.path.exists(single):
        with open(single, "rb") as f:
            return pickle.load(f)
    d = os.path.dirname(model_prefix)
    base = os.path.basename(model_prefix)
    shards = sorted(f for f in os.listdir(d) if f.startswith(base + ".shard") and f.endswith(".pdparams"))
    if not shards:
        raise FileNotFoundError(single)
    parts = []
    for fn in shards:
        with open(os.path.join(d, fn), "rb") as f:
            parts.append(pickle.load(f))
    world = int(parts[0]["__shard__"][1])
    if len(parts) != world:
        raise ValueError("checkpoint has %d of %d shards" % (len(parts), world))
    parts.sort(key=lambda p: int(p["__shard__"][0]))
    n_rows = int(parts[0]["__shard__"][2])
    out = {k: v for k, v in parts[0].items() if k != "__shard__" and k not in TABLE_KEYS}
    for key in TABLE_KEYS:
        if key in parts[0]:
            local = parts[0][key].shape[0]
            full = np.zeros((local * world,) + parts[0][key].shape[1:], parts[0][key].dtype)
            for r, p in enumerate(parts):
                full[r::world] = p[key]              # owner(row) = row % world, local row = row // world
            out[key] = full[:n_rows]
    return out


def load_model(model_path, net, prefix="rec", load_optimizer=True):
    """tools/utils/save_load.py:42-46 (+ optimizer state when present and the layout matches)."""
    model_prefix = os.path.join(model_path, prefix)
    net.set_dict(_load_global(model_prefix))
    opt_file = model_prefix + ".pdopt"
    if load_optimizer and os.path.exists(opt_file) and getattr(net, "comm", None) is None:
        with open(opt_file, "rb") as f:
            set_optimizer_state(net, pickle.load(f))
    return net
