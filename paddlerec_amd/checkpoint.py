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
PS / gpubox tables (ops.PsTable: ShardedDeepFMLayer(table='ps'), BenchmarkDNNLayer(sparse_optimizer='ps')) are saved
as the EXISTING feature values only — `ps.rows` (GLOBAL row ids) + `ps.records` (the whole accessor record: weights,
show / click, g2sums, state, delta_score, unseen_days) — never as a dense [N,D] parameter: a 1.25e9-row shard with a few
million live keys is a few hundred MB on disk, and a loaded value keeps its state (a dense export would come back as
"no such key" and be overwritten by the first push).  There is no Adam row state for such a table.
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
        st["dense.m"], st["dense.v"] = _np(dense.packed(dense.m)), _np(dense.packed(dense.v))
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
        dense.load_packed(dense.m, st["dense.m"])
        dense.load_packed(dense.v, st["dense.v"])
    if any(k.startswith("sparse.") for k in st):
        net._ensure_sparse_state()
        for k in ("m", "v", "m1", "v1"):
            if "sparse." + k in st and k in net.sparse_state:
                dst = net.sparse_state[k]
                dst.copy_(torch.as_tensor(st["sparse." + k]).to(dst.device).reshape(dst.shape))


def _ps_table(net):
    """The ops.PsTable of a net, or None."""
    t = getattr(net, "ps", None)
    return t if t is not None else getattr(net, "table", None)


def _ps_export(net):
    """{ps.rows: GLOBAL ids of the existing values, ps.records: their accessor records, ps.meta} of a PS table."""
    t = _ps_table(net)
    comm = getattr(net, "comm", None)
    world, rank = (comm.world, comm.rank) if comm is not None else (1, 0)
    live = torch.nonzero(t.rec[:, t.state_col] != 0).reshape(-1)
    glob = getattr(net, "global_rows", t.num_rows)
    return {"ps.rows": (live * world + rank).cpu().numpy().astype(np.int64), "ps.records": _np(t.rec[live]),
            "ps.meta": np.asarray([glob, t.emb_dim, t.rec.shape[1], 1 if t.kind == "deepfm" else 0], np.int64)}


def _ps_import(net, sd, zero_first=True):
    """Writes the values of a ps.rows / ps.records pair this rank owns into its table (re-shards by global id)."""
    t = _ps_table(net)
    comm = getattr(net, "comm", None)
    world, rank = (comm.world, comm.rank) if comm is not None else (1, 0)
    meta = sd["ps.meta"]
    glob = getattr(net, "global_rows", t.num_rows)
    if int(meta[0]) != glob or int(meta[1]) != t.emb_dim or int(meta[2]) != t.rec.shape[1] or \
            int(meta[3]) != (1 if t.kind == "deepfm" else 0):
        raise ValueError("checkpoint holds a PS table %s, the model's is %s"
                         % (meta.tolist(), [glob, t.emb_dim, t.rec.shape[1], 1 if t.kind == "deepfm" else 0]))
    if zero_first:
        t.rec.zero_()
    rows = np.asarray(sd["ps.rows"], np.int64)
    mine = np.nonzero(rows % world == rank)[0]
    if len(mine):
        di = torch.as_tensor(rows[mine] // world, device=t.rec.device)
        t.rec[di] = torch.as_tensor(np.asarray(sd["ps.records"])[mine]).to(t.rec.device)


def save_model(net, optimizer, model_path, epoch_id, prefix="rec"):
    """tools/utils/save_load.py:25-31.  `optimizer` may be None (state is read from the net) or a dict."""
    model_path = os.path.join(model_path, str(epoch_id))
    _mkdir_if_not_exist(model_path)
    model_prefix = os.path.join(model_path, prefix)
    comm = getattr(net, "comm", None)
    sd = {k: _np(v) for k, v in net.state_dict().items()}
    if _ps_table(net) is not None:              # the accessor table: existing values only, whole records
        for k in TABLE_KEYS + ("embedding",):
            sd.pop(k, None)
        sd.update(_ps_export(net))
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


def _shard_files(model_prefix, ext=".pdparams"):
    d = os.path.dirname(model_prefix)
    base = os.path.basename(model_prefix)
    if not os.path.isdir(d):
        return []
    return sorted(os.path.join(d, f) for f in os.listdir(d) if f.startswith(base + ".shard") and f.endswith(ext))


def _pick_source(model_prefix):
    """-> ("single", path) or ("shards", [paths]).  A directory that holds BOTH a single-process file and rank shards
    (a sharded run saved over an older single-process one, or the reverse) is ambiguous: the newer set wins and the
    stale one is named in a warning instead of silently shadowing it."""
    import logging
    single = model_prefix + ".pdparams"
    shards = _shard_files(model_prefix)
    if os.path.exists(single) and shards:
        newest_shard = max(os.path.getmtime(f) for f in shards)
        use_single = os.path.getmtime(single) >= newest_shard
        logging.getLogger(__name__).warning(
            "%s holds both %s and %d rank shards: loading the newer %s, ignoring the stale other",
            os.path.dirname(model_prefix), os.path.basename(single), len(shards), "single file" if use_single else "shards")
        return ("single", single) if use_single else ("shards", shards)
    if os.path.exists(single):
        return "single", single
    if shards:
        return "shards", shards
    raise FileNotFoundError(single)


def _read_shards(paths):
    parts = []
    for fn in paths:
        with open(fn, "rb") as f:
            parts.append(pickle.load(f))
    world = int(parts[0]["__shard__"][1])
    if len(parts) != world:
        raise ValueError("checkpoint has %d of %d shards" % (len(parts), world))
    parts.sort(key=lambda p: int(p["__shard__"][0]))
    return parts, world, int(parts[0]["__shard__"][2])


def _load_global(model_prefix):
    """Reads either the single file or all rank shards and returns the GLOBAL state dict (single-process nets)."""
    kind, src = _pick_source(model_prefix)
    if kind == "single":
        with open(src, "rb") as f:
            return pickle.load(f)
    parts, world, n_rows = _read_shards(src)
    out = {k: v for k, v in parts[0].items() if k != "__shard__" and k not in TABLE_KEYS and not k.startswith("ps.")}
    if "ps.rows" in parts[0]:                   # a PS table: concatenate the shards' existing values
        out["ps.rows"] = np.concatenate([p["ps.rows"] for p in parts])
        out["ps.records"] = np.concatenate([p["ps.records"] for p in parts])
        out["ps.meta"] = parts[0]["ps.meta"]
    for key in TABLE_KEYS:
        if key in parts[0]:
            local = parts[0][key].shape[0]
            full = np.zeros((local * world,) + parts[0][key].shape[1:], parts[0][key].dtype)
            for r, p in enumerate(parts):
                full[r::world] = p[key]              # owner(row) = row % world, local row = row // world
            out[key] = full[:n_rows]
    return out


def _reshard_rows(old_rank, old_world, n_local_old, new_rank, new_world, n_rows):
    """Rows of old shard `old_rank` that the new rank owns: (indices into the old shard, local rows in the new one)."""
    g = old_rank + np.arange(n_local_old, dtype=np.int64) * old_world        # global rows the old shard holds
    keep = (g < n_rows) & (g % new_world == new_rank)
    return np.nonzero(keep)[0], g[keep] // new_world


def _shard_id(path):
    """(rank, world) from the file name rec.shard{r}of{G}.pdparams — no unpickling."""
    import re
    m = re.search(r"\.shard(\d+)of(\d+)\.pd(?:params|opt)$", path)
    if not m:
        raise ValueError("not a shard file: %s" % path)
    return int(m.group(1)), int(m.group(2))


def _load_sharded(net, model_prefix, shard_paths, load_optimizer):
    """A row-sharded net resumes from rank shards WITHOUT assembling the global table and without holding more than
    ONE shard in host memory: the plan comes from the file names; with the same world size a rank opens only its own
    file; otherwise it walks the old shards one at a time, keeps the rows it owns now and drops the shard before the
    next.  The per-rank optimizer shards (step, dense and sparse Adam moments) are restored the same way."""
    from .deepfm import DeepFMLayer
    comm = net.comm
    files = sorted(shard_paths, key=lambda f: _shard_id(f)[0])
    old_world = _shard_id(files[0])[1]
    if len(files) != old_world or [_shard_id(f)[0] for f in files] != list(range(old_world)):
        raise ValueError("checkpoint has %d of %d shards" % (len(files), old_world))
    same = old_world == comm.world
    todo = [files[comm.rank]] if same else files
    ps = _ps_table(net) is not None
    net._next_lookup = None
    local = net.state_dict()
    opt_ok = load_optimizer and all(os.path.exists(f[: -len(".pdparams")] + ".pdopt") for f in files)
    first = True
    for fn in todo:
        with open(fn, "rb") as f:
            sd = pickle.load(f)
        r_old, n_rows = int(sd["__shard__"][0]), int(sd["__shard__"][2])
        if n_rows != net.global_rows:
            raise ValueError("checkpoint table has %d rows, the model %d" % (n_rows, net.global_rows))
        opt = None
        if opt_ok:
            with open(fn[: -len(".pdparams")] + ".pdopt", "rb") as f:
                opt = pickle.load(f)
        if first:       # the replicated dense parameters / dense moments / step: any one shard holds them
            DeepFMLayer.set_dict(net, {k: v for k, v in sd.items()
                                       if k != "__shard__" and k not in TABLE_KEYS and not k.startswith("ps.")})
            if opt is not None:
                set_optimizer_state(net, {k: v for k, v in opt.items() if not k.startswith("sparse.")})
                if not ps:
                    net._ensure_sparse_state()          # (a PS table has no Adam row state: nothing to allocate)
        if ps:
            if "ps.rows" not in sd:
                raise ValueError("%s holds no PS table (written by a table='adam' run?)" % fn)
            _ps_import(net, sd, zero_first=first)
        else:
            if "ps.rows" in sd or not any(key in sd for key in TABLE_KEYS):
                raise ValueError("%s holds no embedding table of a table='adam' run (written by a PS-table run?)" % fn)
            for key in TABLE_KEYS:
                if key not in sd:
                    continue
                src_idx, dst_idx = _reshard_rows(r_old, old_world, sd[key].shape[0], comm.rank, comm.world, n_rows)
                if len(src_idx) == 0:
                    continue
                dst = local[key]
                di = torch.as_tensor(dst_idx, device=dst.device)
                dst[di] = torch.as_tensor(sd[key][src_idx]).to(dst.device).reshape(len(src_idx), -1)
            if opt is not None:
                for k in ("m", "v", "m1", "v1"):
                    arr = opt.get("sparse." + k)
                    if arr is None or k not in net.sparse_state:
                        continue
                    src_idx, dst_idx = _reshard_rows(r_old, old_world, arr.shape[0], comm.rank, comm.world, n_rows)
                    if len(src_idx):
                        dst = net.sparse_state[k]
                        di = torch.as_tensor(dst_idx, device=dst.device)
                        dst[di] = torch.as_tensor(arr[src_idx]).to(dst.device).reshape(len(src_idx), -1)
        first = False
        del sd, opt


def load_model(model_path, net, prefix="rec", load_optimizer=True):
    """tools/utils/save_load.py:42-46 (+ optimizer state when present and the layout matches)."""
    model_prefix = os.path.join(model_path, prefix)
    comm = getattr(net, "comm", None)
    if comm is not None and comm.world > 1:
        kind, src = _pick_source(model_prefix)
        if kind == "shards":
            _load_sharded(net, model_prefix, src, load_optimizer)
            return net
    sd = _load_global(model_prefix)
    if "ps.rows" in sd:
        if _ps_table(net) is None:
            raise ValueError("the checkpoint holds a PS accessor table, the model does not")
        _ps_import(net, sd)
        sd = {k: v for k, v in sd.items() if not k.startswith("ps.")}
    net.set_dict(sd)
    opt_file = model_prefix + ".pdopt"
    if load_optimizer and os.path.exists(opt_file) and comm is None:
        with open(opt_file, "rb") as f:
            set_optimizer_state(net, pickle.load(f))
    return net
