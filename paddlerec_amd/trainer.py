"""Train / infer loops over the engine — the caller side of the hot path (BASELINE configs[0]).

Mirrors the dygraph drivers of the reference with the same YAML keys and the same epoch / batch structure:
    tools/trainer.py:40-223   (config -> create_model -> [load_model] -> epochs x batches of
                               train_forward + backward + step -> ips log line -> save_model per epoch)
    tools/infer.py:55-195     (per epoch: load_model(infer_load_path/epoch) -> eval batches -> AUC)
    tools/utils/utils_single.py:42-62,100-141  (yaml flattening "runner.xxx"; the data loader over the files
                               of runner.train_data_dir with drop_last=True)
What differs by construction: a batch never becomes 28 per-sample NumPy arrays (the host parser hands over
[B,26] / [B,13] device tensors), `loss.backward(); optimizer.step()` is the explicit `train_step` chain of the
host mirrors, and the AUC buckets stay on the device (read back only when a log line prints them).

    python -m paddlerec_amd.trainer -m <config.yaml> [--model deepfm|fm|wide_deep|dnn|dcn_v2|din|xdeepfm|dlrm] [-o runner.epochs=1 ...] [--infer]
    python -m torch.distributed.run --nproc-per-node G -m paddlerec_amd.trainer -m <config.yaml>     # collective mode
"""
import argparse
import logging
import os
import time

import torch

from . import checkpoint

logger = logging.getLogger("paddlerec_amd.trainer")

MODELS = ("deepfm", "fm", "wide_deep", "dnn", "dcn_v2", "din", "xdeepfm", "dlrm")


# ------------------------------------------------------------------------------------ configuration
def _flatten(node, prefix, out):
    for k, v in node.items():
        key = prefix + "." + str(k) if prefix else str(k)
        if isinstance(v, dict):
            _flatten(v, key, out)
        else:
            out[key] = v


def load_yaml(path, overrides=()):
    """YAML file -> flat {"runner.train_batch_size": 2, "hyper_parameters.optimizer.learning_rate": ...}
    (utils_single.py:42-62 keeps the parts workspace / runner / hyper_parameters).  `overrides`: "key=value"
    strings, coerced to the type of the value they replace (trainer.py:52-65)."""
    import yaml
    with open(path, "r") as f:
        doc = yaml.safe_load(f) or {}
    cfg = {}
    for part in ("workspace", "runner", "hyper_parameters"):
        if isinstance(doc.get(part), dict):
            _flatten(doc[part], part, cfg)
        elif part in doc:
            cfg[part] = doc[part]
    cfg["config_abs_dir"] = os.path.dirname(os.path.abspath(path))
    for item in overrides:
        key, _, value = item.partition("=")
        old = cfg.get(key)
        if isinstance(old, bool):
            value = value.lower() == "true"
        elif isinstance(old, int):
            value = int(value)
        elif isinstance(old, float):
            value = float(value)
        cfg[key] = value
    return cfg


def guess_model(config_path):
    """The reference picks `dygraph_model.py` next to the YAML; here the directory name selects the host mirror."""
    name = os.path.basename(os.path.dirname(os.path.abspath(config_path)))
    return name if name in MODELS else None


def _dygraph_model(name):
    if name == "deepfm":
        from .deepfm import DygraphModel
    elif name == "fm":
        from .fm import DygraphModel
    elif name == "wide_deep":
        from .wide_deep import DygraphModel
    elif name == "dnn":
        from .dnn import DygraphModel
    elif name == "dcn_v2":
        from .dcn_v2 import DygraphModel
    elif name == "din":
        from .din import DygraphModel
    elif name == "xdeepfm":
        from .xdeepfm import DygraphModel
    elif name == "dlrm":
        from .dlrm import DygraphModel
    else:
        raise ValueError("unknown model %r (known: %s)" % (name, ", ".join(MODELS)))
    return DygraphModel()


def _data_files(config, key):
    d = config.get(key)
    if d is None:
        raise ValueError("%s is not set" % key)
    if not os.path.isabs(d):
        d = os.path.join(config.get("config_abs_dir", "."), d)
    if not os.path.isdir(d):                                   # utils_single.py:100-101
        raise ValueError("%s = %r is not a directory" % (key, d))
    files = sorted(os.path.join(d, x) for x in os.listdir(d) if not x.startswith("."))
    if not files:
        raise ValueError("no data files under %r" % d)
    return files


def create_data_loader(config, model, device, mode="train", shard=None):
    """-> a CALLABLE returning a fresh batch iterator, as the reference uses its DataLoader (`train_dataloader()`)."""
    from . import reader
    files = _data_files(config, "runner.train_data_dir" if mode == "train" else "runner.test_data_dir")
    bs = config.get("runner.train_batch_size" if mode == "train" else "runner.infer_batch_size")
    if model == "din":
        return lambda: iter(reader.DinReader(files, bs, device))
    return lambda: iter(reader.SlotTextReader(files, bs, device, log1p_dense=(model == "dcn_v2"), shard=shard))


def _batch_size(batch):
    return int(batch[0].shape[0])


def _metric_values(dy_model_class, metric_list, metric_names):
    from .deepfm import auc_from_buckets
    fn = getattr(dy_model_class, "metric_value", None)        # models with metrics other than AUC bucket pairs (dlrm)
    if fn is not None:
        return {n: fn(n, m) for n, m in zip(metric_names, metric_list)}
    return {n: auc_from_buckets(m[0], m[1]) for n, m in zip(metric_names, metric_list)}


def _global_metric_values(dy_model_class, metric_list, metric_names, comm):
    """Global AUC of a data-parallel run: the integer buckets of all ranks are summed (one all-reduce of a copy, only
    where a value is printed) — what the reference's fleet metrics do for its static AUC (utils_single.py:143-158)."""
    if comm is None or comm.world == 1:
        return _metric_values(dy_model_class, metric_list, metric_names)
    summed = [tuple(comm.all_reduce_sum(t.clone()) for t in m) for m in metric_list]
    return _metric_values(dy_model_class, summed, metric_names)


def _local_batches(config, mode, shard, batch_size):
    """Batches this rank's files hold (drop_last over the concatenation of its files, as the readers batch)."""
    from . import reader
    from .reader import _FileBatches
    files = _FileBatches(_data_files(config, "runner.train_data_dir" if mode == "train" else "runner.test_data_dir"),
                         batch_size, "cpu", None, shard).file_list
    lines = 0
    for path in files:
        with open(path, "rb") as f:
            lines += reader._count_lines(f.read(), 0) if os.path.getsize(path) else 0
    return lines // batch_size


def _agreed_batches(config, mode, comm, batch_size):
    """Every rank must issue the same number of steps (each step holds collectives): the minimum over the ranks."""
    n = _local_batches(config, mode, (comm.rank, comm.world), batch_size)
    t = torch.tensor([-n], dtype=torch.int64, device="cpu" if comm.staged else torch.device("cuda", torch.cuda.current_device()))
    comm.dist.all_reduce(t, op=comm.dist.ReduceOp.MAX, group=comm.group)         # max of -n = -min n
    return int(-t.item())


def _lookahead(it, limit=None):
    """(batch, next batch or None) pairs — the row-sharded step routes the next batch a step ahead."""
    prev, n = None, 0
    if limit is not None and limit <= 0:
        return
    for cur in it:
        if prev is not None:
            last = limit is not None and n + 1 >= limit      # the agreed last step: no rank may prefetch beyond it
            yield prev, (None if last else cur)              # (a prefetch holds collectives — all ranks or none)
            n += 1
            if last:
                return
        prev = cur
    if prev is not None and (limit is None or n < limit):
        yield prev, None


def _check_status(dy_model, comm, what):
    """Paddle's lookup raises on an id outside the table [EXT]; the engine's kernels flag it in a device status word
    instead (no sync on the hot path).  Read it wherever the device is read back anyway — every rank raises together
    in collective mode (the flag is summed over the ranks first)."""
    st = getattr(dy_model, "status", None)
    if st is None:
        return
    flag = st.clone()
    if comm is not None and comm.world > 1:
        comm.all_reduce_sum(flag)
    if int(flag.item()) != 0:
        from ._lib import RecError
        raise RecError("%s: an id was outside [0, sparse_feature_number) — check the dataset against the config "
                       "(hyper_parameters.sparse_feature_number)" % what)


def _apply_optimizer_config(config, model, dy_model):
    """hyper_parameters.optimizer.lazy_mode (default False = the reference's dygraph Adam, dygraph_model.py:61-65:
    every row's moments decay each step) for the nets that implement both variants; the deviations of the others
    from the reference's dygraph semantics are logged once, loudly."""
    lazy = bool(config.get("hyper_parameters.optimizer.lazy_mode", False))
    if model == "xdeepfm":       # its create_optimizer asks for Adam(lazy_mode=True) itself (xdeepfm/dygraph_model.py:66-67)
        logger.info("sparse optimizer: Adam lazy_mode=True (the reference's own setting for xdeepfm)")
    elif hasattr(type(dy_model), "lazy_mode"):
        dy_model.lazy_mode = lazy
        logger.info("sparse optimizer: Adam lazy_mode=%s%s", lazy,
                    "" if lazy else " (dygraph default: the whole table is streamed every step)")
    elif model in ("wide_deep", "dnn", "dcn_v2"):
        logger.warning("DEVIATION from the reference's dygraph run: %s updates only the rows a batch touches "
                       "(Adam lazy_mode=True); the reference's dygraph Adam also decays the moments of every "
                       "other row each step", model)
    if model == "dcn_v2":
        logger.info("dcn_v2 train mode: Dropout(%.2f) after every element of the DNN tower (dcn_v2/net.py:181-183) with the "
                    "engine's counter-based masks (seed %d; Paddle's own mask stream is not reproducible), L2Decay(%g) on "
                    "its weights after the global-norm clip", getattr(dy_model, "dropout_rate", 0.0),
                    getattr(dy_model, "dropout_seed", 0), getattr(dy_model, "l2_dnn", 0.0))


def _reset(metric_list):
    for m in metric_list:
        m[0].zero_()
        m[1].zero_()


# -------------------------------------------------------------------------------------------- train
def train(config, model, device="cuda", kernels=None, comm=None):
    """tools/trainer.py main(): returns one summary dict per epoch
    {"epoch", "batches", "samples", "loss", "ips", <metric name>: value, "model_dir"}.
    comm (paddlerec_amd.sharded.Comm, world > 1): the `use_fleet` collective mode of trainer.py:113-119 — one process
    per GPU, the files of train_data_dir split over the ranks (criteo_reader.py:30-43), the embedding tables
    row-sharded (deepfm only), loss / AUC / samples reported for the GLOBAL batch, one checkpoint shard per rank."""
    torch.manual_seed(config.get("runner.seed", 12345))
    dy_model_class = _dygraph_model(model)
    kw = {"kernels": kernels} if kernels is not None else {}
    world = comm.world if comm is not None else 1
    if world > 1:
        if model != "deepfm":
            raise ValueError("the row-sharded collective mode is built for deepfm only")
        kw["comm"] = comm
    dy_model = dy_model_class.create_model(config, device, **kw)
    if config.get("runner.model_init_path"):
        checkpoint.load_model(config["runner.model_init_path"], dy_model)
    _apply_optimizer_config(config, model, dy_model)
    epochs = config.get("runner.epochs", 1)
    print_interval = max(int(config.get("runner.print_interval", 1) or 1), 1)
    save_path = config.get("runner.model_save_path", "model_output")
    use_auc = config.get("runner.use_auc", False)
    shard = (comm.rank, comm.world) if world > 1 else None
    loader = create_data_loader(config, model, dy_model.device, "train", shard)
    limit = _agreed_batches(config, "train", comm, config.get("runner.train_batch_size")) if world > 1 else None
    if limit is not None and limit <= 0:       # some rank holds less than one batch: every rank raises (no hang)
        raise ValueError("train_dataloader is null on at least one rank, please ensure batch size < dataset size "
                         "of every rank's file split!")
    summaries = []
    pending_iter = None
    for epoch_id in range(config.get("last_epoch", -1) + 1, epochs):
        metric_list, metric_names = dy_model_class.create_metrics(dy_model.device)
        epoch_begin = time.time()
        reader_cost = run_cost = 0.0
        reader_total = run_total = 0.0            # whole-epoch stage times (the interval ones are reset at every log line)
        interval_samples = total_samples = n_batches = 0
        loss = None
        reader_start = time.time()
        epoch_iter, pending_iter = (pending_iter if pending_iter is not None else loader()), None
        for batch_id, (batch, nxt) in enumerate(_lookahead(epoch_iter, limit)):
            reader_cost += time.time() - reader_start
            reader_total += time.time() - reader_start
            t0 = time.time()
            fkw = {"next_batch": nxt} if world > 1 else {}
            loss, metric_list, _ = dy_model_class.train_forward(dy_model, metric_list, batch, config, **fkw)
            run_cost += time.time() - t0
            run_total += time.time() - t0
            bs = _batch_size(batch) * world
            interval_samples += bs
            total_samples += bs
            n_batches += 1
            if batch_id % print_interval == 0:        # the only place the device is read back
                _check_status(dy_model, comm, "train")
                vals = _global_metric_values(dy_model_class, metric_list, metric_names, comm)
                logger.info("epoch: %d, batch_id: %d, %sloss: %.6f, avg_reader_cost: %.5f sec, avg_batch_cost: "
                            "%.5f sec, avg_samples: %.5f, ips: %.5f ins/s", epoch_id, batch_id,
                            "".join("%s:%.6f, " % kv for kv in vals.items()), float(loss.reshape(-1)[0].item()),
                            reader_cost / print_interval, (reader_cost + run_cost) / print_interval,
                            interval_samples / print_interval,
                            interval_samples / (reader_cost + run_cost + 0.0001))
                reader_cost = run_cost = 0.0
                interval_samples = 0
            reader_start = time.time()
        if n_batches == 0:                                # trainer.py:143-144
            raise ValueError("train_dataloader is null, please ensure batch size < dataset size!")
        if epoch_id + 1 < epochs:     # the next epoch's first files are read and parsed while the device drains and the
            pending_iter = loader()   # checkpoint is written (the readers start their background parse at iter())
        t_sync = time.time()
        if dy_model.device.type == "cuda":
            torch.cuda.synchronize(dy_model.device)
        elapsed = time.time() - epoch_begin
        sync_s = time.time() - t_sync
        _check_status(dy_model, comm, "train")
        vals = _global_metric_values(dy_model_class, metric_list, metric_names, comm)
        if use_auc:
            _reset(metric_list)
        t_ck = time.time()
        model_dir = checkpoint.save_model(dy_model, None, save_path, epoch_id, prefix="rec") \
            if config.get("runner.save_checkpoint", True) else None
        # ips = samples / (first batch requested .. device idle): the checkpoint is NOT inside it; the stage times say
        # where the host spent the epoch (waiting for the reader / issuing steps / draining the device at the end)
        s = dict(epoch=epoch_id, batches=n_batches, samples=total_samples, loss=float(loss.reshape(-1)[0].item()),
                 ips=total_samples / max(elapsed, 1e-9), model_dir=model_dir, epoch_s=elapsed, reader_wait_s=reader_total,
                 step_issue_s=run_total, final_sync_s=sync_s, checkpoint_s=time.time() - t_ck, **vals)
        from .reader import _FileBatches
        if _FileBatches.last_trace:                 # REC_READER_TRACE=1: per file (name, load s, s blocked on the queue)
            s["reader_trace"] = [[a, round(b, 4), round(c, 4)] for a, b, c in _FileBatches.last_trace]
        logger.info("epoch: %d done, %s epoch time: %.2f s", epoch_id,
                    "".join("%s: %.6f," % kv for kv in vals.items()), elapsed)
        summaries.append(s)
    return summaries, dy_model


# -------------------------------------------------------------------------------------------- infer
def infer(config, model, device="cuda", kernels=None, comm=None):
    """tools/infer.py main(): for every saved epoch in [infer_start_epoch, infer_end_epoch) load the checkpoint
    and run the test set; returns one dict per epoch {"epoch", "batches", "samples", <metric>: value}.
    comm: as in train() — test files split over the ranks, sharded lookup, global AUC."""
    dy_model_class = _dygraph_model(model)
    kw = {"kernels": kernels} if kernels is not None else {}
    world = comm.world if comm is not None else 1
    if world > 1:
        kw["comm"] = comm
    dy_model = dy_model_class.create_model(config, device, **kw)
    load_path = config.get("runner.infer_load_path", "model_output")
    shard = (comm.rank, comm.world) if world > 1 else None
    loader = create_data_loader(config, model, dy_model.device, "test", shard)
    limit = _agreed_batches(config, "test", comm, config.get("runner.infer_batch_size")) if world > 1 else None
    if limit is not None and limit <= 0:
        raise ValueError("test_dataloader is null on at least one rank, please ensure batch size < dataset size!")
    use_auc = config.get("runner.use_auc", False)
    metric_list, metric_names = dy_model_class.create_metrics(dy_model.device)
    out = []
    for epoch_id in range(config.get("runner.infer_start_epoch", 0), config.get("runner.infer_end_epoch", 1)):
        checkpoint.load_model(os.path.join(load_path, str(epoch_id)), dy_model, load_optimizer=False)
        n_batches = samples = 0
        for batch, _ in _lookahead(loader(), limit):
            metric_list, _ = dy_model_class.infer_forward(dy_model, metric_list, batch, config)
            n_batches += 1
            samples += _batch_size(batch) * world
        if n_batches == 0:
            raise ValueError("test_dataloader is null, please ensure batch size < dataset size!")
        _check_status(dy_model, comm, "infer")
        vals = _global_metric_values(dy_model_class, metric_list, metric_names, comm)
        if use_auc:
            _reset(metric_list)
        logger.info("epoch: %d done, %s", epoch_id, "".join("%s: %.6f," % kv for kv in vals.items()))
        out.append(dict(epoch=epoch_id, batches=n_batches, samples=samples, **vals))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description="dygraph train / infer over the recengine host mirrors")
    ap.add_argument("-m", "--config_yaml", required=True)
    ap.add_argument("--model", choices=MODELS, default=None)
    ap.add_argument("-o", "--opt", nargs="*", default=[], help="key=value overrides, e.g. runner.epochs=1")
    ap.add_argument("--infer", action="store_true", help="run tools/infer.py's loop instead of the training loop")
    ap.add_argument("--device", default="cuda")
    args = ap.parse_args(argv)
    logging.basicConfig(format="%(asctime)s - %(levelname)s - %(message)s", level=logging.INFO)
    config = load_yaml(args.config_yaml, args.opt)
    model = args.model or guess_model(args.config_yaml)
    if model is None:
        raise SystemExit("cannot tell the model from the path of %s: pass --model" % args.config_yaml)
    comm, device = None, args.device
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:          # launched by torch.distributed.run: one process per GPU (runner.use_fleet of the reference)
        import torch.distributed as dist
        from .sharded import Comm
        if device.startswith("cuda"):
            local = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(local)
            device = "cuda:%d" % local
            dist.init_process_group("nccl", device_id=torch.device(device))
        else:
            dist.init_process_group("gloo")
        comm = Comm()
    run = infer if args.infer else (lambda *a: train(*a)[0])
    for s in run(config, model, device, None, comm):
        if comm is None or comm.rank == 0:
            print(s)
    if comm is not None:
        comm.dist.destroy_process_group()


if __name__ == "__main__":
    main()
