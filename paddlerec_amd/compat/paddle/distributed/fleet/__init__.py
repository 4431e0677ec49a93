"""paddle.distributed.fleet — the worker-side surface of the reference's entry points.

tools/trainer.py:113-119 (collective dygraph): distributed_optimizer / distributed_model are pass-through; the engine's
own collective mode is the row-sharded trainer of paddlerec_amd.trainer (one process per GPU, torch.distributed.run).
tools/static_gpubox_trainer.py:101-260 (parameter-server / gpubox mode): the "servers" are the device tables of the main
program (compat/paddle/static), so is_server() is False and init_worker / stop_worker have nothing to contact;
DistributedStrategy is the attribute bag program_helper.get_strategy fills (its sparse_table_configs carry the accessor
block of the YAML to the tables).  With FLAGS_selected_gpus naming N > 1 GPUs (tools/run_gpubox.sh:21) the launch is N
ranks (paddlerec_amd.run_reference spawns them, one per GPU; compat/paddle/_dist.py): every rank is a worker
(worker_index = rank), the tables are row-sharded over the ranks, util.all_reduce / barrier_worker are real
collectives, and barrier_worker is also where a pass checkpoint requested by the first worker is written by ALL ranks
(every rank holds a shard of the table; the reference only calls save_inference_model on the first worker)."""
import os
import sys

from . import base  # noqa: F401
from .base import role_maker  # noqa: F401


class DistributedStrategy:
    def __init__(self):
        self.a_sync = False
        self.a_sync_configs = {}
        self.trainer_desc_configs = {}
        self.fs_client_param = {}
        self.sparse_table_configs = {}
        self.adam_d2sum = True
        self.is_with_coordinator = False


_state = {"init": False, "collective": False, "strategy": None}


def init(role_maker=None, is_collective=False, strategy=None):  # noqa: A002
    """is_collective=True (tools/trainer.py:112-118, runner.use_fleet): dygraph data parallel over the ranks of the
    launch (`python -m torch.distributed.run --nproc-per-node N -m paddlerec_amd.run_reference tools/trainer.py ...`):
    replicated model, every rank reads its share of the files (the reference's readers split the file list by
    paddle.distributed.get_rank()), gradients averaged before optimizer.step() — distributed_model /
    distributed_optimizer below.  (The engine's own collective mode with ROW-SHARDED tables is paddlerec_amd.trainer.)"""
    _state.update(init=True, collective=bool(is_collective), strategy=strategy)
    from ... import _dist
    _dist.init()


def is_server():
    return os.environ.get("TRAINING_ROLE", "TRAINER") == "PSERVER"


def is_worker():
    return not is_server()


def is_first_worker():
    return worker_index() == 0


def worker_index():
    from ... import _dist
    if _dist.world() > 1:
        return _dist.rank()
    return int(os.environ.get("PADDLE_TRAINER_ID", "0"))


def worker_num():
    from ... import _dist
    if _dist.world() > 1:
        return _dist.world()
    return int(os.environ.get("PADDLE_TRAINERS_NUM", "1"))


def init_server(*a, **k):
    raise NotImplementedError("no parameter-server process exists: the table lives on the worker's GPU")


def run_server():
    raise NotImplementedError("no parameter-server process exists: the table lives on the worker's GPU")


def init_worker():
    pass


def stop_worker():
    pass


def barrier_worker():
    from ... import _dist, _ps_save_shards
    if _dist.world() > 1:
        _ps_save_shards()           # a save the first worker asked for: every rank writes its shard (collective)
        _dist.barrier()


def save_inference_model(executor, dirname, feeded_var_names, target_vars, main_program=None, export_for_deployment=True,
                         mode=0):
    from ... import _dist, _ps_save
    if _dist.world() > 1:           # only the first worker gets here (static_gpubox_trainer.py:211-216): the table is
        _dist._state["pending_save"] = (dirname, int(mode))   # sharded, so the write happens in barrier_worker()
        return
    _ps_save(dirname, mode)


class _StrategyOptimizer:
    """fleet.distributed_optimizer(optimizer, strategy): minimize() also hands the strategy's sparse-table accessor
    block (program_helper.py:81-90: table_parameters.* of the YAML) to the main program's tables."""

    def __init__(self, optimizer, strategy):
        self._opt, self._strategy = optimizer, strategy

    def minimize(self, loss, *a, **k):
        from ... import static
        prog = static.default_main_program()
        prog.strategy = self._strategy
        return self._opt.minimize(loss, *a, **k)

    def __getattr__(self, name):
        return getattr(self._opt, name)


class _CollectiveOptimizer:
    """fleet.distributed_optimizer in collective dygraph mode: step() first averages the gradients over the ranks
    (Paddle's DataParallel scales the loss by 1 / nranks and all-reduces): ONE all-reduce of the flattened dense
    gradients, an all-gather of the SelectedRows of every sparse=True embedding (ids + gradient rows, rank-major — the
    order of the concatenated batch — scaled by 1 / nranks), then the wrapped optimizer's step on every rank."""

    def __init__(self, optimizer):
        self._opt = optimizer

    def step(self):
        import torch as _t
        import torch.distributed as dist
        from ... import _dist
        c = _dist.comm()
        if c is not None and c.world > 1:
            G = c.world
            params = list(self._opt._params)
            dense = [p for p in params if not getattr(p, "_sparse_grads", None)]
            if dense:
                flat = _t.cat([(p.grad if p.grad is not None else _t.zeros_like(p)).reshape(-1) for p in dense])
                c.all_reduce_sum(flat)
                flat /= float(G)
                o = 0
                for p in dense:
                    k = p.numel()
                    p.grad = flat[o:o + k].view_as(p).clone()
                    o += k
            for p in params:
                sg = getattr(p, "_sparse_grads", None)
                if not sg:
                    continue
                ids = _t.cat([g[0] for g in sg]).contiguous()
                rows = (_t.cat([g[1] if g[3] == 1 else g[1].repeat_interleave(g[3], dim=0) for g in sg])
                        / float(G)).contiguous()
                n = _t.tensor([ids.numel()], dtype=_t.int64, device="cpu" if c.staged else ids.device)
                sizes = [_t.zeros_like(n) for _ in range(G)]
                dist.all_gather(sizes, n, group=c.group)
                sizes = [int(x.item()) for x in sizes]
                all_ids = _t.zeros(sum(sizes), dtype=ids.dtype, device=ids.device)
                all_rows = _t.zeros(sum(sizes), rows.shape[1], dtype=rows.dtype, device=rows.device)
                # every rank sends its whole list to every rank: all-to-all with equal sends = all-gather(v)
                c.all_to_all(all_ids, ids.repeat(G), sizes, [ids.numel()] * G)
                c.all_to_all(all_rows, rows.repeat(G, 1), sizes, [ids.numel()] * G)
                p._sparse_grads = [(all_ids, all_rows, sg[0][2], 1)]
        return self._opt.step()

    def __getattr__(self, name):
        return getattr(self._opt, name)


def distributed_optimizer(optimizer, strategy=None):
    if _state.get("collective"):
        return _CollectiveOptimizer(optimizer)
    if strategy is None:
        return optimizer
    return _StrategyOptimizer(optimizer, strategy)


def distributed_model(model):
    """Collective dygraph: the replicas start from rank 0's parameters (Paddle's DataParallel broadcasts them)."""
    from ... import _dist
    c = _dist.comm()
    if _state.get("collective") and c is not None and c.world > 1:
        import torch as _t
        with _t.no_grad():
            for p in model.parameters():
                c.broadcast(p.data, src=0)
    return model


class _Util:
    def all_reduce(self, x, mode="sum", comm_world="worker"):
        from ... import _dist
        return _dist.all_reduce_numpy(x, mode) if _dist.world() > 1 else x

    def get_file_shard(self, files):
        from ... import _dist
        if _dist.world() > 1:     # the ranks of one launch read ALL files and split every global batch between them
            return sorted(files)  # (InMemoryDataset._batches): equal step counts on every rank, whatever the files hold
        n, i = worker_num(), worker_index()
        return [f for k, f in enumerate(sorted(files)) if k % n == i]

    def barrier(self, *a, **k):
        from ... import _dist
        _dist.barrier()


util = _Util()


class MultiSlotDataGenerator:
    """paddle.distributed.fleet.MultiSlotDataGenerator [EXT]: the base class of the reference's pipe_command readers
    (dnn/queuedataset_reader.py:28, slot_dnn/queuedataset_reader.py): run_from_stdin() turns every input line into
    the samples generate_sample(line)() yields and prints each as `len v ... len v ...` in slot order — the text
    protocol InMemoryDataset / QueueDataset parse back (compat/paddle/distributed/__init__.py)."""

    def _gen_str(self, sample):
        parts = []
        for name, values in sample:
            if len(values) == 0:
                raise ValueError("slot %r of a sample is empty: MultiSlot needs at least one value (pad it)" % (name,))
            parts.append(str(len(values)))
            parts.extend(str(v) for v in values)
        return " ".join(parts) + "\n"

    def generate_sample(self, line):
        raise NotImplementedError

    def generate_batch(self, samples):
        def local_iter():
            for s in samples:
                yield s
        return local_iter

    def run_from_stdin(self):
        out = sys.stdout
        for line in sys.stdin:
            for sample in self.generate_sample(line)():
                if sample is None:
                    continue
                out.write(self._gen_str(sample))

    def run_from_memory(self):
        self.run_from_stdin()


MultiSlotStringDataGenerator = MultiSlotDataGenerator
