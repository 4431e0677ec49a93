class ParamAttr:
    def __init__(self, name=None, initializer=None, regularizer=None, learning_rate=1.0, trainable=True):
        self.name, self.initializer, self.regularizer = name, initializer, regularizer
        self.learning_rate, self.trainable = learning_rate, trainable


class _PSGPU:
    """paddle.framework.core.PSGPU (tools/static_gpubox_trainer.py:152-160,207,219): the pass surface of the GPU
    parameter server over the sparse tables of the main program.  begin_pass (build the pass's GPU table from the
    servers [EXT]): the dataset is already in memory and the device table is born lazily — nothing to stage; end_pass
    (dump the GPU table back to the servers [EXT]): the table already is the persistent one — nothing to move.  The
    accessor's Shrink is a separate call in Paddle (fleet.shrink); shrink() below exposes it."""

    def __init__(self):
        self.slots, self.slot_dims, self.gpus, self.passes, self.deleted = None, None, None, 0, []

    def set_slot_vector(self, slots):
        self.slots = [int(s) for s in slots]

    def set_slot_dim_vector(self, dims):
        self.slot_dims = [int(d) for d in dims]

    def init_gpu_ps(self, gpus):
        if self.slots is None or self.slot_dims is None or len(self.slots) != len(self.slot_dims):
            raise ValueError("set_slot_vector / set_slot_dim_vector must be called first, with equal lengths")
        self.gpus = [int(g) for g in gpus]
        from . import _dist
        if len(self.gpus) > 1 and _dist.world() != len(self.gpus):
            raise RuntimeError(
                "init_gpu_ps(%s): %d GPUs, but this process is rank %d of %d.  The engine runs one process per GPU: start "
                "the script with `python -m paddlerec_amd.run_reference tools/static_gpubox_trainer.py ...` — with "
                "FLAGS_selected_gpus naming the GPUs it re-executes itself as one rank per GPU (torch.distributed.run) and "
                "the tables of static.nn.sparse_embedding are row-sharded over the ranks" %
                (self.gpus, len(self.gpus), _dist.rank(), _dist.world()))

    def begin_pass(self):
        if self.gpus is None:
            raise RuntimeError("init_gpu_ps() comes before begin_pass()")
        self.passes += 1

    def end_pass(self):
        if self.passes == 0:
            raise RuntimeError("end_pass() without begin_pass()")

    def shrink(self):
        from . import _backend, static
        K = _backend.kernels()
        ctr = getattr(static.default_main_program(), "ctr_accessor_param", {}) or {}
        n = 0
        for tab in static.default_main_program().tables.values():
            n += K.ps_shrink_rows(tab.table, ctr.get("show_click_decay_rate", 0.98), ctr.get("delete_threshold", 0.8),
                                  ctr.get("delete_after_unseen_days", float("inf")))
        self.deleted.append(n)

    def finalize(self):
        pass


class _EOFException(Exception):
    pass


class core:
    PSGPU = _PSGPU
    EOFException = _EOFException
