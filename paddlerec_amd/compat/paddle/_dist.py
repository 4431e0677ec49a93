"""Multi-GPU state of the compat namespace: the reference's gpubox launch names its GPUs in FLAGS_selected_gpus
(/root/reference/tools/run_gpubox.sh:21) and ONE trainer process drives them all [EXT HeterPS]; the engine's way is one
process per GPU (paddlerec_amd.run_reference re-executes itself under torch.distributed.run, one rank per listed GPU).
Here: rank / world of that launch, the process group, and the exchange object (paddlerec_amd.sharded.Comm: RCCL through
the C-ABI exchange on GPUs, gloo with host staging in the CPU / shared-GPU tests)."""
import os

import torch

_state = {"comm": None, "pending_save": None}


def world():
    return int(os.environ.get("WORLD_SIZE", "1"))


def rank():
    return int(os.environ.get("RANK", "0"))


def comm():
    return _state["comm"]


def init():
    """fleet.init() of a rank of a multi-rank launch: process group + device + Comm.  Idempotent."""
    if world() <= 1 or _state["comm"] is not None:
        return _state["comm"]
    import torch.distributed as dist

    from . import _backend
    if _backend._state["device"] is None:          # the script has not chosen a device yet (gpubox: fleet.init() comes first)
        if torch.cuda.is_available() and not os.environ.get("REC_COMPAT_KERNELS"):
            _backend.set_device("gpu")             # -> this rank's GPU (all ranks on GPU 0 when there are fewer GPUs than ranks)
        else:
            _backend.set_device("cpu")
    on_gpu = _backend.device().type == "cuda"
    ngpu = torch.cuda.device_count() if on_gpu else 0
    # one GPU per rank -> RCCL; fewer GPUs than ranks (tests: every rank on cuda:0) or no GPU -> gloo, host-staged
    backend = os.environ.get("REC_COMPAT_BACKEND") or ("nccl" if ngpu >= world() else "gloo")
    if not dist.is_initialized():
        dist.init_process_group(backend, rank=rank(), world_size=world())
    from paddlerec_amd.sharded import Comm
    _state["comm"] = Comm()
    return _state["comm"]


def all_reduce_numpy(a, op="sum"):
    """fleet.util.all_reduce over the ranks (numpy in, numpy out)."""
    import numpy as np
    c = comm()
    if c is None:
        return a
    import torch.distributed as dist
    arr = np.asarray(a)
    dev = "cpu" if c.staged else torch.device("cuda", torch.cuda.current_device())
    t = torch.as_tensor(np.ascontiguousarray(arr)).to(dev)
    ops = {"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}
    dist.all_reduce(t, op=ops[op], group=c.group)
    return t.cpu().numpy().reshape(arr.shape)


def barrier():
    c = comm()
    if c is not None:
        import torch.distributed as dist
        dist.barrier(group=c.group)
