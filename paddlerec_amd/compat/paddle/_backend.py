"""Device and operator-backend state of the compat namespace."""
import importlib
import os

import torch

_state = {"device": None, "kernels": None, "ws": None, "static": False}


def static_mode():
    return _state["static"]


def kernels():
    """The operator backend: paddlerec_amd.ops (HIP kernels behind the C-ABI).  REC_COMPAT_KERNELS=<module> replaces it
    (tests: an oracle-backed stand-in, so that the host logic runs without a GPU); there is no built-in CPU path."""
    if _state["kernels"] is None:
        name = os.environ.get("REC_COMPAT_KERNELS")
        _state["kernels"] = importlib.import_module(name) if name else importlib.import_module("paddlerec_amd.ops")
    return _state["kernels"]


def device():
    if _state["device"] is None:
        _state["device"] = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    return _state["device"]


def set_device(name):
    name = str(name)
    if name.startswith("gpu"):
        idx = name.split(":")[1] if ":" in name else "0"
        if ":" not in name and int(os.environ.get("WORLD_SIZE", "1")) > 1:
            # a rank of a multi-process launch asking for "the GPU": its own (all ranks on GPU 0 when there are fewer
            # GPUs than ranks: the shared-GPU tests)
            n = torch.cuda.device_count()
            idx = os.environ.get("LOCAL_RANK", "0") if n >= int(os.environ["WORLD_SIZE"]) else "0"
        _state["device"] = torch.device("cuda", int(idx))
        torch.cuda.set_device(_state["device"])
    elif name.startswith("cpu"):
        _state["device"] = torch.device("cpu")
    else:
        raise ValueError("compat paddle.set_device: unsupported device %r (gpu / cpu)" % name)
    return _state["device"]


def workspace():
    if _state["ws"] is None or _state["ws"].device != device():
        _state["ws"] = kernels().Workspace(device())
    return _state["ws"]
