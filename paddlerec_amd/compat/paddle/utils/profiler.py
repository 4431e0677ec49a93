"""paddle.utils.profiler — the hooks tools/profiler.py:82-110 calls from the gpubox loops
(static_gpubox_trainer.py:255,270,316) when the trainer is started with --profiler_options
"batch_range=[10,20];state=GPU;...".

Paddle's own profiler does not exist here; the step window it would have traced is marked as a ROCTX range
("paddle_profiler_window") instead, so that
    rocprofv3 --marker-trace --kernel-trace -- python -m paddlerec_amd.run_reference tools/static_gpubox_trainer.py ...
attributes kernels to the same batch range.  libroctx is bound at run time (rocprofiler-sdk's
librocprofiler-sdk-roctx.so or the legacy libroctx64.so); without one the calls are no-ops — the option never crashes
the trainer."""
import ctypes as _C
import os as _os
import sys as _sys
import time as _time

_state = {"lib": False, "open": 0, "t0": None}


def _roctx():
    if _state["lib"] is False:
        lib = None
        for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
            for path in (name, _os.path.join("/opt/rocm/lib", name)):
                try:
                    lib = _C.CDLL(path)
                    lib.roctxRangePushA.argtypes = [_C.c_char_p]
                    lib.roctxRangePushA.restype = _C.c_int
                    lib.roctxRangePop.restype = _C.c_int
                    break
                except (OSError, AttributeError):
                    lib = None
            if lib is not None:
                break
        _state["lib"] = lib
    return _state["lib"]


def start_profiler(state="All", tracer_option="Default"):
    lib = _roctx()
    if lib is not None:
        lib.roctxRangePushA(("paddle_profiler_window state=%s tracer=%s" % (state, tracer_option)).encode())
    _state["open"] += 1
    _state["t0"] = _time.perf_counter()


def stop_profiler(sorted_key=None, profile_path="/tmp/profile"):
    if _state["open"] <= 0:
        return
    _state["open"] -= 1
    lib = _roctx()
    if lib is not None:
        lib.roctxRangePop()
    dt = _time.perf_counter() - (_state["t0"] or _time.perf_counter())
    print("[recengine] profiler window closed after %.3f s (ROCTX range %s; kernel timings come from rocprofv3, not from "
          "a Paddle profile file: %s is not written)" % (dt, "emitted" if lib is not None else "unavailable", profile_path),
          file=_sys.stderr, flush=True)


def reset_profiler():
    pass
