"""paddle.utils: the profiler hooks tools/profiler.py names (only entered with --profiler_options)."""
from . import profiler  # noqa: F401
