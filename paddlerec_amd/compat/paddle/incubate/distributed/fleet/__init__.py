"""paddle.incubate.distributed.fleet (namespace package of the compat layer)."""
from . import fleet_util  # noqa: F401
