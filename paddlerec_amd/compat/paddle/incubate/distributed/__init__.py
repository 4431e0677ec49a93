"""paddle.incubate.distributed (namespace package of the compat layer)."""
from . import fleet  # noqa: F401
