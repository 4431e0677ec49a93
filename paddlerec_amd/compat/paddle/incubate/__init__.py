"""paddle.incubate (namespace package of the compat layer)."""
from . import distributed  # noqa: F401
