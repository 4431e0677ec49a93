"""paddle.distributed.fleet.base"""
from . import role_maker  # noqa: F401
