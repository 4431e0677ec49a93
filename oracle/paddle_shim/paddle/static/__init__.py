"""paddle.static pieces the reference's slot_dnn/net.py touches (oracle-side shim, see ../__init__.py)."""
from . import nn  # noqa: F401
