"""paddlerec_amd — MI355X-native sparse-embedding + feature-interaction engine for the hot path of
PaddleRec's models/rank CTR stack.  Kernels: paddlerec_amd/csrc/*.hip behind the
C-ABI of include/recengine.h; this package is the host-side mirror of the reference's plugin API."""
from ._lib import RecError, LIB_PATH  # noqa: F401

__version__ = "0.1.0"
