import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(autouse=True)
def _seed_everything():
    """Models draw their initial weights from torch's global generator: pin it so that every test sees the same
    parameters on every run (a parity failure must be reproducible, not a draw of the initialisation)."""
    import torch
    torch.manual_seed(20250404)
    yield


@pytest.fixture(scope="session")
def oracle_lib():
    """ctypes handle of oracle/_build/liboracle.so (built on demand with gcc)."""
    import ctypes as C
    path = os.path.join(REPO, "oracle", "_build", "liboracle.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle")])
    return C.CDLL(path)


@pytest.fixture(scope="session")
def engine_lib():
    """librecengine.so, built in-tree with hipcc if missing (cross-compiles without a GPU)."""
    from paddlerec_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build()
    return _lib.lib()
