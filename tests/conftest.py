import os
import sys

# tests execute files of /root/reference (read-only input): neither this process nor its children write __pycache__
# next to them
sys.dont_write_bytecode = True
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "first_hw_run: written while no GPU time was left — has not run on hardware yet; "
                                       "collected AFTER the tests that have (drop the mark once it has passed there)")


def pytest_collection_modifyitems(config, items):
    import torch
    # tests that have never run on hardware go last: with `-x` a surprise in one of them must not hide the others
    # (among themselves: the ones built from already-verified entry points first)
    order = ["test_multi_value_slots_file_to_sum_pool", "test_fm_layer_gpu", "test_wide_deep_layer_gpu",
             "test_dnn_layer_gpu", "test_autograd_adapter_gpu", "test_other_mirrors_through_the_loops_gpu",
             "test_train_infer_resume_gpu", "test_stream_helpers"]
    rank = {n: i + 1 for i, n in enumerate(order)}
    items.sort(key=lambda it: (rank.get(it.originalname or it.name, len(order) + 1) if "first_hw_run" in it.keywords
                               else 0))
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
