"""The C-ABI shared library loads without a GPU and exports every symbol include/recengine.h declares."""
import os
import re

from conftest import REPO


def _declared():
    src = open(os.path.join(REPO, "include", "recengine.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(rec_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_header_symbols_exported(engine_lib):
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(engine_lib, n), "missing export: " + n


def test_python_signature_table_covers_header(engine_lib):
    from paddlerec_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()


def test_error_reporting_without_gpu(engine_lib):
    """Argument validation happens before any launch: callable on a GPU-less host."""
    import ctypes as C
    from paddlerec_amd import _lib
    n = C.c_size_t(0)
    rc = engine_lib.rec_ids_group_workspace_bytes(-5, 10, C.byref(n))
    assert rc == -1 and b"bad" in engine_lib.rec_last_error()
    d = _lib.DeepFMDesc(4, 26, 99, 16, 16, 10, 0)        # num_dense too large
    rc = engine_lib.rec_deepfm_fm_bwd_workspace_bytes(C.byref(d), C.byref(n))
    assert rc == -2
    assert engine_lib.rec_version() >= 100


def test_product_never_imports_oracle():
    """The product path must not route through the oracle (or any CPU fallback)."""
    pkg = os.path.join(REPO, "paddlerec_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(root, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
                assert "liboracle" not in txt, f
