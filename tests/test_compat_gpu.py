"""The product `paddle` compat namespace on the GPU (paddlerec_amd/compat): Embedding / Linear / Adam written against
the Paddle API run on the HIP kernels and match the same net in plain torch fp32.  (The reference's own trainer runs
over this namespace in tests/test_reference_entrypoint.py — build container only: /root/reference is not on the GPU box.)"""
import os
import subprocess
import sys

import pytest

from conftest import REPO


@pytest.mark.gpu
def test_compat_layers_and_adam_on_hip_kernels(engine_lib):
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "_compat_gpu_script.py")], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "compat gpu ok" in r.stdout
