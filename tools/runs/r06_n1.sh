#!/bin/bash
# round 6: the lookup / sum-pool / PS pull-push custom operators on the GPU (shim == ops.py, patched trees == unpatched == oracle)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06n1; mkdir -p "$O"; cd "$R"
timeout 1500 python -m pytest tests/test_paddle_custom_ops.py tests/test_slot_dnn_custom_ops.py tests/test_reference_gpubox_entrypoint.py tests/test_slot_dnn.py -m gpu -x -q 2>&1 | tail -25 > "$O/pytest.txt"
cat "$O/pytest.txt"
