#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04suite; mkdir -p "$O"; cd "$R"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee "$O/pytest_gpu.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee "$O/smoke.txt"
