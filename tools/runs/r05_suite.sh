#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05suite; mkdir -p "$O"; cd "$R"
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee "$O/pytest_gpu.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee "$O/smoke.txt"
