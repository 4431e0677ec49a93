#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
for wn in 1 2 1 2; do echo "== REC_X3_WN=$wn"; REC_X3_WN=$wn timeout 300 python tools/x3_wn_bench.py 2>&1 | grep -v amdgpu; done
timeout 600 python -m pytest tests/test_gemm_gpu.py -m gpu -x -q 2>&1 | tail -5
