#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04c2; mkdir -p "$O"; cd "$R"
timeout 600 python -m pytest tests/test_group_slots_gpu.py -x -q 2>&1 | tail -15 | tee "$O/pytest_group_slots.txt"
DBGS="0" bash tools/runs/r04_probe.sh 2>&1 | tee "$O/probe.txt"
