#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
timeout 600 python -m pytest ${TESTS:-tests/test_group_slots_gpu.py} -m gpu -x -q 2>&1 | tail -${TAIL:-15}
