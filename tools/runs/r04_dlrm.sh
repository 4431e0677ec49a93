#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 600 python -m pytest $R/tests/test_dlrm.py -m gpu -x -q 2>&1 | tail -2
$R/tools/runs/r04_dlrm_trace.sh 2>&1 | grep -E "DLRM B|bn_colreduce"
