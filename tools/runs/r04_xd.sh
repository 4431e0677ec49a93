#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest $R/tests/test_xdeepfm.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do for kp in 0 1; do echo -n "REC_CIN_KEEP=$kp  xDeepFM B 65536 ms per step (chunk 4096 MB): "; REC_CIN_KEEP=$kp timeout 300 python $R/tools/xdeepfm_chunk_bench.py 4096 2>&1 | grep -v amdgpu | tail -1; done; done | tee $O/xd_keep.txt
