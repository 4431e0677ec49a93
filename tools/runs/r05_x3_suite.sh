#!/bin/bash
# bf16 x 3 GEMM with the round-to-nearest split: lab (accuracy / time, plain and non-temporal stores), GEMM tests, the whole
# GPU suite with the switch ON (which tests notice)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/x3; mkdir -p "$O"; cd "$R"
(timeout 200 python tools/gemm_lab/bf16x3_lab.py --rounds 2 --only fwd0,fwd1,dx1,edge; echo "--- non-temporal C stores"; timeout 100 python tools/gemm_lab/bf16x3_lab.py --rounds 2 --lib tools/gemm_lab/_build/libx3lab_nt.so --only fwd1,dx1) 2>&1 | grep -v amdgpu.ids | tee "$O/call3_lab.txt"
timeout 300 python -m pytest tests/test_gemm_gpu.py -x -q 2>&1 | tail -25 | tee "$O/test_gemm.txt"
REC_GEMM_BF16X3=1 timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40 | tee "$O/suite_x3_on.txt"
