#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r02_call6
mkdir -p $out
cd $root
echo "== auto";  python tools/gemm_lab/run.py --engine-only 2>&1 | grep "^==" | tee $out/auto.txt
for c in 0 1 2 3 5 6; do echo "== forced cfg $c"; REC_GEMM_FORCE_CFG=$c python tools/gemm_lab/run.py --engine-only --shapes fwd0,fwd1,dx0,dx1,cross,slot0 2>&1 | grep "^==" | cut -c1-110 | tee $out/cfg$c.txt; done
for c in 3 4 5; do echo "== forced cfg $c (dW)"; REC_GEMM_FORCE_CFG=$c python tools/gemm_lab/run.py --engine-only --shapes dw0,dw1 2>&1 | grep "^==" | cut -c1-110 | tee $out/dwcfg$c.txt; done
echo "== gemm tests"; timeout 600 python -m pytest tests/test_gemm_gpu.py -m gpu -q -x 2>&1 | tail -2
