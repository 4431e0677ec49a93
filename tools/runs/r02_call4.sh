#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r02_call4
mkdir -p $out
cd $root
echo "== tests";  timeout 600 python -m pytest tests/test_slot_dnn.py -m gpu -q -x > $out/pytest.log 2>&1; tail -3 $out/pytest.log
echo "== slot_dnn";    timeout 300 python tools/slot_dnn_bench.py 2>$out/slot.err | tail -1 | tee $out/slot_dnn_adam.json | cut -c1-700
echo "== gemm lab";    timeout 900 python tools/gemm_lab/run.py --shapes fwd0,fwd1,dx0,dx1,cross,slot0,sq4096 2>&1 | grep -v amdgpu | tee $out/gemm_lab.txt
timeout 600 python tools/gemm_lab/run.py --shapes dw0,dw1 --splits 32,40 2>&1 | grep -v amdgpu | tee $out/gemm_lab_dw.txt
