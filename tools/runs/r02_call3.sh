#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r02_call3
mkdir -p $out
cd $root
echo "== tests";  timeout 600 python -m pytest tests/test_slot_dnn.py tests/test_deepfm_gpu.py -m gpu -q -x > $out/pytest.log 2>&1; tail -4 $out/pytest.log
echo "== slot_dnn";    timeout 300 python tools/slot_dnn_bench.py 2>$out/slot.err | tail -1 | tee $out/slot_dnn_adam.json | cut -c1-900
echo "== pool pmc";    timeout 600 bash tools/pmc.sh r02_call3/pmc_pool multislot compute -- python $root/tools/slot_dnn_bench.py --pool-only 6 > $out/pmc_pool.log 2>&1; tail -40 $out/pmc_pool.log
echo "== gemm lab dW splits";    timeout 600 python tools/gemm_lab/run.py --shapes dw0,dw1 --splits 24,32,40,48,64,80,96 2>&1 | grep -v amdgpu | tee $out/gemm_lab_dw.txt
