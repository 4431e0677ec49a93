#!/bin/bash
# round 2, GPU call 1: new kernels' parity tests first, then the whole gpu suite, FM sweep, bench, slot_dnn bench
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r02_call1
mkdir -p $out
cd $root
echo "== new tests";   timeout 600 python -m pytest tests/test_slot_dnn.py -m gpu -q -x > $out/pytest_slot_dnn.log 2>&1; tail -5 $out/pytest_slot_dnn.log
echo "== gpu suite";   timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_slot_dnn.py > $out/pytest_gpu.log 2>&1; tail -5 $out/pytest_gpu.log
echo "== bench";       timeout 300 python -u bench.py 2>$out/bench.err | grep "^{" > $out/bench.json; cut -c1-400 $out/bench.json
echo "== fm sweep";    timeout 600 python tools/fm_sweep.py 2>&1 | grep -v amdgpu | tee $out/fm_sweep.txt
echo "== slot_dnn";    timeout 300 python tools/slot_dnn_bench.py 2>$out/slot.err | tail -1 | tee $out/slot_dnn_adam.json | cut -c1-600
timeout 300 python tools/slot_dnn_bench.py --opt ps 2>>$out/slot.err | tail -1 | tee $out/slot_dnn_ps.json | cut -c1-600
nproc > $out/nproc.txt
