#!/bin/bash
# round 2, GPU call 14: DIN compile-time-shaped backward on saved activations — parity, then timing against the recompute kernel
mkdir -p gpurun_out/r02_call14
o=gpurun_out/r02_call14
timeout 900 python -m pytest tests/test_din_gpu.py tests/test_models_fullsize_gpu.py -x -q -m gpu > $o/pytest_din.txt 2>&1; echo "pytest rc=$?" >> $o/pytest_din.txt
tail -5 $o/pytest_din.txt
timeout 300 python tools/din_bench.py --bwd --cases 4096x512,4096x100,32x152 >> $o/din_bench.txt 2>&1
REC_DIN_BWD_GENERIC=1 timeout 300 python tools/din_bench.py --bwd --cases 4096x512,32x152 >> $o/din_bench.txt 2>&1
cat $o/din_bench.txt
