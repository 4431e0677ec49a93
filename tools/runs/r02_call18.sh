#!/bin/bash
# round 2, GPU call 18: gpubox GPU test; host trace of the world-1 sharded step and of the plain step
mkdir -p gpurun_out/r02_call18
o=gpurun_out/r02_call18
timeout 600 python -m pytest tests/test_gpubox.py -x -q -m gpu > $o/pytest.txt 2>&1; echo "pytest rc=$?" >> $o/pytest.txt
tail -4 $o/pytest.txt
timeout 600 python tools/host_trace.py --force-sharded --no-cpu-baseline --steps 40 --warmup 10 > $o/trace_sharded.txt 2>&1
timeout 600 python tools/host_trace.py --no-cpu-baseline --steps 40 --warmup 10 > $o/trace_plain.txt 2>&1
grep -v "^{" $o/trace_sharded.txt | tail -45
echo ==== plain; grep -v "^{" $o/trace_plain.txt | tail -25
