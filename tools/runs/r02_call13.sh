#!/bin/bash
# round 2, GPU call 13: DIN compile-time-shaped forward — parity tests, then the three variants against the generic kernel
mkdir -p gpurun_out/r02_call13
o=gpurun_out/r02_call13
timeout 600 python -m pytest tests/test_din_gpu.py -x -q -m gpu > $o/pytest_din.txt 2>&1; echo "pytest rc=$?" >> $o/pytest_din.txt
tail -5 $o/pytest_din.txt
for v in pf2 nopf2 pf1; do
  REC_DIN_FWD_VARIANT=$v timeout 300 python tools/din_bench.py --cases 4096x512,4096x100,32x152 >> $o/din_bench.txt 2>&1
done
REC_DIN_FWD_GENERIC=1 timeout 300 python tools/din_bench.py --cases 4096x512,32x152 >> $o/din_bench.txt 2>&1
cat $o/din_bench.txt
