#!/bin/bash
# round 2, GPU call 30: CIN contract kernels (Y association) — parity, then the xDeepFM train step
mkdir -p gpurun_out/r02_call30
o=gpurun_out/r02_call30
timeout 600 python -m pytest tests/test_xdeepfm.py -x -q -m gpu > $o/pytest.txt 2>&1; echo "pytest rc=$?" >> $o/pytest.txt; tail -3 $o/pytest.txt
timeout 900 python tools/bench_models.py > $o/models.txt 2>&1
grep -v "^{" $o/models.txt | grep -i "xdeepfm\|dlrm"
