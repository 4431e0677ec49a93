#!/bin/bash
# round 2, GPU call 15: xdeepfm parity (CIN kernels, layer vs golden / oracle), DIN train step timing
mkdir -p gpurun_out/r02_call15
o=gpurun_out/r02_call15
timeout 900 python -m pytest tests/test_xdeepfm.py -x -q -m gpu > $o/pytest_xdeepfm.txt 2>&1; echo "pytest rc=$?" >> $o/pytest_xdeepfm.txt
tail -25 $o/pytest_xdeepfm.txt
timeout 600 python tools/bench_models.py > $o/models.txt 2>&1; tail -12 $o/models.txt | cut -c1-260
