#!/bin/bash
# round 2, GPU call 16: dlrm + gpubox parity on the device
mkdir -p gpurun_out/r02_call16
o=gpurun_out/r02_call16
timeout 900 python -m pytest tests/test_dlrm.py tests/test_gpubox.py -x -q -m gpu > $o/pytest.txt 2>&1; echo "pytest rc=$?" >> $o/pytest.txt
tail -30 $o/pytest.txt
