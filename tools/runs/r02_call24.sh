#!/bin/bash
# round 2, GPU call 24: sibling-net train steps (xDeepFM, DLRM) in tools/bench_models.py
mkdir -p gpurun_out/r02_call24
timeout 900 python tools/bench_models.py > gpurun_out/r02_call24/models.txt 2>&1
grep -v "^{" gpurun_out/r02_call24/models.txt | tail -8
