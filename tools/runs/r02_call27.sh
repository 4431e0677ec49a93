#!/bin/bash
# round 2, GPU call 27: first thing on a fresh box — step time by block of 10 steps
mkdir -p gpurun_out/r02_call27
timeout 300 python tools/warmup_probe.py 2>/dev/null | tee gpurun_out/r02_call27/probe1.txt
timeout 300 python tools/warmup_probe.py 2>/dev/null | tee gpurun_out/r02_call27/probe2.txt
