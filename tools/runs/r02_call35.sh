#!/bin/bash
# round 2, GPU call 35: forward GEMM throughput against the number of block rounds
mkdir -p gpurun_out/r02_call35
timeout 300 python tools/gemm_rounds.py 2>/dev/null | tee gpurun_out/r02_call35/gemm_rounds.txt
