#!/bin/bash
# round 2, GPU call 32: PMC compute counters of the bench's three GEMM shapes
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r02_call32
mkdir -p $out
cd $root
timeout 900 bash tools/pmc.sh r02_call32/pmc_gemm gemm_f32_kernel compute -- python $root/tools/gemm_loop.py > $out/pmc_gemm.log 2>&1; tail -100 $out/pmc_gemm.log | cut -c1-140
