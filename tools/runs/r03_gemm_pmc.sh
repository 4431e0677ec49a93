#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
bash tools/pmc.sh r03_pmc_ring "gemm_f32_glds" compute -- python $R/tools/gemm_one.py 131072 400 1600 0 bias_relu 4 > /dev/null 2>&1
cp gpurun_out/r03_pmc_ring/summary.txt gpurun_out/r03_pmc_ring_summary.txt
REC_GEMM_GLDS=0 bash tools/pmc.sh r03_pmc_reg "gemm_f32_pipe" compute -- python $R/tools/gemm_one.py 131072 400 1600 0 bias_relu 4 > /dev/null 2>&1
cp gpurun_out/r03_pmc_reg/summary.txt gpurun_out/r03_pmc_reg_summary.txt
rm -rf gpurun_out/r03_pmc_ring gpurun_out/r03_pmc_reg
paste gpurun_out/r03_pmc_ring_summary.txt gpurun_out/r03_pmc_reg_summary.txt | cut -c1-200
