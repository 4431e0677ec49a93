#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r02_call12
mkdir -p $out
cd $root
timeout 600 bash tools/pmc.sh r02_call12/pmc_din din_attention_fwd compute -- python $root/tools/din_loop.py > $out/pmc_din.log 2>&1; tail -36 $out/pmc_din.log
