#!/bin/bash
# end-of-round evidence run (one gpurun call): GPU suite, smoke, bench line, rocprofv3 kernel stats + PMC traffic of the same tree
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r03_final; rm -rf $out; mkdir -p $out
cd $root
timeout 1500 python -m pytest tests -q -m gpu > $out/pytest_gpu.txt 2>&1; tail -3 $out/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a $out/pytest_gpu.txt
timeout 600 python bench.py > $out/bench.log 2>&1; tail -1 $out/bench.log > $out/bench.json; cut -c1-400 $out/bench.json
timeout 900 bash tools/profile_bench.sh r03 > $out/profile.log 2>&1; tail -5 $out/profile.log
