#!/bin/bash
# round-2 evidence run: rocprofv3 stats + PMC of bench.py, DCN-v2 / DIN / slot_dnn kernel stats
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r02_profile
mkdir -p $out
cd $root
bash tools/profile_bench.sh r02 > $out/profile_bench.log 2>&1; tail -30 $out/profile_bench.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $out/models -o m --output-format csv -- python $root/tools/bench_models.py > $out/models_under_rocprof.log 2>&1; echo "models trace rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats -d $out/slot -o s --output-format csv -- python $root/tools/slot_dnn_bench.py > $out/slot_under_rocprof.log 2>&1; echo "slot trace rc=$?"
cd $root
timeout 200 python -u bench.py 2>/dev/null | grep "^{" > $out/bench_plain.json
ls $out $out/models $out/slot | head -40
