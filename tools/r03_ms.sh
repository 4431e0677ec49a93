root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r03_ms
for o in ps adam; do timeout 300 python tools/slot_dnn_bench.py --opt $o 2>/dev/null | tail -1 > gpurun_out/r03_ms/slot_dnn_$o.json; cut -c1-400 gpurun_out/r03_ms/slot_dnn_$o.json; done
REC_MS_LANE=0 timeout 300 python tools/slot_dnn_bench.py --opt ps 2>/dev/null | tail -1 > gpurun_out/r03_ms/slot_dnn_ps_rowgroup_kernel.json; cut -c1-400 gpurun_out/r03_ms/slot_dnn_ps_rowgroup_kernel.json
timeout 800 bash tools/pmc.sh r03_ms/pmc multislot compute -- python $root/tools/slot_dnn_bench.py --opt ps --pool-only 6 2>&1 | tail -32
