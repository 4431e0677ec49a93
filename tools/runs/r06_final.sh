#!/bin/bash
# evidence on the final round-6 tree: GPU suite + smoke, the default bench line, rocprofv3 stats / trace / PMC of bench.py,
# the step timeline, layout 2b kernel stats, L-trainer, the model benches
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06final; mkdir -p "$O"; cd "$R"
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > "$O/pytest_gpu.txt"
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 >> "$O/pytest_gpu.txt"
cat "$O/pytest_gpu.txt"
timeout 900 python bench.py 2> "$O/bench.err" | tail -1 > "$O/bench_line.json"
cut -c1-400 "$O/bench_line.json"
bash tools/profile_bench.sh r06 > "$O/profile.log" 2>&1
tail -5 "$O/profile.log"
f=$(find gpurun_out/prof_r06/trace -name "*kernel_stats.csv" | head -1); cp "$f" "$O/kernel_stats.csv"
t=$(find gpurun_out/prof_r06/trace -name "*kernel_trace.csv" | head -1); gzip -c "$t" > "$O/kernel_trace.csv.gz"
python tools/kernel_populations.py "$t" > "$O/populations.txt" 2>&1 || true
python tools/trace_timeline.py "$t" ctr_head_fold mid > "$O/step_timeline.txt" 2>&1 || true
cp gpurun_out/prof_r06/pmc_traffic.json "$O/pmc_traffic.json"
rm -rf gpurun_out/prof_r06/trace gpurun_out/prof_r06/pmc?
echo "== trainer level"; timeout 300 python tools/trainer_bench.py --lines 1048576 2>/dev/null | tail -1 > "$O/trainer_bench.json"; cut -c1-400 "$O/trainer_bench.json"
echo "== models"; timeout 600 python tools/bench_models.py > "$O/bench_models.txt" 2>&1; grep -E "^DIN|^DCN" "$O/bench_models.txt" | cut -c1-260
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace2b -o t --output-format csv -- python $R/bench.py --shared-table --dim 9 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $O/bench2b_under_rocprof.log 2>&1
f=$(find $O/trace2b -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_layout2b_D9.csv
rm -rf $O/trace2b
ls -la "$O"
