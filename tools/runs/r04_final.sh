#!/bin/bash
# evidence on the final round-4 tree: GPU suite + smoke, the default bench line, rocprofv3 stats / trace / PMC of bench.py
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04final; mkdir -p "$O"; cd "$R"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > "$O/pytest_gpu.txt"
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 >> "$O/pytest_gpu.txt"
cat "$O/pytest_gpu.txt"
timeout 900 python bench.py 2> "$O/bench.err" | tail -1 > "$O/bench_line.json"
cut -c1-400 "$O/bench_line.json"
bash tools/profile_bench.sh r04 > "$O/profile.log" 2>&1
tail -5 "$O/profile.log"
f=$(find gpurun_out/prof_r04/trace -name "*kernel_stats.csv" | head -1); cp "$f" "$O/kernel_stats.csv"
t=$(find gpurun_out/prof_r04/trace -name "*kernel_trace.csv" | head -1); gzip -c "$t" > "$O/kernel_trace.csv.gz"
python tools/kernel_populations.py "$t" > "$O/populations.txt" 2>&1 || true
cp gpurun_out/prof_r04/pmc_traffic.json "$O/pmc_traffic.json"
rm -rf gpurun_out/prof_r04/trace gpurun_out/prof_r04/pmc?
ls -la "$O"
