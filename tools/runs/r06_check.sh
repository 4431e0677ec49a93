#!/bin/bash
# flakiness check on the final tree: the GPU suite three times + smoke; the default bench line three times
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06check; mkdir -p "$O"; cd "$R"
: > "$O/pytest_gpu.txt"
for i in 1 2 3; do
  timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3 >> "$O/pytest_gpu.txt"
done
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 >> "$O/pytest_gpu.txt"
cat "$O/pytest_gpu.txt"
for i in 1 2 3; do
  timeout 900 python bench.py --no-other-configs --no-cpu-baseline 2>> "$O/bench.err" | tail -1 | cut -c1-330
done | tee "$O/bench_lines.txt"
