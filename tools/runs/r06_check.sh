#!/bin/bash
# re-entry check: GPU suite + smoke + the default bench line on the current tree
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06check; mkdir -p "$O"; cd "$R"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > "$O/pytest_gpu.txt"
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 >> "$O/pytest_gpu.txt"
cat "$O/pytest_gpu.txt"
timeout 900 python bench.py 2> "$O/bench.err" | tail -1 > "$O/bench_line.json"
cut -c1-600 "$O/bench_line.json"
