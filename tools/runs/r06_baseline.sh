#!/bin/bash
# round 6, first call: the round-5 tree's bench line (no other configs) and the step timeline on this round's box
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06base; mkdir -p "$O"; cd "$R"
timeout 600 python bench.py --no-other-configs --no-cpu-baseline 2> "$O/bench.err" | tail -1 > "$O/bench_line.json"
cut -c1-600 "$O/bench_line.json"
bash tools/runs/r05_x3_trace.sh > "$O/trace.log" 2>&1
cp gpurun_out/x3trace/timeline.txt "$O/timeline.txt"; head -60 "$O/timeline.txt"
