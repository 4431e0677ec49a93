#!/bin/bash
# per-stream kernel timeline of one bench step on the bf16 x 3 tree
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/x3trace; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-other-configs > $O/log.txt 2>&1
t=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_timeline.py "$t" ctr_head > $O/timeline.txt
head -90 $O/timeline.txt
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); head -30 "$f" > $O/kernel_stats_head.csv
rm -rf $O/trace
