#!/bin/bash
# per-stream kernel timeline of one bench step; env passes through (e.g. REC_DEEPFM_PIPELINED=1); $1 = tag
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-trace}; O=$R/gpurun_out/r06$T; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-other-configs > $O/log.txt 2>&1
t=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_timeline.py "$t" ctr_head > $O/timeline.txt
head -70 $O/timeline.txt
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); head -40 "$f" > $O/kernel_stats_head.csv
rm -rf $O/trace
