#!/bin/bash
# round 2, GPU call 19: kernel timeline of the world-1 sharded step
mkdir -p gpurun_out/r02_call19
o=$PWD/gpurun_out/r02_call19
repo=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $o/trace -o t --output-format csv -- python $repo/bench.py --force-sharded --no-cpu-baseline --steps 12 --warmup 6 > $o/bench.json 2> $o/bench.err
cd $repo
f=$(find $o/trace -name "*kernel_trace.csv" | head -1)
python tools/trace_timeline.py $f fm_fwd_ > $o/timeline.txt 2>&1
cat $o/timeline.txt | cut -c1-150
rm -f $f
