# round 6: kernel timelines of the launch-bound steps (DeepFM bs 512 on 26 tables and on one shared table, DIN bs 32)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp; out=$GRAFT_REPO_ROOT/gpurun_out/r06small; mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats -d $out/t1 -o t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --batch 512 --steps 100 --warmup 20 --no-cpu-baseline --no-other-configs > $out/bench512.log 2>&1
rocprofv3 --kernel-trace --stats -d $out/t2 -o t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --batch 512 --shared-table --dim 9 --steps 100 --warmup 20 --no-cpu-baseline --no-other-configs > $out/bench512s.log 2>&1
rocprofv3 --kernel-trace --stats -d $out/t3 -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/din_small_bench.py > $out/din32.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_timeline.py $(find $out/t1 -name '*kernel_trace.csv' | head -1) ctr_head_fold > $out/timeline_bs512.txt
python tools/trace_timeline.py $(find $out/t2 -name '*kernel_trace.csv' | head -1) ctr_head_fold > $out/timeline_bs512_shared.txt
python tools/trace_timeline.py $(find $out/t3 -name '*kernel_trace.csv' | head -1) din_att_fwd > $out/timeline_din32.txt || python tools/trace_timeline.py $(find $out/t3 -name '*kernel_trace.csv' | head -1) attention > $out/timeline_din32.txt
cp $(find $out/t3 -name '*kernel_stats.csv' | head -1) $out/din32_stats.csv
rm -rf $out/t1 $out/t2 $out/t3
cat $out/timeline_bs512.txt $out/timeline_bs512_shared.txt; head -40 $out/timeline_din32.txt; tail -3 $out/din32.log
