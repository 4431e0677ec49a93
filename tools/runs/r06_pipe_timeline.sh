# round 6: kernel timelines of the headline step under the default and the pipelined schedule (REC_DEEPFM_PIPELINED=1)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp; out=$GRAFT_REPO_ROOT/gpurun_out/r06pipe3; mkdir -p $out
ms() { python -c "import sys,json; [print('$1 %.4f ms' % json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"; }
for rep in 1 2; do
for p in 0 1; do
REC_DEEPFM_PIPELINED=$p python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | ms "pipelined=$p" >> $out/ab.txt
done; done
cd /tmp
for p in 0 1; do
REC_DEEPFM_PIPELINED=$p rocprofv3 --kernel-trace --stats -d $out/t$p -o t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-other-configs > $out/bench$p.log 2>&1
python $GRAFT_REPO_ROOT/tools/trace_timeline.py $(find $out/t$p -name '*kernel_trace.csv' | head -1) ctr_head_fold > $out/timeline_pipe$p.txt
rm -rf $out/t$p
done
cat $out/ab.txt $out/timeline_pipe1.txt
