# round 6, final tree: the launch-bound steps (DeepFM bs 512 on 26 slot tables and on the reference's one shared table, DIN bs 32)
# — switch by switch, and the kernel timeline of one step each (rocprofv3 --kernel-trace)  -> profiles/r06_small_batch.txt
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp; out=$GRAFT_REPO_ROOT/gpurun_out/r06smallfinal; mkdir -p $out; rm -f $out/*.txt
ms() { python -c "import sys,json; [print('  %-78s %.4f ms per step' % ('$1', json.loads(l)['ms_per_step'])) for l in sys.stdin if l.startswith('{')]"; }
B="--steps 300 --warmup 30 --no-cpu-baseline --no-other-configs"
{
echo "== DeepFM bs 512 (deepfm/config_bigdata.yaml:23), bench.py, 300 timed steps; switches cumulative from the round-5 step to the default"
for lay in "--batch 512" "--batch 512 --shared-table --dim 9"; do
echo "-- bench.py $lay"
REC_GEMM_DIRECT=0 REC_SMALL_BUCKET=0 REC_SMALL_C_STEP=0 python bench.py $lay $B 2>/dev/null | ms "round-5 step: tiled GEMMs + split-K reduces, wave-per-lookup merge (REC_GEMM_DIRECT=0 ...)"
REC_SMALL_BUCKET=0 REC_SMALL_C_STEP=0 python bench.py $lay $B 2>/dev/null | ms "+ one-launch GEMMs, dW / dX pair (REC_SMALL_BUCKET=0 REC_SMALL_C_STEP=0)"
REC_SMALL_C_STEP=0 python bench.py $lay $B 2>/dev/null | ms "+ merge + update by row buckets (REC_SMALL_C_STEP=0: recorded call list)"
REC_SMALL_TAIL=0 python bench.py $lay $B 2>/dev/null | ms "  the same launches from rec_deepfm_train_step (REC_SMALL_TAIL=0)"
python bench.py $lay $B 2>/dev/null | ms "+ folds, dense Adam, the weight fold of layer 0 as roles (default)"
done
echo "== DIN bs 32 T 152 (din/config.yaml:20), tools/din_small_bench.py"
echo "-- REC_SMALL_TAIL=0"; REC_SMALL_TAIL=0 python tools/din_small_bench.py 2>/dev/null | tail -2
echo "-- default"; python tools/din_small_bench.py 2>/dev/null | tail -2
} > $out/ab.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats -d $out/t1 -o t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --c-step --batch 512 --steps 100 --warmup 20 --no-cpu-baseline --no-other-configs > $out/b1.log 2>&1
rocprofv3 --kernel-trace --stats -d $out/t2 -o t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --c-step --batch 512 --shared-table --dim 9 --steps 100 --warmup 20 --no-cpu-baseline --no-other-configs > $out/b2.log 2>&1
rocprofv3 --kernel-trace --stats -d $out/t3 -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/din_c_loop.py 200 > $out/b3.log 2>&1
cd $GRAFT_REPO_ROOT
{
echo "== kernel timeline of one step, rec_deepfm_train_step, bs 512 on 26 slot tables (D 16)"
python tools/trace_timeline.py $(find $out/t1 -name '*kernel_trace.csv' | head -1) ctr_head_kernel
echo "== ... on ONE shared table (D 9: the reference's layout)"
python tools/trace_timeline.py $(find $out/t2 -name '*kernel_trace.csv' | head -1) ctr_head_kernel
echo "== rec_din_train_step, bs 32, T 152"
python tools/trace_timeline.py $(find $out/t3 -name '*kernel_trace.csv' | head -1) din_attention_fwd
} > $out/timelines.txt 2>&1
rm -rf $out/t1 $out/t2 $out/t3 $out/b?.log
cat $out/ab.txt $out/timelines.txt
