# round 6: the row-bucket merge of the launch-bound step on ONE shared table (sparse_bucket_kernel) — parity tests, A/B, timeline
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp; mkdir -p gpurun_out/r06bucket
out=gpurun_out/r06bucket
timeout 900 python -m pytest tests/test_row_update_shapes_gpu.py tests/test_deepfm_gpu.py tests/test_deepfm_step_c.py -m gpu -x -q 2>&1 | tail -15 > $out/pytest.txt
ms() { python -c "import sys,json; [print('$1 %.4f ms' % json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"; }
for rep in 1 2; do
for b in 0 16 32 64 128 256; do
echo "== REC_SMALL_BUCKET_ROWS=$b (0: REC_SMALL_BUCKET=0)" >> $out/ab.txt
if [ $b = 0 ]; then export REC_SMALL_BUCKET=0; else export REC_SMALL_BUCKET=1 REC_SMALL_BUCKET_ROWS=$b; fi
python bench.py --batch 512 --shared-table --dim 9 --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs 2>&1 | ms "shared D9 bs512" >> $out/ab.txt
python bench.py --batch 512 --shared-table --dim 16 --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs 2>&1 | ms "shared D16 bs512" >> $out/ab.txt
done; done
unset REC_SMALL_BUCKET REC_SMALL_BUCKET_ROWS
python bench.py --batch 512 --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs 2>&1 | ms "26 tables bs512" >> $out/ab.txt
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/trace -o t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --batch 512 --shared-table --dim 9 --steps 100 --warmup 20 --no-cpu-baseline --no-other-configs > $GRAFT_REPO_ROOT/$out/rocprof2b.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_timeline.py $(find $out/trace -name '*kernel_trace.csv' | head -1) fm_fwd_ > $out/timeline2b.txt
rm -rf $out/trace
cat $out/pytest.txt $out/ab.txt $out/timeline2b.txt
