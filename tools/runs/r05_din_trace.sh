# kernel trace + per-queue timeline of the DIN train step at B 4096 (tools/din_step_loop.py), T 100
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/din_trace
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for T in ${DIN_T:-100}; do
timeout 200 rocprofv3 --kernel-trace --stats -d $out/t$T -o t --output-format csv -- python $root/tools/din_step_loop.py 4096 $T 30 > $out/log$T.txt 2>&1
python $root/tools/trace_timeline.py $out/t$T/t_kernel_trace.csv din_attention_fwd_ct > $out/timeline$T.txt
cat $out/timeline$T.txt
done
