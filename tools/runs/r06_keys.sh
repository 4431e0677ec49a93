#!/bin/bash
# keys_kernel with eight ids in flight per thread: grouping tests, kernel durations of a bench run under rocprofv3, three bench lines
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06keys; mkdir -p "$O"; cd "$R"
timeout 900 python -m pytest tests/test_group_slots_gpu.py tests/test_deepfm_gpu.py -x -q -m gpu 2>&1 | tail -2 | tee "$O/pytest.txt"
for i in 1 2 3; do
  timeout 600 python bench.py --no-other-configs --no-cpu-baseline 2>> "$O/bench.err" | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step %.4f' % d['ms_per_step'], {k: round(v, 4) for k, v in d['kernels_ms'].items()})"
done | tee "$O/lines.txt"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/rocprof.log 2>&1
f=$(find $O/trace -name "*kernel_stats.csv" | head -1)
grep -E "sg::|fm_fwd_kernel|gemm_bf16x3_kernel<13, 2" "$f" | cut -d, -f1-4 | cut -c1-150 | tee "$O/kernels.txt"
t=$(find $O/trace -name "*kernel_trace.csv" | head -1); python $R/tools/trace_timeline.py "$t" ctr_head_fold mid > "$O/step_timeline.txt" 2>&1; grep -E "sg::|fm_fwd|step length|13, 2, 4" "$O/step_timeline.txt" | cut -c1-120
rm -rf $O/trace
