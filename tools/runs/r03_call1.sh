#!/bin/bash
# r03 GPU call 1: suite under REC_GEMM_PIPE=1, bench A/B (default vs pipe), kernel trace of the bench split into
# in-step / back-to-back populations, fm_fwd in-step probe.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03c1
mkdir -p "$O"
cd "$R"
REC_GEMM_PIPE=1 timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > "$O/pytest_gpu_pipe.txt"
cat "$O/pytest_gpu_pipe.txt"
for i in 1 2; do
  timeout 120 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > "$O/bench_default_$i.json"
  REC_GEMM_PIPE=1 timeout 120 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > "$O/bench_pipe_$i.json"
done
python - <<'PY'
import json, glob, os
o = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r03c1")
for f in sorted(glob.glob(os.path.join(o, "bench_*.json"))):
    try:
        d = json.loads(open(f).read())
        print(os.path.basename(f), "%.3f ms  %.2f M/s  frac %.3f in-step %.3f" % (d["ms_per_step"], d["value"] / 1e6, d["roofline"]["frac"], d["roofline"]["in_step_event"]["frac"]), {k: round(v, 3) for k, v in d["kernels_ms"].items()})
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
PY
timeout 200 python tools/fm_instep_probe.py 2>&1 | grep -v amdgpu.ids | tee "$O/fm_instep_probe.txt"
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d "$O/trace" -o t -- python "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$O/bench_under_rocprof.log" 2>&1
f=$(find "$O/trace" -name "*kernel_trace.csv" | head -1)
cd "$R"
[ -n "$f" ] && python tools/kernel_populations.py "$f" fm_fwd_kernel fm_bwd_kernel sparse_adam_record_kernel | tee "$O/populations.txt"
[ -n "$f" ] && python tools/trace_timeline.py "$f" > "$O/timeline.txt" 2>&1
[ -n "$f" ] && gzip -c "$f" > "$O/kernel_trace.csv.gz" && rm -rf "$O/trace"
ls -la "$O"
