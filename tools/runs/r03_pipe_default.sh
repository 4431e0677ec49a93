#!/bin/bash
# First GPU call of the next round: decide whether gemm_f32_pipe_kernel becomes the default.
#   gpurun --timeout 900 -- 'bash tools/runs/r03_pipe_default.sh'
# Writes gpurun_out/r03/: the full GPU suite with REC_GEMM_PIPE=1, bench.py with and without it (two runs each,
# alternating), and the kernel stats of the flagged bench.  Make it the default only if the suite is green and the
# flagged bench is faster in both pairs.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03
mkdir -p "$O"
cd "$R"
REC_GEMM_PIPE=1 timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > "$O/pytest_gpu_pipe.txt"
cat "$O/pytest_gpu_pipe.txt"
for i in 1 2; do
  timeout 120 python bench.py 2>/dev/null | tail -1 > "$O/bench_default_$i.json"
  REC_GEMM_PIPE=1 timeout 120 python bench.py 2>/dev/null | tail -1 > "$O/bench_pipe_$i.json"
done
python - <<'PY'
import json, glob, os
o = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r03")
for f in sorted(glob.glob(os.path.join(o, "bench_*.json"))):
    try:
        d = json.loads(open(f).read())
        print(os.path.basename(f), "%.3f ms  %.2f M samples/s  frac %.3f" % (d["ms_per_step"], d["value"] / 1e6, d["roofline"]["frac"]))
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
PY
cd /tmp && export TMPDIR=/tmp
REC_GEMM_PIPE=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_pipe" -o p -- python "$R/bench.py" > "$O/bench_pipe_under_rocprof.log" 2>&1
f=$(find "$O/prof_pipe" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -12 "$f" | cut -c1-60,180-330
