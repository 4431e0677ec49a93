#!/bin/bash
# r04 GPU call 1: the slot-local grouping + sorted row gradients: parity tests, standalone grouping time,
# bench A/B against the general sort, kernel stats of the new step.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04c1
mkdir -p "$O"
cd "$R"
timeout 600 python -m pytest tests/test_group_slots_gpu.py -x -q 2>&1 | tail -15 | tee "$O/pytest_group_slots.txt"
timeout 600 python -m pytest tests/test_deepfm_gpu.py tests/test_row_update_shapes_gpu.py -x -q 2>&1 | tail -5 | tee "$O/pytest_deepfm.txt"
timeout 120 python tools/group_bench.py 2>&1 | grep -v amdgpu.ids | tee "$O/group_bench.txt"
for i in 1 2; do
  REC_DEEPFM_SORTED=0 timeout 200 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 > "$O/bench_general_$i.json"
  timeout 200 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 > "$O/bench_sorted_$i.json"
done
python - <<'PY'
import json, glob, os
o = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r04c1")
for f in sorted(glob.glob(os.path.join(o, "bench_*.json"))):
    try:
        d = json.loads(open(f).read())
        print(os.path.basename(f), "%.3f ms  %.2f M/s  frac %.3f" % (d["ms_per_step"], d["value"] / 1e6, d["roofline"]["frac"]), {k: round(v, 3) for k, v in d.get("kernels_ms", {}).items()})
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -o t -- python "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > "$O/bench_under_rocprof.log" 2>&1
f=$(find "$O/trace" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$O/kernel_stats.csv" && head -45 "$f" | cut -c1-200
t=$(find "$O/trace" -name "*kernel_trace.csv" | head -1)
cd "$R"
[ -n "$t" ] && python tools/kernel_populations.py "$t" fm_fwd_kernel fm_bwd_kernel sparse_adam_record_kernel | tee "$O/populations.txt"
rm -rf "$O/trace"
ls -la "$O"
