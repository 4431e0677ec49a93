#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03c3; mkdir -p "$O"; cd "$R"
python tools/fm_cold_probe.py 2>&1 | grep -v amdgpu.ids | tee "$O/fm_cold_probe.txt"
for i in 1 2; do
  for nt in 0 1; do
    REC_SPARSE_NT=$nt timeout 120 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > "$O/bench_nt${nt}_$i.json"
  done
done
python - <<'PY'
import json, glob, os
o = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r03c3")
for f in sorted(glob.glob(os.path.join(o, "bench_*.json"))):
    try:
        d = json.loads(open(f).read())
        print(os.path.basename(f), "%.3f ms  %.2f M/s  frac %.3f in-step %.3f" % (d["ms_per_step"], d["value"] / 1e6, d["roofline"]["frac"], d["roofline"]["in_step_event"]["frac"]), {k: round(v, 3) for k, v in d["kernels_ms"].items()})
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
PY
