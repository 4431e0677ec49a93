#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03c5; mkdir -p "$O"; cd "$R"
timeout 400 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -25 | tee "$O/pytest_gpu.txt"
timeout 300 python bench.py 2>"$O/bench.err" | tail -1 > "$O/bench.json"
python - <<'PY'
import json, os
o = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r03c5")
d = json.loads(open(os.path.join(o, "bench.json")).read())
print("%.3f ms  %.2f M/s  frac %.3f in-step %.3f" % (d["ms_per_step"], d["value"] / 1e6, d["roofline"]["frac"], d["roofline"]["in_step_event"]["frac"]), {k: round(v, 3) for k, v in d["kernels_ms"].items()})
print(json.dumps(d["cpu_baseline"], indent=1))
PY
tail -5 "$O/bench.err"
