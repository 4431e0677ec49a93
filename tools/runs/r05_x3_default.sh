#!/bin/bash
# bf16 x 3 as the library default: the whole GPU suite + smoke, then the default bench line (with other_configs)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/x3; mkdir -p "$O"; cd "$R"
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee "$O/suite_default_on.txt"
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a "$O/suite_default_on.txt"
timeout 900 python bench.py 2> "$O/bench.err" | tail -1 > "$O/bench_line.json"
python - <<'P'
import json
d=json.load(open("gpurun_out/x3/bench_line.json"))
print({k: d[k] for k in ("value","ms_per_step","dtype")}); print(d["roofline"]["frac"], d["mlp_gemm"]); print(d.get("cpu_baseline",{}).get("value"))
for c in d.get("other_configs", []):
    print("  ", (c.get("config") or c.get("command")), (c.get("workload") or "")[:70], c.get("ms") or c.get("ms_per_step") or c.get("train_step_ms") or c.get("error"))
P
