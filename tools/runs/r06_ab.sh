#!/bin/bash
# generic bench A/B: each argument is "tag:ENV=val,ENV2=val"; three interleaved rounds
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06ab; mkdir -p "$O"; cd "$R"
for i in 1 2 3; do for spec in "$@"; do
  tag=${spec%%:*}; envs=$(echo "${spec#*:}" | tr ',' ' ')
  env $envs timeout 300 python bench.py --no-other-configs --no-cpu-baseline 2>"$O/$tag.err" | tail -1 > "$O/$tag.json"
  python - "$O/$tag.json" "$tag" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("%-14s %.4f" % (sys.argv[2], d["ms_per_step"]), {k:round(v,3) for k,v in d.get("kernels_ms",{}).items()})
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
done; done
