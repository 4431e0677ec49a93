#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04c5; mkdir -p "$O"; cd "$R"
for i in 1 2; do
  for v in "X=0" "REC_DEEPFM_GROUP_CUS=8" "REC_DEEPFM_GROUP_CUS=4" "REC_DEEPFM_GROUP_CUS=2" "REC_DEEPFM_GROUP_CUS=16" "REC_DEEPFM_GROUP_CUS=0,32" "REC_DEEPFM_GROUP_CUS=0,64"; do
    n=$(echo "$v" | tr ' =,' '___')
    env $v timeout 200 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 > "$O/bench_${n}_$i.json"
  done
done
python - <<'PY'
import json, glob, os
o = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r04c5")
for f in sorted(glob.glob(os.path.join(o, "bench_*.json"))):
    try:
        d = json.loads(open(f).read())
        print(os.path.basename(f), "%.3f ms  %.2f M/s" % (d["ms_per_step"], d["value"] / 1e6), {k: round(v, 3) for k, v in d.get("kernels_ms", {}).items()})
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
PY
