#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03c7; mkdir -p "$O"; cd "$R"
run() { tag=$1; shift; env "$@" timeout 120 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > "$O/bench_$tag.json"; }
for i in 1 2; do
  run base_$i A=1
  run deferall_$i REC_DEEPFM_DEFER_ALL=1
  run dwstream0_$i REC_MLP_DW_STREAM=0
done
python - <<'PY'
import json, glob, os
o = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r03c7")
for f in sorted(glob.glob(os.path.join(o, "bench_*.json"))):
    try:
        d = json.loads(open(f).read()); r = d["roofline"]
        print("%-22s %.3f ms  b2b %.3f  IN-STEP fwd %.1f bwd %.1f frac %.3f" % (os.path.basename(f), d["ms_per_step"], r["frac"], 1e3*r["in_step_event"]["fm_fwd_ms"], 1e3*r["in_step_event"]["fm_bwd_ms"], r["in_step_event"]["frac"]), {k: round(v, 3) for k, v in d["kernels_ms"].items()})
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
PY
