#!/bin/bash
# r04: the default bench line (other_configs, cpu baselines), the L-trainer level re-measured, the 2-rank shared-GPU bench path
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04c7; mkdir -p "$O"; cd "$R"
timeout 900 python bench.py 2> "$O/bench_stderr.txt" | tail -1 > "$O/bench.json"
python - <<'PY'
import json, os
o = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r04c7")
d = json.loads(open(os.path.join(o, "bench.json")).read())
print("main: %.3f ms %.2f M/s roofline %s" % (d["ms_per_step"], d["value"]/1e6, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d["roofline"].items() if not isinstance(v, (dict, str))}))
print("cpu_baseline:", {k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if kk in ("value", "error", "cores")}) for k, v in d.get("cpu_baseline", {}).items() if k != "sample"})
for e in d.get("other_configs", []):
    print("  ", e.get("command"), "|", e.get("config"), "|", {k: (round(e[k], 3) if isinstance(e[k], float) else e[k]) for k in ("ms", "ms_per_step", "pool_fwd_ms", "train_step_ms", "error", "wall_s") if k in e}, (e.get("roofline") or {}).get("bound"), round((e.get("roofline") or {}).get("frac") or 0, 3))
PY
timeout 600 python tools/trainer_bench.py --lines 13107200 --files 4 --epochs 2 --tables --rows 1000000 2> "$O/trainer_stderr.txt" | tail -1 | tee "$O/trainer_bench.json"
REC_BENCH_SHARE_GPU=1 REC_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --hashed-rows 20000000 --no-cpu-baseline 2> "$O/bench2_stderr.txt" | tail -1 > "$O/bench_2ranks_shared_gpu.json"
python - <<'PY'
import json, os
o = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r04c7")
try:
    d = json.loads(open(os.path.join(o, "bench_2ranks_shared_gpu.json")).read())
    print("2 ranks on one GPU (gloo): %.2f ms" % d["ms_per_step"], json.dumps(d.get("exchange"))[:900])
except Exception as e:
    print("2-rank line unreadable:", e, open(os.path.join(o, "bench2_stderr.txt")).read()[-1500:])
PY
tail -5 "$O/bench_stderr.txt"
