#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04c4; mkdir -p "$O"; cd "$R"
for i in 1 2; do
  for v in "REC_FM_BWD_RG_NT=0" "REC_FM_BWD_RG_NT=1" "REC_DEEPFM_SORTED=0" "REC_DEEPFM_SKIP_GROUP=1"; do
    n=$(echo "$v" | tr ' =' '__')
    env $v timeout 200 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 > "$O/bench_${n}_$i.json"
  done
done
python - <<'PY'
import json, glob, os
o = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r04c4")
for f in sorted(glob.glob(os.path.join(o, "bench_*.json"))):
    try:
        d = json.loads(open(f).read())
        print(os.path.basename(f), "%.3f ms  %.2f M/s" % (d["ms_per_step"], d["value"] / 1e6), {k: round(v, 3) for k, v in d.get("kernels_ms", {}).items()})
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -o t -- python "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > "$O/bench_under_rocprof.log" 2>&1
f=$(find "$O/trace" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$O/kernel_stats.csv"
python - "$O/kernel_stats.csv" <<'PY'
import csv, sys
tot = 0
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if 'rec::' in n:
        short = n.split('(')[0].replace('void ', '')[:64]
        print("%-66s calls %4s avg %8.1f min %8.1f max %8.1f" % (short, r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
rm -rf "$O/trace"
