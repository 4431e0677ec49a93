#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06dedup; mkdir -p "$O"; cd "$R"
timeout 900 python -m pytest tests/test_sharded.py -m gpu -x -q 2>&1 | tail -3
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --force-sharded --table ps --hashed-rows 1250000000 --no-cpu-baseline --steps 20 --warmup 5 2>"$O/$tag.err" | tail -1 > "$O/$tag.json"; python - "$O/$tag.json" "$tag" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("%-28s %.4f ms" % (sys.argv[2], d["ms_per_step"]), {k:round(v,3) for k,v in d.get("kernels_ms",{}).items()})
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
for i in 1 2; do
run nolinks A=1
run nolinks_dedup REC_SHARD_DEDUP=1
run links REC_EMULATE_LINKS=8
run links_dedup REC_EMULATE_LINKS=8 REC_SHARD_DEDUP=1
done
tail -3 $O/nolinks_dedup.err
