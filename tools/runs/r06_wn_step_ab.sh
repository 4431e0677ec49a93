#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06wn; mkdir -p "$O"; cd "$R"
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-other-configs --no-cpu-baseline 2>"$O/$tag.err" | tail -1 > "$O/$tag.json"; python - "$O/$tag.json" "$tag" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d["ms_per_step"],4), {k:round(v,3) for k,v in d.get("kernels_ms",{}).items()})
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
for i in 1 2 3; do
run wn2_$i REC_X3_WN=2
run wn1_$i REC_X3_WN=1
run wn1_pipe_$i REC_X3_WN=1 REC_DEEPFM_PIPELINED=1
done
