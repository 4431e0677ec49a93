#!/bin/bash
# round 6: schedule A/B on the bench step: default vs every dW GEMM in the tail (REC_DEEPFM_DEFER_ALL=1), with and without bf16x3 dW
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06sched; mkdir -p "$O"; cd "$R"
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-other-configs --no-cpu-baseline 2>/dev/null | tail -1 > "$O/$tag.json"; python - "$O/$tag.json" "$tag" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d["ms_per_step"],4), {k:round(v,3) for k,v in d.get("kernels_ms",{}).items()})
PY
}
run default A=1
run defer_all REC_DEEPFM_DEFER_ALL=1
run default2 A=1
run defer_all_dw0 REC_DEEPFM_DEFER_ALL=1 REC_GEMM_BF16X3_DW=0
run dw0 REC_GEMM_BF16X3_DW=0
run x3_1 REC_GEMM_BF16X3=1
run defer_all_x3_1 REC_DEEPFM_DEFER_ALL=1 REC_GEMM_BF16X3=1
