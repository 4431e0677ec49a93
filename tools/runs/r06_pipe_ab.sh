#!/bin/bash
# round 6: the pipelined tail + one-launch weight images: bit-identity tests, then bench A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06pipe; mkdir -p "$O"; cd "$R"
timeout 900 python -m pytest tests/test_deepfm_pipeline_gpu.py -m gpu -x -q 2>&1 | tail -15
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-other-configs --no-cpu-baseline 2>"$O/$tag.err" | tail -1 > "$O/$tag.json"; python - "$O/$tag.json" "$tag" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d["ms_per_step"],4), {k:round(v,3) for k,v in d.get("kernels_ms",{}).items()})
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
run default A=1
run images_off REC_GEMM_IMAGES=0
run pipe REC_DEEPFM_PIPELINED=1
run pipe_noimg REC_DEEPFM_PIPELINED=1 REC_GEMM_IMAGES=0
run default2 A=1
run pipe2 REC_DEEPFM_PIPELINED=1
tail -3 $O/pipe.err
