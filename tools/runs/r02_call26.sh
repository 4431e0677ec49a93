#!/bin/bash
# round 2, GPU call 26: all dW GEMMs in the tail (defer_all) with the grouping under the dX chain or in the tail
mkdir -p gpurun_out/r02_call26
o=gpurun_out/r02_call26
run() { env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 40 2>/dev/null | grep "^{" ; }
run REC_DEEPFM_GROUP_AT=bwd > $o/a_bwd.json
run REC_DEEPFM_GROUP_AT=bwd REC_DEEPFM_DEFER_ALL=1 > $o/b_bwd_deferall.json
run REC_DEEPFM_GROUP_AT=tail REC_DEEPFM_DEFER_ALL=1 > $o/c_tail_deferall.json
run REC_DEEPFM_GROUP_AT=tail > $o/d_tail.json
run REC_DEEPFM_GROUP_AT=bwd > $o/e_bwd_again.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02_call26/*.json")):
    b = json.loads(open(f).read().strip().splitlines()[0])
    print(f.split("/")[-1], "%.3f ms" % b["ms_per_step"], {k: round(v, 3) for k, v in b["kernels_ms"].items()})
PY
