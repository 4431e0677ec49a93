#!/bin/bash
# round 2, GPU call 31: dW GEMMs of the MLP backward on a third stream (tail filling)
mkdir -p gpurun_out/r02_call31
o=gpurun_out/r02_call31
run() { env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 40 2>/dev/null | grep "^{" ; }
run REC_MLP_DW_STREAM=0 > $o/a_off.json
run REC_MLP_DW_STREAM=1 > $o/b_on.json
run REC_MLP_DW_STREAM=0 > $o/c_off.json
run REC_MLP_DW_STREAM=1 > $o/d_on.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02_call31/*.json")):
    b = json.loads(open(f).read().strip().splitlines()[0])
    print(f.split("/")[-1], "%.3f ms" % b["ms_per_step"], {k: round(v, 3) for k, v in b["kernels_ms"].items()}, "loss", b["config"]["loss"])
PY
