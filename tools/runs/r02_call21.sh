#!/bin/bash
# round 2, GPU call 21: MLP chain alone vs in the step
mkdir -p gpurun_out/r02_call21
o=gpurun_out/r02_call21
timeout 300 python tools/mlp_chain_bench.py > $o/mlp_chain.txt 2>&1; cat $o/mlp_chain.txt
REC_DEEPFM_OVERLAP=0 timeout 300 python bench.py --no-cpu-baseline --steps 30 > $o/bench_serial.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --steps 30 > $o/bench.json 2>/dev/null
python - <<'PY'
import json
for n in ("bench_serial", "bench"):
    b = json.loads(open("gpurun_out/r02_call21/%s.json" % n).read().strip().splitlines()[0])
    print(n, b["ms_per_step"], {k: round(v, 3) for k, v in b["kernels_ms"].items()})
PY
