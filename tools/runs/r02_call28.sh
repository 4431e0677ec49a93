#!/bin/bash
# round 2, GPU call 28: bench.py as the driver runs it (no flags), first thing on a fresh box, twice; and with --warmup 2 --steps 5
mkdir -p gpurun_out/r02_call28
o=gpurun_out/r02_call28
timeout 600 python bench.py 2>/dev/null | grep "^{" > $o/bench1.json
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | grep "^{" > $o/bench2.json
timeout 600 python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | grep "^{" > $o/bench3.json
timeout 600 python bench.py --force-sharded --no-cpu-baseline 2>/dev/null | grep "^{" > $o/bench4_sh.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02_call28/*.json")):
    b = json.loads(open(f).read().strip().splitlines()[0])
    print(f.split("/")[-1], "%.3f ms" % b["ms_per_step"], "%.2f M/s" % (b["value"] / 1e6), "frac %.3f" % b["roofline"]["frac"], {k: round(v, 3) for k, v in b["kernels_ms"].items()})
PY
