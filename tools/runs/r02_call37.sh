#!/bin/bash
# round 2, GPU call 37: full GPU suite + smoke + bench on the current tree
mkdir -p gpurun_out/r02_call37
o=gpurun_out/r02_call37
timeout 1500 python -m pytest tests -x -q -m gpu > $o/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $o/pytest_gpu.txt; tail -3 $o/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $o/smoke.txt 2>&1; tail -1 $o/smoke.txt
timeout 600 python bench.py 2>/dev/null | grep "^{" > $o/bench.json
python - <<'PY'
import json
b = json.loads(open("gpurun_out/r02_call37/bench.json").read().strip().splitlines()[0])
print("bench %.3f ms  %.2f M/s  frac %.3f  cpu %s" % (b["ms_per_step"], b["value"] / 1e6, b["roofline"]["frac"], b["cpu_baseline"]["value"]))
PY
