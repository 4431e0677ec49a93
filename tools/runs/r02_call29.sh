#!/bin/bash
# round 2, GPU call 29: feasign hash on the device at configs[4] sizes, PS / sharded suites, bench --table ps
mkdir -p gpurun_out/r02_call29
o=gpurun_out/r02_call29
timeout 900 python -m pytest tests/test_slot_dnn.py tests/test_sharded.py tests/test_ps_gpu.py tests/test_gpubox.py -x -q -m gpu > $o/pytest.txt 2>&1; echo "pytest rc=$?" >> $o/pytest.txt; tail -3 $o/pytest.txt
timeout 600 python bench.py --table ps --no-cpu-baseline 2>/dev/null | grep "^{" > $o/bench_ps.json
python - <<'PY'
import json
b = json.loads(open("gpurun_out/r02_call29/bench_ps.json").read().strip().splitlines()[0])
print("ps", "%.3f ms" % b["ms_per_step"], "%.2f M/s" % (b["value"] / 1e6), b["config"]["workload"][:120], b["config"].get("exchange"))
PY
