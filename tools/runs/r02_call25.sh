#!/bin/bash
# round 2, GPU call 25: where to overlap the id grouping; DIN known-answer tests on the compile-time-shaped kernel; world-1 sharded with the exchange self-test
mkdir -p gpurun_out/r02_call25
o=gpurun_out/r02_call25
timeout 600 python -m pytest tests/test_din_gpu.py -x -q -m gpu > $o/pytest.txt 2>&1; echo "pytest rc=$?" >> $o/pytest.txt; tail -3 $o/pytest.txt
for at in fwd bwd; do
  for rep in 1 2; do
    REC_DEEPFM_GROUP_AT=$at timeout 300 python bench.py --no-cpu-baseline --steps 40 2>/dev/null | grep "^{" > $o/bench_$at$rep.json
  done
done
timeout 300 python bench.py --force-sharded --no-cpu-baseline --steps 30 2>$o/sh.err | grep "^{" > $o/bench_sh.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02_call25/bench_*.json")):
    b = json.loads(open(f).read().strip().splitlines()[0])
    print(f.split("/")[-1], "%.3f ms" % b["ms_per_step"], {k: round(v, 3) for k, v in b["kernels_ms"].items()}, b["config"].get("exchange"))
PY
grep -i "self-test\|FAILED" $o/sh.err | head -3
