#!/bin/bash
# round 2, GPU call 20: does the world-1 sharded step slow down with the number of steps?
mkdir -p gpurun_out/r02_call20
o=gpurun_out/r02_call20
for n in 10 30 90; do
  REC_BENCH_STEP_TIMES=1 timeout 300 python bench.py --force-sharded --no-cpu-baseline --steps $n --warmup 5 > $o/sh_$n.json 2> $o/sh_$n.err
  python - <<PY
import json
b=json.loads(open("$o/sh_$n.json").read().strip().splitlines()[-1])
print("steps $n: %.3f ms/step  kernels %s" % (b["ms_per_step"], {k: round(v,3) for k,v in b["kernels_ms"].items()}))
PY
  grep "host ms" $o/sh_$n.err | cut -c1-700
done
REC_BENCH_STEP_TIMES=1 timeout 300 python bench.py --no-cpu-baseline --steps 90 --warmup 5 > $o/plain_90.json 2> $o/plain_90.err
python - <<PY
import json
b=json.loads(open("$o/plain_90.json").read().strip().splitlines()[-1])
print("plain steps 90: %.3f ms/step" % b["ms_per_step"])
PY
grep "host ms" $o/plain_90.err | cut -c1-500
