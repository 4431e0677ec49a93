#!/bin/bash
# First GPU call of a round: everything whose numbers were left open at the end of the previous one, in order of
# value per GPU-minute, each under its own timeout, all output under gpurun_out/round_start/.
#   gpurun --timeout 900 -- 'bash tools/round_start.sh'
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/round_start
mkdir -p $out
cd $root
echo "== gpu tests";      timeout 300 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
echo "== bench";          timeout 200 python -u bench.py 2>/dev/null | grep "^{" > $out/bench.json; cut -c1-200 $out/bench.json
echo "== sharded path, world 1 (12 and 30 steps, tail modes)"
for steps in 12 30; do
  for mode in overlap serial; do
    REC_SHARD_TAIL=$mode timeout 120 python -u bench.py --steps $steps --warmup 3 --force-sharded --no-cpu-baseline 2>/dev/null \
      | grep "^{" > $out/sharded_${mode}_$steps.json
    python - "$out/sharded_${mode}_$steps.json" "$mode" "$steps" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
    print(sys.argv[2], sys.argv[3], "steps:", round(d["ms_per_step"], 3), "ms", {k: round(v, 3) for k, v in d["kernels_ms"].items()})
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
  done
done
echo "== sharded timeline (kernel trace)"
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace -d $out/trace -o t --output-format csv -- \
    python $root/bench.py --steps 8 --warmup 3 --force-sharded --no-cpu-baseline > $out/trace.log 2>&1 )
f=$(ls $out/trace/*kernel_trace.csv $out/trace/*/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$f" ] && python tools/trace_timeline.py $f > $out/sharded_timeline.txt && head -2 $out/sharded_timeline.txt
echo "== zipf ids";       timeout 200 python tools/zipf_bench.py 2>&1 | grep -v amdgpu | tail -3 | tee $out/zipf.txt
echo "== trainer ips";    timeout 300 python tools/trainer_bench.py --lines 1048576 2>/dev/null | tail -1 | tee $out/trainer_bench.json | cut -c1-300
echo "== host parser";    timeout 200 python tools/reader_bench.py 2>&1 | tail -6 | tee $out/reader_bench.txt
