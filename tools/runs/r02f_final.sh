#!/bin/bash
# round 2, session f: evidence run after the last changes — full GPU suite, smoke, rocprofv3 stats + PMC of bench.py
# (tag r02f), the slot_dnn step on both tables, the bench variants.
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r02f_final
mkdir -p $out
cd $root
timeout 1500 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $out/pytest_gpu.txt; tail -4 $out/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1; tail -2 $out/smoke.txt
timeout 300 python -u bench.py 2>/dev/null | grep "^{" > $out/bench.json
bash tools/profile_bench.sh r02f > $out/profile_bench.log 2>&1; tail -3 $out/profile_bench.log
for o in ps adam; do timeout 300 python tools/slot_dnn_bench.py --opt $o 2>/dev/null | grep "^{" > $out/slot_dnn_$o.json; done
timeout 300 python -u bench.py --force-sharded --no-cpu-baseline 2>/dev/null | grep "^{" > $out/bench_force_sharded.json
timeout 600 python -u bench.py --table ps --no-cpu-baseline 2>/dev/null | grep "^{" > $out/bench_ps.json
timeout 300 python -u bench.py --shared-table --dim 9 --no-cpu-baseline 2>/dev/null | grep "^{" > $out/bench_shared_D9.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02f_final/bench*.json")):
    try:
        b = json.loads(open(f).read().strip().splitlines()[0])
        print(f.split("/")[-1], "%.3f ms" % b["ms_per_step"], "%.2f M/s" % (b["value"] / 1e6), "frac", round(b["roofline"]["frac"], 3))
    except Exception as e:
        print(f, "ERR", e)
for f in sorted(glob.glob("gpurun_out/r02f_final/slot_dnn_*.json")):
    b = json.loads(open(f).read().strip().splitlines()[0])
    print(f.split("/")[-1], "%.2f ms" % b["train_step_ms"], {k: round(v, 2) for k, v in b["kernels_ms"].items()})
PY
