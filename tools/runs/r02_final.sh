#!/bin/bash
# round 2, evidence run after the last kernel changes: full GPU suite, smoke, rocprofv3 stats + PMC of bench.py (tag r02c),
# model / slot_dnn kernel stats, the bench variants (configs[4] PS table, world-1 sharded, shared table)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r02_final
mkdir -p $out
cd $root
timeout 1500 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $out/pytest_gpu.txt; tail -4 $out/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1; tail -2 $out/smoke.txt
bash tools/profile_bench.sh r02c > $out/profile_bench.log 2>&1; tail -5 $out/profile_bench.log
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $out/models -o m --output-format csv -- python $root/tools/bench_models.py > $out/models.txt 2>&1; echo "models trace rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats -d $out/slot -o s --output-format csv -- python $root/tools/slot_dnn_bench.py > $out/slot_dnn.txt 2>&1; echo "slot trace rc=$?"
cd $root
find $out -name "*kernel_trace.csv" -delete
timeout 300 python -u bench.py 2>/dev/null | grep "^{" > $out/bench.json
timeout 300 python -u bench.py --force-sharded --no-cpu-baseline 2>/dev/null | grep "^{" > $out/bench_force_sharded.json
timeout 600 python -u bench.py --table ps --no-cpu-baseline 2>/dev/null | grep "^{" > $out/bench_ps.json
timeout 300 python -u bench.py --shared-table --dim 9 --no-cpu-baseline 2>/dev/null | grep "^{" > $out/bench_shared_D9.json
timeout 300 python -u bench.py --ids zipf --no-cpu-baseline 2>/dev/null | grep "^{" > $out/bench_zipf.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02_final/bench*.json")):
    try:
        b = json.loads(open(f).read().strip().splitlines()[0])
        print(f.split("/")[-1], "%.3f ms" % b["ms_per_step"], "%.2f M/s" % (b["value"] / 1e6), "frac", round(b["roofline"]["frac"], 3))
    except Exception as e:
        print(f, "ERR", e)
PY
