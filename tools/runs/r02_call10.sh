#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r02_call10
mkdir -p $out
cd $root
summ() { python - $1 <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read())
    print("ms/step %.3f  samples/s %.3e  roofline frac %.3f (in-step %.3f) rows_total %s" % (d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["in_step_event"]["frac"], d["config"]["table_rows_total"]))
    print({k: round(v,3) for k,v in d["kernels_ms"].items()}, "gemm TF", round(d["mlp_gemm"]["achieved"],1), "oob", d["config"]["index_oob_flag"])
except Exception as e:
    print("bench FAILED", e); print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
}
echo "== gpu suite";  timeout 900 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.log 2>&1; tail -4 $out/pytest_gpu.log
echo "== bench";   timeout 300 python -u bench.py --no-cpu-baseline 2>$out/bench.err | grep "^{" > $out/bench.json; summ $out/bench.json
echo "== bench force-sharded (adam, native exchange)";   timeout 300 python -u bench.py --no-cpu-baseline --force-sharded 2>$out/bench_sh.err | grep "^{" > $out/bench_sh.json; summ $out/bench_sh.json
echo "== bench configs[4]: PS table 1.25e9 rows";   timeout 600 python -u bench.py --no-cpu-baseline --table ps 2>$out/bench_ps.err | grep "^{" > $out/bench_ps.json; summ $out/bench_ps.json
