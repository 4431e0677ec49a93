#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r02_call5
mkdir -p $out
cd $root
echo "== gpu suite";  timeout 900 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.log 2>&1; tail -4 $out/pytest_gpu.log
echo "== engine gemm"; timeout 300 python tools/gemm_lab/run.py --engine-only 2>&1 | grep -v amdgpu | tee $out/gemm_engine.txt
echo "== bench";       timeout 300 python -u bench.py 2>$out/bench.err | grep "^{" > $out/bench.json; python - $out/bench.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read())
    print("ms/step %.3f  samples/s %.3e  roofline frac %.3f (in-step %.3f, designed %.3f)" % (d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["in_step_event"]["frac"], d["roofline"]["frac_designed_bytes"]))
    print({k: round(v,3) for k,v in d["kernels_ms"].items()}, "gemm TF", round(d["mlp_gemm"]["achieved"],1))
except Exception as e:
    print("bench FAILED", e); print(open(sys.argv[1].replace(".json",".err")).read()[-600:])
PY
echo "== slot_dnn";    timeout 300 python tools/slot_dnn_bench.py 2>$out/slot.err | tail -1 | tee $out/slot_dnn_adam.json | cut -c1-900
echo "== models";      timeout 600 python tools/bench_models.py 2>&1 | grep -v amdgpu | tee $out/models.txt
