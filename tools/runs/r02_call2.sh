#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r02_call2
mkdir -p $out
cd $root
echo "== slot_dnn tests";  timeout 600 python -m pytest tests/test_slot_dnn.py tests/test_deepfm_gpu.py tests/test_ps_gpu.py -m gpu -q -x > $out/pytest.log 2>&1; tail -4 $out/pytest.log
echo "== bench";       timeout 300 python -u bench.py 2>$out/bench.err | grep "^{" > $out/bench.json; python - $out/bench.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read())
    print("ms/step %.3f  samples/s %.3e  roofline frac %.3f (in-step %.3f, designed %.3f)" % (d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["in_step_event"]["frac"], d["roofline"]["frac_designed_bytes"]))
    print({k: round(v,3) for k,v in d["kernels_ms"].items()}, "gemm TF", round(d["mlp_gemm"]["achieved"],1))
    print(d["roofline"]["per_kernel"])
except Exception as e:
    print("bench FAILED", e); print(open(sys.argv[1].replace(".json",".err")).read()[-600:])
PY
echo "== slot_dnn";    timeout 300 python tools/slot_dnn_bench.py 2>$out/slot.err | tail -1 | tee $out/slot_dnn_adam.json | cut -c1-900
echo "== gemm lab";    timeout 900 python tools/gemm_lab/run.py 2>&1 | grep -v amdgpu | tee $out/gemm_lab.txt
