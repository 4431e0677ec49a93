#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r02_call7
mkdir -p $out
cd $root
summ() { python - $1 <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read())
    print("ms/step %.3f  samples/s %.3e  roofline frac %.3f (in-step %.3f)" % (d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["in_step_event"]["frac"]))
    print({k: round(v,3) for k,v in d["kernels_ms"].items()}, "gemm TF", round(d["mlp_gemm"]["achieved"],1))
except Exception as e:
    print("bench FAILED", e)
PY
}
echo "== bench overlap";   timeout 300 python -u bench.py --no-cpu-baseline 2>$out/bench.err | grep "^{" > $out/bench_overlap.json; summ $out/bench_overlap.json
echo "== bench serial";    REC_DEEPFM_OVERLAP=0 timeout 300 python -u bench.py --no-cpu-baseline 2>$out/bench2.err | grep "^{" > $out/bench_serial.json; summ $out/bench_serial.json
echo "== gpu suite";  timeout 900 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
echo "== models";      timeout 600 python tools/bench_models.py 2>&1 | grep -v amdgpu | tee $out/models.txt
