#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r02_call11
mkdir -p $out
cd $root
echo "== gpu suite";  timeout 900 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.log 2>&1; tail -4 $out/pytest_gpu.log
echo "== models";      timeout 600 python tools/bench_models.py 2>&1 | grep -v amdgpu | tee $out/models.txt | grep -v "^{"
echo "== shared table D=9 / D=10 (layout 2b)"
for d in 9 10; do timeout 300 python -u bench.py --no-cpu-baseline --shared-table --rows-per-table 1000001 --dim $d 2>$out/b$d.err | grep "^{" > $out/bench_shared_D$d.json; python - $out/bench_shared_D$d.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read()); r=d["roofline"]
    print("ms/step %.3f  samples/s %.3e  fm_fwd %.1f us fm_bwd %.1f us  frac %.3f" % (d["ms_per_step"], d["value"], r["per_kernel"]["fm_fwd"]["us"], r["per_kernel"]["fm_bwd"]["us"], r["frac"]))
except Exception as e:
    print("FAILED", e); print(open(sys.argv[1].replace("bench_shared_D","b").replace(".json",".err")).read()[-800:])
PY
done
echo "== sharded tail modes"
for mode in overlap serial; do REC_SHARD_TAIL=$mode timeout 120 python -u bench.py --steps 30 --warmup 5 --force-sharded --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$mode', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernels_ms'].items()})"; done
