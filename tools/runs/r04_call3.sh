#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04c3; mkdir -p "$O"; cd "$R"
timeout 600 python -m pytest tests/test_gemm_gpu.py -x -q 2>&1 | tail -5 | tee "$O/pytest_gemm.txt"
python tools/dw0_bench.py 2>&1 | grep -v amdgpu.ids | tee "$O/dw0_bench.txt"
REC_GEMM_144=0 python tools/dw0_bench.py 2>&1 | grep -v amdgpu.ids | tee -a "$O/dw0_bench.txt"
for i in 1 2; do
  for v in "REC_GEMM_144=0" "REC_GEMM_144=1" "REC_GEMM_144=1 REC_DW0_SPLIT=32" "REC_GEMM_144=1 REC_DW0_SPLIT=8"; do
    n=$(echo "$v" | tr ' =' '__')
    env $v timeout 200 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 > "$O/bench_${n}_$i.json"
  done
done
python - <<'PY'
import json, glob, os
o = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r04c3")
for f in sorted(glob.glob(os.path.join(o, "bench_*.json"))):
    try:
        d = json.loads(open(f).read())
        print(os.path.basename(f), "%.3f ms  %.2f M/s" % (d["ms_per_step"], d["value"] / 1e6), {k: round(v, 3) for k, v in d.get("kernels_ms", {}).items()})
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e)
PY
