#!/bin/bash
# round 2, GPU call 23: split-K reduce (wave per quarter of the splits, column sums folded in) — parity, chain timing, bench
mkdir -p gpurun_out/r02_call22
o=gpurun_out/r02_call22
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_dcn_v2_gpu.py tests/test_din_gpu.py -x -q -m gpu > $o/pytest.txt 2>&1; echo "pytest rc=$?" >> $o/pytest.txt
tail -4 $o/pytest.txt
timeout 300 python tools/mlp_chain_bench.py > $o/mlp_chain.txt 2>&1; cat $o/mlp_chain.txt
timeout 300 python bench.py --no-cpu-baseline --steps 30 > $o/bench.json 2>/dev/null
python - <<'PY'
import json
b = json.loads(open("gpurun_out/r02_call22/bench.json").read().strip().splitlines()[0])
print("bench", b["ms_per_step"], b["value"], {k: round(v, 3) for k, v in b["kernels_ms"].items()}, b["roofline"]["frac"])
PY
