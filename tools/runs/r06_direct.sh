#!/bin/bash
# round 6: the launch-bound GEMMs in one launch (csrc/gemm_direct.h, REC_GEMM_DIRECT) — GPU suite, then the reference's own batch
# sizes with and without it (DeepFM bs 512 mirror and C step, DIN bs 32, DCN-v2 bs 512)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06direct; mkdir -p "$O"; cd "$R"
timeout 600 python -m pytest tests/test_gemm_gpu.py -m gpu -x -q 2>&1 | tail -5 | tee "$O/pytest_gemm.txt"
for d in 1 0 1 0; do
  echo "== REC_GEMM_DIRECT=$d"
  for extra in "" "--c-step"; do
    REC_GEMM_DIRECT=$d timeout 300 python bench.py --batch 512 --steps 300 --warmup 30 --no-other-configs --no-cpu-baseline $extra 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('deepfm bs512 $extra', round(d['ms_per_step'],4), 'ms')"
  done
  REC_GEMM_DIRECT=$d timeout 300 python tools/din_small_bench.py 2>&1 | tail -4
  REC_GEMM_DIRECT=$d timeout 300 python tools/dcn_small_bench.py 2>&1 | tail -3
done 2>&1 | tee "$O/ab.txt"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee "$O/pytest_all.txt"
