#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest $R/tests/test_gemm_gpu.py $R/tests/test_deepfm_step_c.py $R/tests/test_deepfm_gpu.py $R/tests/test_dcn_v2_gpu.py $R/tests/test_din_gpu.py -m gpu -x -q 2>&1 | tail -3
run() { timeout 200 python $R/bench.py --no-cpu-baseline --no-other-configs "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-30s ms_per_step %.4f  value %.3e' % ('$LABEL', d['ms_per_step'], d['value']))"; }
for rep in 1 2; do
  LABEL="B65536"; run
  LABEL="B512 planned"; run --batch 512 --steps 400 --warmup 40
  LABEL="B512 C step"; run --batch 512 --steps 400 --warmup 40 --c-step
done 2>&1 | tee $O/reduce_fold.txt
