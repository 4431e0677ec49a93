#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest $R/tests/test_deepfm_step_c.py -m gpu -x -q 2>&1 | tail -4
run() { timeout 200 python $R/bench.py --no-cpu-baseline --no-other-configs "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-44s ms_per_step %.4f  value %.3e' % ('$LABEL', d['ms_per_step'], d['value']))"; }
for rep in 1 2; do
  LABEL="B65536 python mirror"; run
  LABEL="B65536 rec_deepfm_train_step + side stream"; run --c-step
  LABEL="B65536 rec_deepfm_train_step, one stream"; REC_DEEPFM_OVERLAP=0 run --c-step
done 2>&1 | tee $O/cstep_b64k.txt
