#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest $R/tests/test_deepfm_gpu.py $R/tests/test_deepfm_step_c.py $R/tests/test_trainer.py $R/tests/test_checkpoint.py $R/tests/test_reference_entrypoint.py $R/tests/test_compat_gpu.py $R/tests/test_xdeepfm.py $R/tests/test_autograd.py -m gpu -x -q 2>&1 | tail -5
run() { timeout 200 python $R/bench.py --no-cpu-baseline --no-other-configs "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-40s ms_per_step %.4f  value %.3e loss %.5f' % ('$LABEL', d['ms_per_step'], d['value'], d['config']['loss']))"; }
for D in 9 10; do
  LABEL="shared table D $D, dense feat (PAD0=0)"; REC_DEEPFM_PAD0=0 run --shared-table --dim $D
  LABEL="shared table D $D, padded layer-0 input"; run --shared-table --dim $D
done 2>&1 | tee $O/pad0.txt
