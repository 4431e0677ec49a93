#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
for dg in 10 11; do REC_RSORT_DIGIT=$dg timeout 600 python -m pytest $R/tests/test_deepfm_gpu.py $R/tests/test_group_slots_gpu.py $R/tests/test_row_update_shapes_gpu.py -m gpu -x -q 2>&1 | tail -1; done
run() { timeout 200 python $R/bench.py --no-cpu-baseline --no-other-configs "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-40s ms_per_step %.4f  value %.3e' % ('$LABEL', d['ms_per_step'], d['value']))"; }
for rep in 1 2; do for dg in 9 10 11; do
  LABEL="shared table D 10, digit $dg"; REC_RSORT_DIGIT=$dg run --shared-table --dim 10
done; done 2>&1 | tee $O/digit.txt
for dg in 9 10; do LABEL="gpubox model... slot_dnn digit $dg"; REC_RSORT_DIGIT=$dg timeout 300 python $R/tools/slot_dnn_bench.py --opt ps 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('slot_dnn digit $dg train_step_ms %.3f' % d['train_step_ms'])"; done | tee -a $O/digit.txt
