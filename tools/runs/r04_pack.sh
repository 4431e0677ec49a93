#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest $R/tests/test_deepfm_gpu.py $R/tests/test_group_slots_gpu.py $R/tests/test_row_update_shapes_gpu.py $R/tests/test_slot_dnn.py $R/tests/test_gpubox.py $R/tests/test_ps_gpu.py $R/tests/test_sharded.py $R/tests/test_din_gpu.py $R/tests/test_dcn_v2_gpu.py -m gpu -x -q 2>&1 | tail -2
for pk in 0 1; do echo "REC_RSORT_PACK=$pk"; REC_RSORT_PACK=$pk timeout 200 python $R/tools/gpubox_sort_probe.py 2>&1 | grep -v amdgpu | tail -1; done | tee $O/pack.txt
run() { timeout 200 python $R/bench.py --no-cpu-baseline --no-other-configs "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-40s ms_per_step %.4f  value %.3e' % ('$LABEL', d['ms_per_step'], d['value']))"; }
for rep in 1 2; do for pk in 0 1; do
  LABEL="shared table D 10, pack $pk"; REC_RSORT_PACK=$pk run --shared-table --dim 10
done; done 2>&1 | tee -a $O/pack.txt
for rep in 1 2; do for pk in 0 1; do REC_RSORT_PACK=$pk timeout 300 python $R/tools/slot_dnn_bench.py --opt ps 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('gpubox model pack $pk train_step_ms %.3f' % d['train_step_ms'])"; done; done | tee -a $O/pack.txt
