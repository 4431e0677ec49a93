#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
timeout 1200 python -m pytest $R/tests/test_sharded.py $R/tests/test_slot_dnn.py $R/tests/test_gpubox.py $R/tests/test_ps_gpu.py $R/tests/test_checkpoint.py $R/tests/test_trainer.py -m gpu -x -q 2>&1 | tail -4
for rep in 1 2; do
for f in 0 1; do
REC_CTR_HEAD_FUSED=$f timeout 300 python $R/bench.py --table ps --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('configs[4] share, fused=$f  ms_per_step %.4f  value %.3e' % (d['ms_per_step'], d['value']))"
done; done 2>&1 | tee $O/head_ab2.txt
for f in 0 1; do
REC_CTR_HEAD_FUSED=$f timeout 300 python $R/tools/slot_dnn_bench.py --opt ps 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('gpubox model step, fused=$f  train_step_ms %.3f  pool_fwd_ms %.3f' % (d['train_step_ms'], d['pool_fwd_ms']))"
done 2>&1 | tee -a $O/head_ab2.txt
