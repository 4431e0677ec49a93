#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
for rep in 1 2 3; do for f in 0 1; do
REC_CTR_HEAD_FUSED=$f timeout 300 python $R/tools/slot_dnn_bench.py --opt ps 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('gpubox model step, fused=$f  train_step_ms %.3f  pool_fwd_ms %.3f  kernels %s' % (d['train_step_ms'], d['pool_fwd_ms'], {k: round(v,3) for k,v in d['kernels_ms'].items()}))"
done; done 2>&1 | tee $O/head_ab3.txt
