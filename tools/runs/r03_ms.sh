mkdir -p gpurun_out/r03_ms
for o in ps adam; do timeout 300 python tools/slot_dnn_bench.py --opt $o 2>/dev/null | tail -1 > gpurun_out/r03_ms/slot_dnn_$o.json; python -c "
import json; d=json.load(open('gpurun_out/r03_ms/slot_dnn_$o.json')); print('$o', round(d['pool_fwd_ms'],3), round(d['train_step_ms'],3), round(d['samples_per_s']), {k: round(v,2) for k,v in d['kernels_ms'].items()})"; done
