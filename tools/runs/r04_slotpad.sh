#!/bin/bash
# padded pool stride in the gpubox model: parity, then the model step with the switch off / on
mkdir -p gpurun_out; O=gpurun_out/slotpad.txt; : > $O
timeout 900 python -m pytest tests/test_slot_dnn.py tests/test_gpubox.py tests/test_sharded_slot_dnn.py tests/test_ps_gpu.py -m gpu -x -q 2>&1 | tail -4 >> $O
for pad in 0 1 0 1; do
  echo "REC_SLOT_PAD0=$pad" >> $O
  REC_SLOT_PAD0=$pad timeout 300 python tools/slot_dnn_bench.py --opt ps 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('train_step_ms','pool_fwd_ms','kernels_ms','mlp_tflops_in_step')})" >> $O
done
cat $O
