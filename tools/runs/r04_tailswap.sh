#!/bin/bash
# which of {sparse update, dW_0 + fold + dense Adam} leaves the main stream after fm_bwd: A/B on the bench step
mkdir -p gpurun_out; O=gpurun_out/tailswap.txt; : > $O
timeout 600 python -m pytest tests/test_deepfm_gpu.py tests/test_deepfm_step_c.py -m gpu -x -q 2>&1 | tail -3 >> $O
for sw in 0 1 0 1 0 1; do
  echo "REC_DEEPFM_TAIL_SWAP=$sw" >> $O
  REC_DEEPFM_TAIL_SWAP=$sw timeout 300 python bench.py --no-other-configs 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('kernels_ms'))" >> $O
done
cat $O
