#!/bin/bash
mkdir -p gpurun_out; O=gpurun_out/b512b.txt; : > $O
timeout 600 python -m pytest tests/test_deepfm_gpu.py tests/test_deepfm_step_c.py tests/test_ctr_head_gpu.py tests/test_slot_dnn.py -m gpu -x -q 2>&1 | tail -3 >> $O
for i in 1 2; do
  timeout 200 python bench.py --batch 512 --steps 400 --warmup 50 --no-other-configs 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('DeepFM B 512', d['ms_per_step'], d['value'])" >> $O
done
cat $O
