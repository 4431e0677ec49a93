#!/bin/bash
# fingerprint scan + vector staging of the one-launch merge: parity, then DIN bs 32 and DeepFM bs 512 with the switch off / on
mkdir -p gpurun_out; O=gpurun_out/small_fp.txt; : > $O
timeout 900 python -m pytest tests/test_din_gpu.py tests/test_row_update_shapes_gpu.py tests/test_deepfm_gpu.py tests/test_deepfm_step_c.py -m gpu -x -q 2>&1 | tail -4 >> $O
for fp in 0 1 0 1; do
  echo "REC_SMALL_FP=$fp" >> $O
  REC_SMALL_FP=$fp timeout 200 python tools/din_small_bench.py 2>&1 | grep "DIN train step" >> $O
done
for fp in 0 1; do
  echo "REC_SMALL_FP=$fp  (DeepFM B 512: slot-local merge, unaffected by design)" >> $O
  REC_SMALL_FP=$fp timeout 200 python bench.py --batch 512 --steps 400 --warmup 50 --no-other-configs 2>&1 | tail -1 | cut -c1-330 >> $O
done
cat $O
