#!/bin/bash
# multislot pool tile-shape experiment: REC_MS_CFG 0 = 8 waves x 2 slots (shipped), 1 = 4x2, 2 = 4x4, 3 = 8x1, 4 = 8x4
for c in 0 1 2 3 4 0; do
  echo "== REC_MS_CFG=$c"
  REC_MS_CFG=$c python tools/slot_dnn_bench.py --pool-bench 1 2>&1 | grep -v amdgpu.ids
done
