#!/bin/bash
# one-launch dense fold + layer-0 zero rows inside the flat buffers: parity, then the two batch sizes
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_deepfm_gpu.py tests/test_deepfm_step_c.py tests/test_sharded_gpu.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/fold_tests.txt
for i in 1 2; do
timeout 200 python bench.py --batch 512 --steps 400 --warmup 50 --no-other-configs 2>&1 | tail -1 >> gpurun_out/fold_b512.txt
done
timeout 200 python bench.py --batch 512 --steps 400 --warmup 50 --no-other-configs --c-step 2>&1 | tail -1 >> gpurun_out/fold_b512.txt
for i in 1 2; do
timeout 300 python bench.py --no-other-configs 2>&1 | tail -1 >> gpurun_out/fold_b65536.txt
done
