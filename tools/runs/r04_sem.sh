#!/bin/bash
# split-K folded inside the GEMM launch for small outputs: parity, then the launch-bound steps with the switch off / on
mkdir -p gpurun_out; O=gpurun_out/sem.txt; : > $O
timeout 600 python -m pytest tests/test_gemm_gpu.py -m gpu -x -q 2>&1 | tail -3 >> $O
for sem in 0 1 0 1; do
  echo "REC_GEMM_SPLITK_SEM=$sem" >> $O
  REC_GEMM_SPLITK_SEM=$sem timeout 200 python bench.py --batch 512 --steps 400 --warmup 50 --no-other-configs 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('DeepFM B 512', d['ms_per_step'], d['value'])" >> $O
  REC_GEMM_SPLITK_SEM=$sem timeout 200 python tools/din_small_bench.py 2>&1 | grep "DIN train step" >> $O
done
cat $O
