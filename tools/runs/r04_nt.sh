#!/bin/bash
# non-temporal C stores of the whole-tile GEMM: off / dX + ReLU' form (default) / every GEMM, on the bench step and DCN-v2
mkdir -p gpurun_out; O=gpurun_out/nt_step.txt; : > $O
for nt in 0 1 2 0 1 2; do
  echo "REC_GEMM_NT_STORE=$nt" >> $O
  REC_GEMM_NT_STORE=$nt timeout 300 python bench.py --no-other-configs 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['mlp_gemm_frac'], d.get('kernels_ms'))" >> $O
done
cat $O
