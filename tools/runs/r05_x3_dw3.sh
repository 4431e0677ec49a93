#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/x3; mkdir -p "$O"; cd "$R"
timeout 600 python tools/gemm_lab/bf16x3_lab.py --dw 2>&1 | grep -v amdgpu.ids | tee "$O/call9_dw_lab.txt"
timeout 600 python -m pytest tests/test_gemm_gpu.py -q 2>&1 | tail -4 | tee "$O/test_gemm.txt"
for v in 1 2 1 2; do
  echo "REC_GEMM_BF16X3=$v" | tee -a "$O/ab5.txt"
  REC_GEMM_BF16X3=$v timeout 300 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('ms_per_step %.4f  value %.3fM  kernels_ms %s  mlp_gemm_tflops %.1f loss %s' % (d['ms_per_step'], d['value']/1e6, {k: round(v,3) for k,v in d['kernels_ms'].items()}, r.get('mlp_gemm_tflops',0), d['config'].get('loss')))" | tee -a "$O/ab5.txt"
done
