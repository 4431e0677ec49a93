#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/x3; mkdir -p "$O"; cd "$R"
timeout 400 python tools/gemm_lab/bf16x3_lab.py --dw 2>&1 | grep -v amdgpu.ids | tee "$O/call6_dw_lab.txt"
timeout 400 python -m pytest tests/test_gemm_gpu.py -q 2>&1 | tail -8 | tee "$O/test_gemm.txt"
for v in "1 0" "1 1" "1 0" "1 1"; do
  set -- $v
  echo "REC_GEMM_BF16X3=$1 REC_GEMM_BF16X3_DW=$2" | tee -a "$O/ab3.txt"
  REC_GEMM_BF16X3=$1 REC_GEMM_BF16X3_DW=$2 timeout 300 python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('ms_per_step %.4f  value %.3fM  kernels_ms %s  mlp_gemm_tflops %.1f loss %s' % (d['ms_per_step'], d['value']/1e6, {k: round(v,3) for k,v in d['kernels_ms'].items()}, r.get('mlp_gemm_tflops',0), d['config'].get('loss')))" | tee -a "$O/ab3.txt"
done
