#!/bin/bash
# bench A/B: tiled GEMMs vs the row-panel kernel for the forward (B [K,N]) GEMMs / for every eligible GEMM
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
for rep in 1 2; do
for m in off nn 1; do
  v=$m; [ $m = off ] && v=0
  REC_GEMM_PANEL=$v timeout 200 python $R/bench.py --no-other-configs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('panel=$m  ms_per_step %.4f  value %.3e  mlp_gemm_frac %s' % (d['ms_per_step'], d['value'], r.get('mlp_gemm_frac')))"
done; done 2>&1 | tee $O/panel_ab.txt
echo "--- coarse data (two mantissa bits) and constant data: same instructions"
for d in rand coarse zeros; do echo DATA $d; timeout 100 python $R/tools/gemm_lab/panel_lab.py --rounds 2 --data $d 2>&1 | grep -v amdgpu.ids; done | tee -a $O/panel_ab.txt
