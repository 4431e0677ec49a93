#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest $R/tests/test_din_gpu.py $R/tests/test_models_fullsize_gpu.py -m gpu -x -q 2>&1 | tail -2
for sp in 0 1; do echo "REC_DIN_TILE_SPLIT=$sp"; REC_DIN_TILE_SPLIT=$sp timeout 200 python $R/tools/din_small_bench.py 2>&1 | grep -v amdgpu | tail -3; done | tee $O/din_split.txt
for sp in 0 1; do echo "REC_DIN_TILE_SPLIT=$sp"; REC_DIN_TILE_SPLIT=$sp timeout 300 python $R/tools/bench_models.py --only din 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('  ', d.get('workload','')[:90], d.get('ms'), d.get('ms_per_step'))"; done | tee -a $O/din_split.txt
