#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r02_call9
mkdir -p $out
cd $root
echo "== group tests";  timeout 900 python -m pytest tests/test_deepfm_gpu.py tests/test_sharded.py -m gpu -q -x -k "group or route or shard or edge" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
echo "== group bench"; timeout 300 python tools/group_bench.py 2>&1 | grep -v amdgpu | tee $out/group_bench.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace -o g --output-format csv -- python $root/tools/group_bench.py > $out/trace.log 2>&1
python - $out/trace/g_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    n=r["Name"].replace("void ","").replace("rec::","")
    n=n[:n.index("(")] if "(" in n else n
    print("%-90s calls %4s avg %9.1f us" % (n[:90], r["Calls"], float(r["AverageNs"])/1e3))
PY
