#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/dlrm; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
cat > /tmp/dlrm_run.py <<PY
import sys, torch
sys.path.insert(0, "$R")
from paddlerec_amd.dlrm import DLRMLayer
DEV="cuda"; g=torch.Generator(device=DEV).manual_seed(3)
B=int(sys.argv[1])
ids=torch.randint(0,1000001,(B,26),device=DEV,generator=g); dense=torch.rand(B,13,device=DEV,generator=g)
label=(torch.rand(B,1,device=DEV,generator=g)<0.25).long()
m=DLRMLayer(13,[512,256,64,16],1000001,16,[512,256,2],26,device=DEV)
for _ in range(5): m.train_step(ids,dense,label,lr=1e-3)
torch.cuda.synchronize()
a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): m.train_step(ids,dense,label,lr=1e-3)
b.record(); torch.cuda.synchronize(); print("DLRM B", B, a.elapsed_time(b)/10, "ms")
PY
for B in 4096 65536; do
python /tmp/dlrm_run.py $B 2>&1 | grep -v amdgpu
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$B -o t -- python /tmp/dlrm_run.py $B > /dev/null 2>&1
f=$(find $O/t$B -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot=0
for r in rows[:24]:
    print(r['Name'].replace('void ','').replace('rec::','')[:86].ljust(86), r['Calls'].rjust(5), "%9.1f us" % (float(r['AverageNs'])/1e3), "%5.1f%%" % float(r['Percentage']))
print("kernels per step ~", sum(int(r['Calls']) for r in rows)/15)
PY
rm -rf $O/t$B
done
