#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/dcn; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
cat > /tmp/d_run.py <<PY
import sys, torch
sys.path.insert(0, "$R")
DEV="cuda"; g=torch.Generator(device=DEV).manual_seed(3)
B = int(sys.argv[1]); mix = sys.argv[2] == "mix"
from paddlerec_amd.dcn_v2 import DCN_V2Layer
m = DCN_V2Layer(1100001, 40, 13, 26, [768, 768], 2 if mix else 3, is_Stacked=True, use_low_rank_mixture=mix, low_rank=256, num_experts=4, device=DEV)
ids = torch.randint(1, 1100001, (B, 26), device=DEV, generator=g); dense = torch.rand(B, 13, device=DEV, generator=g)
label = (torch.rand(B, 1, device=DEV, generator=g) < 0.25).long()
fn = lambda: m.train_step(ids, dense, label, lr=1e-3)
for _ in range(3): fn()
torch.cuda.synchronize()
a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5): fn()
b.record(); torch.cuda.synchronize(); print("dcn", sys.argv[2], "B", B, "%.3f ms" % (a.elapsed_time(b)/5))
PY
for spec in "65536 v2" "65536 mix"; do
set -- $spec
python /tmp/d_run.py $1 $2 2>&1 | grep -v amdgpu | tail -1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$2 -o t -- python /tmp/d_run.py $1 $2 > /dev/null 2>&1
f=$(find $O/t$2 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'at::' not in r['Name'] and 'erfinv' not in r['Name'] and 'spin' not in r['Name']]
for r in rows[:18]:
    print("  ", r['Name'].replace('void ','').replace('rec::','').replace('(anonymous namespace)::','')[:84].ljust(84), r['Calls'].rjust(5), "%9.1f us" % (float(r['AverageNs'])/1e3), "%5.1f%%" % float(r['Percentage']))
PY
rm -rf $O/t$2
done
