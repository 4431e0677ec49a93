#!/bin/bash
# kernel stats of the sibling nets' train steps at B 65536 (wide_deep, dnn, fm)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/strace; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
cat > /tmp/s_run.py <<PY
import sys, torch
sys.path.insert(0, "$R")
DEV="cuda"; g=torch.Generator(device=DEV).manual_seed(3)
which, B = sys.argv[1], int(sys.argv[2])
N = 1000001
ids = torch.randint(0, N, (B, 26), device=DEV, generator=g); dense = torch.rand(B, 13, device=DEV, generator=g)
label = (torch.rand(B, 1, device=DEV, generator=g) < 0.25).long()
if which == "wd":
    from paddlerec_amd.wide_deep import WideDeepLayer
    m = WideDeepLayer(N, 9, 13, 26, [512, 256, 256, 128], device=DEV)
elif which == "dnn":
    from paddlerec_amd.dnn import DNNLayer
    m = DNNLayer(N, 9, 13, 26, [512, 256, 128, 32], device=DEV)
else:
    from paddlerec_amd.fm import FMLayer
    m = FMLayer(N, 9, 13, 26, device=DEV)
fn = lambda: m.train_step(ids, dense, label, lr=1e-3)
for _ in range(5): fn()
torch.cuda.synchronize()
a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): fn()
b.record(); torch.cuda.synchronize(); print(which, "B", B, "%.3f ms" % (a.elapsed_time(b)/10))
PY
for spec in "wd 65536" "dnn 65536" "fm 65536"; do
set -- $spec
python /tmp/s_run.py $1 $2 2>&1 | grep -v amdgpu | tail -3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$1 -o t -- python /tmp/s_run.py $1 $2 > /dev/null 2>&1
f=$(find $O/t$1 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'at::' not in r['Name'] and 'erfinv' not in r['Name'] and 'spin' not in r['Name']]
for r in rows[:12]:
    print("  ", r['Name'].replace('void ','').replace('rec::','').replace('(anonymous namespace)::','')[:84].ljust(84), r['Calls'].rjust(5), "%9.1f us" % (float(r['AverageNs'])/1e3), "%5.1f%%" % float(r['Percentage']))
PY
rm -rf $O/t$1
done
