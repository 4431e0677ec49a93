#!/bin/bash
# kernel stats of the secondary models' train steps (one model per process)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/mtrace; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
cat > /tmp/m_run.py <<PY
import sys, torch
sys.path.insert(0, "$R")
DEV="cuda"; g=torch.Generator(device=DEV).manual_seed(3)
which, B = sys.argv[1], int(sys.argv[2])
if which == "dcn":
    from paddlerec_amd.dcn_v2 import DCN_V2Layer
    m = DCN_V2Layer(1100001, 40, 13, 26, [768, 768], 3, is_Stacked=True, use_low_rank_mixture=False, low_rank=256, num_experts=4, device=DEV)
    ids = torch.randint(1, 1100001, (B, 26), device=DEV, generator=g); dense = torch.rand(B, 13, device=DEV, generator=g)
    label = (torch.rand(B, 1, device=DEV, generator=g) < 0.25).long()
    fn = lambda: m.train_step(ids, dense, label, lr=1e-3)
elif which == "xd":
    from paddlerec_amd.xdeepfm import xDeepFMLayer
    m = xDeepFMLayer(1000001, 9, 13, 26, [128, 32], [512, 256, 128], device=DEV)
    ids = torch.randint(0, 1000001, (B, 26), device=DEV, generator=g); dense = torch.rand(B, 13, device=DEV, generator=g)
    label = (torch.rand(B, 1, device=DEV, generator=g) < 0.25).long()
    fn = lambda: m.train_step(ids, dense, label, lr=1e-3)
else:
    from paddlerec_amd.din import DINLayer
    T = 100
    m = DINLayer(64, 64, "sigmoid", False, True, 63001, 801, device=DEV)
    hi = torch.randint(0, 63001, (B, T), device=DEV, generator=g); hc = torch.randint(0, 801, (B, T), device=DEV, generator=g)
    ti = torch.randint(0, 63001, (B, 1), device=DEV, generator=g); tc = torch.randint(0, 801, (B, 1), device=DEV, generator=g)
    lens = torch.randint(1, T + 1, (B, 1), device=DEV, generator=g)
    mask = torch.where(torch.arange(T, device=DEV)[None] < lens, 0, -1000000000).long()
    label = (torch.rand(B, 1, device=DEV, generator=g) < 0.5).float()
    tis, tcs = ti.expand(B, T).contiguous(), tc.expand(B, T).contiguous()
    fn = lambda: m.train_step(hi, hc, ti, tc, label, mask, tis, tcs)
for _ in range(5): fn()
torch.cuda.synchronize()
a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): fn()
b.record(); torch.cuda.synchronize(); print(which, "B", B, "%.3f ms" % (a.elapsed_time(b)/10))
PY
for spec in "dcn 512" "xd 4096" "din 4096"; do
set -- $spec
python /tmp/m_run.py $1 $2 2>&1 | grep -v amdgpu
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$1 -o t -- python /tmp/m_run.py $1 $2 > /dev/null 2>&1
f=$(find $O/t$1 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("  ", r['Name'].replace('void ','').replace('rec::','').replace('(anonymous namespace)::','')[:84].ljust(84), r['Calls'].rjust(5), "%9.1f us" % (float(r['AverageNs'])/1e3), "%5.1f%%" % float(r['Percentage']))
print("   kernels per step ~ %.0f" % (sum(int(r['Calls']) for r in rows)/15))
PY
rm -rf $O/t$1
done
