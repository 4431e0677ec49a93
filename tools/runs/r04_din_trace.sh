#!/bin/bash
# kernel trace of one DIN train step at the shipped batch size (32 x T 152)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/din; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 200 python $R/tools/din_small_bench.py 2>&1 | grep -v amdgpu | tail -4
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- python $R/tools/din_small_bench.py > $O/log.txt 2>&1
python3 - $O <<'PY'
import csv, glob, sys, re
f = glob.glob(sys.argv[1] + "/t/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# a step ends with the last SGD kernel; find a repeating period by the attention fwd kernel
idx = [i for i, r in enumerate(rows) if "din_attention" in r["Kernel_Name"] and "fwd" in r["Kernel_Name"]]
a, b = idx[len(idx)//2], idx[len(idx)//2 + 1]
one = rows[a:b]
t0 = int(one[0]["Start_Timestamp"])
print("kernels per step:", len(one), " wall %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
for r in one:
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("rec::", "")[:70]
    print("%7.1f %6.1f %s  grid %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, n, r.get("Grid_Size_X", r.get("Grid_Size", ""))))
PY
rm -rf $O/t
