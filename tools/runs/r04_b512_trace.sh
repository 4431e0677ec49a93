#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/b512; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- python $R/bench.py --batch 512 --steps 50 --warmup 20 --no-cpu-baseline --no-other-configs > $O/bench.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys, re
f = glob.glob(sys.argv[1] + "/t/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam_dense_kernel" in r["Kernel_Name"]]
a, b = idx[-12], idx[-2]
steps = 10
seg = rows[a + 1:b + 1]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e3 / steps
span = (int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3 / steps
print("per step under rocprof: %d kernels, sum of kernel durations %.1f us, wall %.1f us" % (len(seg) / steps, busy, span))
one = rows[idx[-3] + 1: idx[-2] + 1]
t0 = int(one[0]["Start_Timestamp"])
for r in one:
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("rec::", "")[:64]
    print("%7.1f %6.1f %s  grid %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, n, r.get("Grid_Size_X", "")))
PY
rm -rf $O/t
