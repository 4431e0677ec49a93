root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r03_small; rm -rf $out; mkdir -p $out
timeout 300 python bench.py --batch 512 --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B512 eager', d['ms_per_step'], d['host_issue_ms'].get('step_issue_total'))"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $out/trace -o t --output-format csv -- python $root/bench.py --batch 512 --steps 50 --warmup 20 --no-cpu-baseline > $out/bench.log 2>&1
python3 - $out <<'PY'
import csv, glob, sys, re
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
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
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("rec::", "")[:60]
    print("%7.1f %6.1f %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, n))
PY
