root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r03_xd; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- python $root/tools/xdeepfm_chunk_bench.py 4096 > $out/log.txt 2>&1
tail -1 $out/log.txt
python3 - $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)
rows = []
for ff in f: rows += list(csv.DictReader(open(ff)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    print("%-100s calls %5s avg %9.1f us  %5.1f%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
# shapes of the GEMMs of one step: grid sizes from the trace
t = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
seen = {}
for r in csv.DictReader(open(t)):
    if "gemm_f32" in r["Kernel_Name"]:
        k = (r["Kernel_Name"][:60], r["Grid_Size_X"], r["Grid_Size_Y"])
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        seen.setdefault(k, []).append(d)
for k, v in sorted(seen.items(), key=lambda kv: -sum(kv[1])):
    print(k, "n=%d avg %.1f us" % (len(v), sum(v) / len(v)))
PY
