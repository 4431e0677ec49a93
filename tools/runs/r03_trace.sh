# kernel-trace stats of bench.py on the current tree -> gpurun_out/r03_trace/stats.txt
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r03_trace; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- python $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench.log 2>&1
echo rc=$?
tail -1 $out/bench.log
python3 - $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open(sys.argv[1] + "/stats.txt", "w") as o:
    for r in rows[:40]:
        line = "%-90s calls %6s avg %10.1f ns total %6.2f%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]), float(r["Percentage"]))
        print(line); o.write(line + "\n")
PY
