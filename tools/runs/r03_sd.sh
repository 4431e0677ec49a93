root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r03_sd; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $out/trace -o t --output-format csv -- python $root/tools/slot_dnn_bench.py --opt ps > $out/log.txt 2>&1
python3 - $out <<'PY'
import csv, glob, sys, re
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam_dense_kernel" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
one = rows[a + 1: b + 1]
t0 = int(one[0]["Start_Timestamp"])
print(len(one), "kernels, wall %.1f us" % ((int(one[-1]["End_Timestamp"]) - t0) / 1e3))
for r in one:
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("rec::", "")[:64]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d > 15: print("%8.1f %8.1f q%s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, d, r["Queue_Id"], n))
PY
