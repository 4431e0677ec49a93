#!/bin/bash
# kernel-trace stats of the gpubox model's train step (slot_dnn_bench --opt ps)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/slotdnn; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o t -- python $R/tools/slot_dnn_bench.py --opt ps > $O/line.json 2> $O/err.log
f=$(find $O/t -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv
tail -1 $O/line.json | cut -c1-900
python - "$O/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:40]:
    print(r['Name'].replace('void ','').replace('rec::','')[:84].ljust(84), r['Calls'].rjust(5), "%9.1f us" % (float(r['AverageNs'])/1e3), "%5.1f%%" % float(r['Percentage']))
PY
rm -rf $O/t
