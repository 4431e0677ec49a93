#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/c4; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o t -- python $R/bench.py --table ps --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $O/line.json 2> $O/err.log
tail -1 $O/line.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
f=$(find $O/t -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'at::' not in r['Name'] and 'erfinv' not in r['Name'] and 'spin' not in r['Name']]
for r in rows[:26]:
    print("  ", r['Name'].replace('void ','').replace('rec::','').replace('(anonymous namespace)::','')[:84].ljust(84), r['Calls'].rjust(5), "%9.1f us" % (float(r['AverageNs'])/1e3), "%5.1f%%" % float(r['Percentage']))
PY
rm -rf $O/t
