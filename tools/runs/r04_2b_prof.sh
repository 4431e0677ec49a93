#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/l2b; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for D in 9 10; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$D -o t -- python $R/bench.py --shared-table --dim $D --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $O/line$D.json 2> $O/err$D.log
f=$(find $O/t$D -name "*kernel_stats.csv" | head -1)
echo "== D $D"; tail -1 $O/line$D.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    if 'at::' in r['Name'] or 'erfinv' in r['Name']: continue
    print(r['Name'].replace('void ','').replace('rec::','')[:80].ljust(80), r['Calls'].rjust(5), "%9.1f us" % (float(r['AverageNs'])/1e3), "%5.1f%%" % float(r['Percentage']))
PY
rm -rf $O/t$D
done
