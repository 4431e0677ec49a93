#!/bin/bash
# per-kernel standalone times of the slot-local grouping, with store-ablation flags
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04probe; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
for dbg in ${DBGS:-0 1 3 7 8 24}; do
  REC_SG_DBG=$dbg timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/t$dbg" -o t -- python "$R/tools/slot_group_probe.py" 50 2>&1 | grep "ids_group_slots"
  f=$(find "$O/t$dbg" -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'rec::' in r['Name']:
        print("   %-50s avg %7.1f min %7.1f" % (r['Name'].split('(')[0].replace('void rec::sg::','')[:50], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
  rm -rf "$O/t$dbg"
done
