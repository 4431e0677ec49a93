#!/usr/bin/env python3
"""Prints the per-queue kernel timeline of the LAST step found in a rocprofv3 --kernel-trace csv
(kernel name, queue, start offset, duration, gap to the previous kernel on the same queue) — used to see what
overlaps with what and where a stream idles.  Usage: trace_timeline.py <kernel_trace.csv> [anchor-substring] [mid]
(mid: the step in the MIDDLE of the trace instead of one of the last — bench.py's last steps carry event brackets whose
markers show up as 7-16 us gaps; its timed region sits in the middle of the run)"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else "fm_fwd_"
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"),
              r.get("Stream_Id", "?")) for r in rows), key=lambda x: x[0])
starts = [i for i, k in enumerate(ks) if anchor in k[2]]
if len(starts) < 3:
    sys.exit("anchor kernel not found often enough")
if len(sys.argv) > 3 and sys.argv[3] == "mid":
    a, b = starts[len(starts) // 2], starts[len(starts) // 2 + 1]
else:
    a, b = starts[-3], starts[-2]      # one full step between two anchors, not the last (teardown) one
t0 = ks[a][0]
last_end = {}
print("step length %.3f ms" % ((ks[b][0] - t0) / 1e6))
for s, e, name, q, st in ks[a:b]:
    key = (q, st)
    gap = (s - last_end[key]) / 1e3 if key in last_end else 0.0
    last_end[key] = e
    short = name.split("(")[0].replace("void ", "").replace("rec::", "")[:70]
    print("q%-3s s%-3s  +%8.1f us  dur %7.1f us  gap %7.1f  %s" % (q, st, (s - t0) / 1e3, (e - s) / 1e3, gap, short))
