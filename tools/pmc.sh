#!/bin/bash
# usage: tools/pmc.sh <outdir-under-gpurun_out> <kernel-name-regex> -- <command...>
# One rocprofv3 --pmc pass per counter group (never combined with tracing), results summarised per kernel.
set -u
out=$1; shift; regex=$1; shift; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out/$out
cd /tmp && export TMPDIR=/tmp
groups=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VALU"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCR_TCP_STALL_CYCLES_sum"
 "TCP_TCC_WRITE_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum"
 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum"
 "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_TAG_STALL_sum"
 "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum TCC_REQ_sum"
 "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum"
 "GRBM_GUI_ACTIVE GRBM_TA_BUSY GRBM_TC_BUSY GRBM_EA_BUSY"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
i=0
for g in "${groups[@]}"; do
  rocprofv3 --pmc $g -d $root/gpurun_out/$out/p$i -o p --output-format csv -- "$@" > $root/gpurun_out/$out/p$i.log 2>&1
  i=$((i+1))
done
python3 - "$root/gpurun_out/$out" "$regex" <<'PY'
import csv, glob, re, sys, collections
root, rx = sys.argv[1], re.compile(sys.argv[2])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if rx.search(k):
            acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(root + "/summary.txt", "w") as out:
    for k, d in acc.items():
        out.write("kernel %s\n" % k)
        for c in sorted(d):
            v = sorted(d[c]); out.write("  %-40s median %.6g  (n=%d)\n" % (c, v[len(v)//2], len(v)))
print(open(root + "/summary.txt").read())
PY
