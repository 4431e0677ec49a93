#!/bin/bash
# usage: tools/pmc.sh <outdir-under-gpurun_out> <kernel-name-regex> <group-set: mem|compute> -- <command...>
# One rocprofv3 --pmc pass per counter group (never combined with tracing; each pass under its own
# timeout), results summarised per kernel into gpurun_out/<outdir>/summary.txt.
set -u
out=$1; shift; regex=$1; shift; set_=$1; shift; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out/$out
cd /tmp && export TMPDIR=/tmp
if [ "$set_" = "mem" ]; then
groups=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU"
 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum"
 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
else
groups=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA"
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
 "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
 "GRBM_GUI_ACTIVE"
)
fi
i=0
for g in "${groups[@]}"; do
  timeout 90 rocprofv3 --pmc $g -d $root/gpurun_out/$out/p$i -o p --output-format csv -- "$@" > $root/gpurun_out/$out/p$i.log 2>&1
  echo "pass $i rc=$?"
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
            acc[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(root + "/summary.txt", "w") as out:
    for k, d in acc.items():
        out.write("kernel %s\n" % k)
        for c in sorted(d):
            v = sorted(d[c]); out.write("  %-40s median %.6g  (n=%d)\n" % (c, v[len(v)//2], len(v)))
print(open(root + "/summary.txt").read())
PY
