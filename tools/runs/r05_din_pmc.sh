# PMC passes over the DIN attention kernels at B 4096, T 100 (tools/din_step_loop.py): where the wave cycles go
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/din_pmc
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for g in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_FLAT"; do
  timeout 200 rocprofv3 --pmc $g -d $out/p$i -o p --output-format csv -- python $root/tools/din_step_loop.py 4096 ${DIN_T:-100} 6 > $out/log$i.txt 2>&1
  echo "pmc $i rc=$?"
  i=$((i+1))
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "din_attention" not in k: continue
        k = "fwd_ct" if "fwd_ct" in k else "bwd_ct" if "bwd_ct" in k else k[:40]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for k, d in acc.items():
        fo.write("== %s\n" % k)
        for c, v in sorted(d.items()):
            fo.write("  %-28s mean %.4g  (n %d)\n" % (c, sum(v) / len(v), len(v)))
print(open(out + "/summary.txt").read())
PY
