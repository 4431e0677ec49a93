#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace stats of bench.py, then separate --pmc passes for the
# HBM request counters of the FM kernels.  Outputs under gpurun_out/prof_<tag>/ (copy summaries to profiles/).
tag=${1:-r01}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
CMD="python $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs"
timeout 200 rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- $CMD > $out/bench_under_rocprof.log 2>&1
echo "trace rc=$?"
i=0
for g in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
         "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 200 rocprofv3 --pmc $g -d $out/pmc$i -o p --output-format csv -- $CMD > $out/pmc$i.log 2>&1
  echo "pmc $i rc=$?"
  i=$((i+1))
done
python3 - "$out" <<'PY'
import csv, glob, json, re, sys, collections
out = sys.argv[1]
# counters keyed by the FULL kernel name incl. template arguments: two instantiations of one template are two
# kernels (r01 took one median over both sparse_adam_rows_kernel<4,4> and <1,1> launches)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*$", "", r.get("Kernel_Name", "")).replace("void ", "").replace("rec::", "").strip()
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, d in acc.items():
    med = {c: sorted(v)[len(v) // 2] for c, v in d.items()}
    rd = med.get("TCC_EA0_RDREQ_128B_sum", 0) * 128 + med.get("TCC_EA0_RDREQ_64B_sum", 0) * 64 + med.get("TCC_EA0_RDREQ_32B_sum", 0) * 32
    w64 = med.get("TCC_EA0_WRREQ_64B_sum", 0)
    wr = w64 * 64 + (med.get("TCC_EA0_WRREQ_sum", 0) - w64) * 32
    res[k] = dict(counters=med, hbm_read_bytes=rd, hbm_write_bytes=wr, hbm_bytes=rd + wr, launches=len(next(iter(d.values()))))
# aliases bench.py reads: the ONE instantiation of each FM kernel the bench shape launches
for alias in ("fm_fwd_kernel", "fm_bwd_kernel", "sparse_adam_record_kernel"):
    hits = [k for k in res if k.startswith(alias + "<")]
    if len(hits) == 1:
        res[alias] = dict(res[hits[0]], instantiation=hits[0])
    elif hits:
        print("WARNING: %d instantiations of %s: %s" % (len(hits), alias, hits))
json.dump(res, open(out + "/pmc_traffic.json", "w"), indent=1)
big = sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes"])[:14]
print(json.dumps({k: {x: v[x] for x in ("hbm_read_bytes", "hbm_write_bytes", "hbm_bytes", "launches")} for k, v in big}, indent=1))
PY
ls $out/trace/* | head
