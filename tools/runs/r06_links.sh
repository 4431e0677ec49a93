#!/bin/bash
# round 6: the configs[4] share (1.25e9 hashed PS rows, world 1 through the row-sharded code path) with the exchanges of an
# 8-GPU run emulated by paced link kernels (REC_EMULATE_LINKS=8), against the same step without them
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06links; mkdir -p "$O"; cd "$R"
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --force-sharded --table ps --hashed-rows 1250000000 --no-cpu-baseline --steps 20 --warmup 5 2>"$O/$tag.err" | tail -1 > "$O/$tag.json"; python - "$O/$tag.json" "$tag" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("%-28s %.4f ms" % (sys.argv[2], d["ms_per_step"]), {k:round(v,3) for k,v in d.get("kernels_ms",{}).items()})
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
for i in 1 2; do
run nolinks A=1
run links REC_EMULATE_LINKS=8
run links_dw_exact REC_EMULATE_LINKS=8 REC_GEMM_BF16X3_DW=0
run links_part16 REC_EMULATE_LINKS=8 REC_SHARD_TAIL=partition REC_SHARD_SIDE_CUS=16
run links_part8 REC_EMULATE_LINKS=8 REC_SHARD_TAIL=partition REC_SHARD_SIDE_CUS=8
run links_part16_exact REC_EMULATE_LINKS=8 REC_SHARD_TAIL=partition REC_SHARD_SIDE_CUS=16 REC_GEMM_BF16X3_DW=0
run links_serial REC_EMULATE_LINKS=8 REC_SHARD_TAIL=serial
run links_dedup REC_EMULATE_LINKS=8 REC_SHARD_DEDUP=1
run nolinks_dedup REC_SHARD_DEDUP=1
done
tail -3 $O/links.err
