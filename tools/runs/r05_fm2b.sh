#!/bin/bash
# layout 2b (one shared [1 000 001, D] table, D 9 / 10): row-group vs block-tile FM kernels, 128-B vs 64-B records
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05_fm2b; mkdir -p "$O"; cd "$R"
for D in 9 10; do
for cfg in "0 32" "1 32" "0 16" "1 16"; do
  set -- $cfg
  REC_FM_TILE=$1 REC_TABLE_RECORD_FLOATS=$2 timeout 300 python bench.py --shared-table --dim $D --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 > $O/d${D}_tile$1_rec$2.json 2> $O/d${D}_tile$1_rec$2.err
  python - "$O/d${D}_tile$1_rec$2.json" $D $1 $2 <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]; pk = r["per_kernel"]
print("D %s tile %s rec %s: step %.3f ms | back-to-back fm_fwd %.1f us fm_bwd %.1f us frac %.3f | in-step fwd %.1f us bwd %.1f us frac %.3f | sparse_adam %.0f us"
      % (sys.argv[2], sys.argv[3], sys.argv[4], d["ms_per_step"], pk["fm_fwd"]["us"], pk["fm_bwd"]["us"], r["frac"],
         1e3 * r["in_step_event"]["fm_fwd_ms"], 1e3 * r["in_step_event"]["fm_bwd_ms"], r["in_step_frac"], 1e3 * d["kernels_ms"].get("sparse_adam", 0)))
PY
done; done 2>&1 | tee $O/summary.txt
