#!/bin/bash
# the C step entry: parity tests, then B 512 timings (eager python / planned python / one C call) in alternating pairs
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
timeout 600 python -m pytest $R/tests/test_deepfm_step_c.py -m gpu -x -q 2>&1 | tail -5
run() { timeout 200 python $R/bench.py --batch 512 --steps 400 --warmup 40 --no-cpu-baseline --no-other-configs "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-28s ms_per_step %.4f  samples/s %.3e  host issue ms %.4f' % ('$LABEL', d['ms_per_step'], d['value'], d.get('host_issue_ms',{}).get('step_issue_total', -1)))"; }
for rep in 1 2; do
  LABEL="eager python (ctypes/launch)"; REC_STEP_PLAN=0 run
  LABEL="planned python (call list)"; run
  LABEL="rec_deepfm_train_step"; run --c-step
done 2>&1 | tee $O/cstep_b512.txt
LABEL="B 65536 python overlapped"; timeout 200 python $R/bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B65536 python', d['ms_per_step'])" | tee -a $O/cstep_b512.txt
timeout 200 python $R/bench.py --no-cpu-baseline --no-other-configs --c-step 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B65536 c-step (one stream)', d['ms_per_step'])" | tee -a $O/cstep_b512.txt
