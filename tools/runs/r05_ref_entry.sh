#!/bin/bash
# GPU box: the reference's UNMODIFIED tools/trainer.py on models/rank/deepfm/config_bigdata.yaml (bs 512, D 9, 400x3),
#   (a) unpatched net.py over the compat namespace,  (b) net.py patched by integration/deepfm_net.patch (custom op through
#   the shim),  (c) paddlerec_amd.trainer on the same files — `ips` of each, and rocprofv3 kernel stats of (a) and (b).
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05_ref_entry; mkdir -p "$O"; cd "$R"
export PYTHONPATH=$R:$PYTHONPATH PYTHONDONTWRITEBYTECODE=1 REC_COMPAT_SEED=3
D=/tmp/slot_synth; mkdir -p $D
python - <<'PY'
import numpy as np
rng = np.random.default_rng(20250404)
B = 512 * 60
ids = rng.integers(1, 1000001, (B, 26)); ids[rng.random((B, 26)) < 0.03] = 0
dense = rng.random((B, 13)); lab = (rng.random(B) < 0.25).astype(int)
with open("/tmp/slot_synth/part-0", "w") as f:
    for b in range(B):
        f.write("click:%d " % lab[b] + " ".join("dense_feature:%.6f" % v for v in dense[b]) + " " +
                " ".join("%d:%d" % (s + 1, ids[b, s]) for s in range(26)) + "\n")
PY
ARGS="-o runner.train_data_dir=$D runner.epochs=1 runner.print_interval=10 runner.use_gpu=True runner.model_save_path=/tmp/ck_out"
for t in PaddleRec PaddleRec_rec_ops; do
  T=$R/oracle/_ref/$t
  (cd $T && timeout 600 python -m paddlerec_amd.run_reference tools/trainer.py -m models/rank/deepfm/config_bigdata.yaml $ARGS) > $O/ips_$t.log 2>&1
  echo "$t rc=$?"; grep "ips:" $O/ips_$t.log | tail -3
done
timeout 600 python -m paddlerec_amd.trainer -m $R/oracle/_ref/PaddleRec/models/rank/deepfm/config_bigdata.yaml $ARGS > $O/ips_engine_trainer.log 2>&1
echo "engine trainer rc=$?"; grep "ips" $O/ips_engine_trainer.log | tail -3
cd /tmp && export TMPDIR=/tmp
for t in PaddleRec PaddleRec_rec_ops; do
  T=$R/oracle/_ref/$t
  (cd $T && timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace_$t -o t --output-format csv -- python -m paddlerec_amd.run_reference tools/trainer.py -m models/rank/deepfm/config_bigdata.yaml $ARGS) > $O/trace_$t.log 2>&1
  echo "trace $t rc=$?"
  f=$(find $O/trace_$t -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/kernel_stats_$t.csv && head -30 $f | cut -c1-160
  find $O/trace_$t -name "*kernel_trace.csv" -size +30M -delete
done
