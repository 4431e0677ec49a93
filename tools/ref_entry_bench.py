#!/usr/bin/env python3
"""L-trainer of the reference's own entry point on the GPU (SURVEY §8(d) measurement levels): the UNMODIFIED
tools/trainer.py on models/rank/deepfm/config_bigdata.yaml (bs 512, D 9, fc 400x3, non-lazy Adam) over synthetic slot
text, three ways —
  unpatched : net.py as the reference ships it, every paddle op through the compat namespace
  patched   : net.py with integration/deepfm_net.patch (FM block = the custom operator rec_deepfm_fm through the shim)
  engine    : paddlerec_amd.trainer on the same config + files (the engine's own loop: C++ parser, fused step)
One JSON line each: the loop's own `ips` (mean of the printed intervals after the first).  Needs the staged trees
(oracle/_ref, built by __graft_entry__.build() in the build container)."""
import json
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(batches=40):
    trees = {"unpatched": os.path.join(REPO, "oracle", "_ref", "PaddleRec"),
             "patched": os.path.join(REPO, "oracle", "_ref", "PaddleRec_rec_ops")}
    if not all(os.path.isdir(os.path.join(t, "tools")) for t in trees.values()):
        print(json.dumps({"workload": "reference tools/trainer.py, deepfm config_bigdata.yaml", "error": "staged trees missing"}))
        return
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", REC_COMPAT_SEED="3")
    env.pop("REC_COMPAT_KERNELS", None)
    env["PYTHONPATH"] = os.pathsep.join([REPO, env.get("PYTHONPATH", "")])
    rng = np.random.default_rng(20250404)
    with tempfile.TemporaryDirectory() as tmp:
        data = os.path.join(tmp, "train")
        os.makedirs(data)
        n = 512 * batches
        ids = rng.integers(1, 1000001, (n, 26))
        dense = rng.random((n, 13))
        lab = rng.random(n) < 0.25
        with open(os.path.join(data, "part-0"), "w") as f:
            for b in range(n):
                f.write("click:%d " % lab[b] + " ".join("dense_feature:%.6f" % v for v in dense[b]) + " " +
                        " ".join("%d:%d" % (s + 1, ids[b, s]) for s in range(26)) + "\n")
        over = ["-o", "runner.train_data_dir=%s" % data, "runner.epochs=1", "runner.print_interval=10", "runner.use_gpu=True",
                "runner.model_save_path=%s" % os.path.join(tmp, "ck")]
        jobs = [(k, [sys.executable, "-m", "paddlerec_amd.run_reference", os.path.join(t, "tools", "trainer.py"), "-m",
                     os.path.join(t, "models/rank/deepfm/config_bigdata.yaml")] + over, t) for k, t in trees.items()]
        jobs.append(("engine", [sys.executable, "-m", "paddlerec_amd.trainer", "-m",
                                os.path.join(trees["unpatched"], "models/rank/deepfm/config_bigdata.yaml")] + over, REPO))
        for kind, cmd, cwd in jobs:
            r = subprocess.run(cmd, cwd=cwd, env=env, capture_output=True, text=True, timeout=300)
            log = r.stdout + r.stderr
            ips = [float(x) for x in re.findall(r"ips: ([0-9.]+) ins/s", log)]
            line = {"workload": "tools/trainer.py loop, deepfm config_bigdata.yaml (bs 512, D 9, fc 400x3), %d batches of "
                                "synthetic slot text, GPU" % batches, "entry": kind}
            if r.returncode != 0 or len(ips) < 2:
                line["error"] = "rc %d: %s" % (r.returncode, log[-200:])
            else:
                line["samples_per_s"] = round(sum(ips[1:]) / len(ips[1:]), 1)
                rc = re.findall(r"avg_reader_cost: ([0-9.]+) sec, avg_batch_cost: ([0-9.]+) sec", log)
                if rc:
                    line["reader_ms"], line["batch_ms"] = (round(1e3 * float(rc[-1][0]), 3), round(1e3 * float(rc[-1][1]), 3))
            print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
