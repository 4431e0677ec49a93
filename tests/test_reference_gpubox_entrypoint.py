"""Row N1, second entry point: the reference's OWN tools/static_gpubox_trainer.py, unmodified, on its own
models/rank/dnn/config_gpubox.yaml (the shipped gpubox config: tools/run_gpubox.sh:24) through
paddlerec_amd.run_reference over the compat namespace — paddle.enable_static / static.data / static.nn.sparse_embedding /
continuous_value_model / static.auc / Executor.train_from_dataset, fleet.init / is_worker / distributed_optimizer /
DistributedStrategy / util, InMemoryDataset with the reference's own pipe_command reader (a subprocess per file),
framework.core.PSGPU, FleetUtil.set_zero.
  * not gpu: operator backend = the oracle-backed stand-in (host logic of the tape executor);
  * -m gpu : runner.use_gpu=1, the HIP kernels (rec_feasign_rows, rec_emb_gather, rec_gemm_f32, rec_ids_group,
             rec_ps_push_rows, rec_auc_histogram).
Checked: the script's own log lines, and the saved pass checkpoint — the PS table's show / click counters must equal the
occurrence / click counts of the data file (every lookup of every epoch reached the accessor exactly once)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO

# The script writes into its working directory (./N_worker_*_program.prototxt, train_result_dict.txt): it always runs
# in the STAGED byte copy of the reference files (oracle/_ref/PaddleRec, made by oracle/make_ref_tree.py; build() does
# that in the build container and the copy travels to the GPU box) — never inside /root/reference.
REF = os.path.join(REPO, "oracle", "_ref", "PaddleRec")
pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "tools", "static_gpubox_trainer.py")),
                                reason="staged reference tree not present (python oracle/make_ref_tree.py)")


def _run(tmp_path, gpu, epochs=2):
    env = dict(os.environ, FLAGS_selected_gpus="0", TRAINING_ROLE="TRAINER", PADDLE_TRAINER_ID="0", OMP_NUM_THREADS="4",
               PYTHONDONTWRITEBYTECODE="1")
    if gpu:
        env.pop("REC_COMPAT_KERNELS", None)
    else:
        env["REC_COMPAT_KERNELS"] = "cpu_kernels"
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(REPO, "tests"), REPO, env.get("PYTHONPATH", "")])
    out = tmp_path / "out"
    cmd = [sys.executable, "-m", "paddlerec_amd.run_reference", os.path.join(REF, "tools", "static_gpubox_trainer.py"),
           "-m", "models/rank/dnn/config_gpubox.yaml", "-o", "runner.epochs=%d" % epochs,
           "runner.use_gpu=%d" % (1 if gpu else 0), "runner.model_save_path=%s" % out]
    r = subprocess.run(cmd, cwd=REF, env=env, capture_output=True, text=True, timeout=900)
    for f in os.listdir(REF):                                   # the script dumps its programs into the cwd
        if f.endswith("_program.prototxt") or f == "train_result_dict.txt":
            os.remove(os.path.join(REF, f))
    log = r.stdout + r.stderr
    assert r.returncode == 0, log[-4000:]
    return log, out


def _check(tmp_path, gpu):
    epochs = 2
    log, out = _run(tmp_path, gpu, epochs)
    assert "Run Success, Exit." in log
    ep = re.findall(r"Epoch: (\d+), using time: ([0-9.]+) second, ips: ([0-9.]+) example/sec. auc: ([0-9.]+)", log)
    assert [int(e[0]) for e in ep] == list(range(epochs)), log[-3000:]
    aucs = [float(e[3]) for e in ep]
    assert all(0.0 < a < 1.0 for a in aucs) and aucs[1] > aucs[0]          # the second pass ranks better than the first
    assert log.count("self.reader.load_into_memory cost") == epochs and log.count("begin_pass cost") == epochs
    assert "sync_mode = gpubox" in log                                     # get_strategy took the gpubox branch
    # the data file: 80 lines of `click:L dense_feature:... 1:id ... 26:id`
    data = open(os.path.join(REF, "models/rank/dnn/data/sample_data/train/sample_train.txt")).read().strip().split("\n")
    n_occ = sum(1 for ln in data for t in ln.split() if re.match(r"^\d+:", t) and int(t.split(":")[1]) != 0)
    n_click = sum(int(t.split(":")[1]) * 26 for ln in data for t in ln.split() if t.startswith("click:"))
    z = np.load(os.path.join(str(out), str(epochs - 1), "rec_gpubox.npz"))
    rec = z["table.embedding.records"]
    D = 9
    assert rec.shape[1] >= D + 7 and len(z["table.embedding.rows"]) == rec.shape[0] > 100
    assert float(rec[:, D].sum()) == epochs * n_occ, "show counters: every lookup pushed exactly once"
    assert float(rec[:, D + 1].sum()) == epochs * n_click, "click counters"
    assert np.all(rec[:, D + 4] >= 1) and np.abs(rec[:, 0]).max() > 0        # existing values, embed_w moved
    assert np.all(np.isfinite(rec))
    assert sum(1 for k in z.files if k.startswith("dense.")) == 10         # 5 Linear layers: weight + bias


def test_static_gpubox_trainer_runs_unmodified_cpu_backend(tmp_path):
    _check(tmp_path, gpu=False)


@pytest.mark.gpu
def test_static_gpubox_trainer_runs_unmodified_on_the_hip_kernels(tmp_path, engine_lib):
    _check(tmp_path, gpu=True)
