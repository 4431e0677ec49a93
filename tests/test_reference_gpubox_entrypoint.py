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
PATCHED = os.path.join(REPO, "oracle", "_ref", "PaddleRec_rec_ops")      # the same tree with integration/*.patch applied
needs_patched = pytest.mark.skipif(not os.path.isfile(os.path.join(PATCHED, "models", "rank", "dnn", "net.py")),
                                   reason="patched reference tree not present (python integration/apply.py)")
pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "tools", "static_gpubox_trainer.py")),
                                reason="staged reference tree not present (python oracle/make_ref_tree.py)")


MODELS = {   # the two nets the reference wires for gpubox (dnn/net.py:71, wide_deep/net.py:80): config, data, dense tensors
    "dnn": ("models/rank/dnn/config_gpubox.yaml", "models/rank/dnn/data/sample_data/train/sample_train.txt", 10),
    "wide_deep": ("models/rank/wide_deep/config_gpups.yaml", "models/rank/wide_deep/data/sample_data/train/sample_train.txt",
                  12),          # + the wide part's Linear(13, 1)
}


def _run(tmp_path, gpu, epochs=2, gpus="0", batch=None, name="out", extra_env=None, model="dnn", args=(), tree=REF):
    env = dict(os.environ, FLAGS_selected_gpus=gpus, TRAINING_ROLE="TRAINER", PADDLE_TRAINER_ID="0", OMP_NUM_THREADS="4",
               PYTHONDONTWRITEBYTECODE="1")
    env.pop("WORLD_SIZE", None)
    env.setdefault("REC_COMPAT_SEED", "7")                      # the script sets no seed: pin the initial parameters
    env.update(extra_env or {})
    if gpu:
        env.pop("REC_COMPAT_KERNELS", None)
    else:
        env["REC_COMPAT_KERNELS"] = "cpu_kernels"
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(REPO, "tests"), REPO, env.get("PYTHONPATH", "")])
    out = tmp_path / name
    cmd = [sys.executable, "-m", "paddlerec_amd.run_reference", os.path.join(tree, "tools", "static_gpubox_trainer.py"),
           "-m", MODELS[model][0]] + list(args) + ["-o", "runner.epochs=%d" % epochs,
           "runner.use_gpu=%d" % (1 if gpu else 0), "runner.model_save_path=%s" % out]
    if batch:
        cmd.append("runner.train_batch_size=%d" % batch)
    r = subprocess.run(cmd, cwd=tree, env=env, capture_output=True, text=True, timeout=900)
    for f in os.listdir(tree):                                  # the script dumps its programs into the cwd
        if f.endswith("_program.prototxt") or f == "train_result_dict.txt":
            os.remove(os.path.join(tree, f))
    log = r.stdout + r.stderr
    assert r.returncode == 0, log[-4000:]
    return log, out


def _check(tmp_path, gpu, model="dnn", args=()):
    epochs = 2
    log, out = _run(tmp_path, gpu, epochs, model=model, args=args)
    assert "Run Success, Exit." in log
    ep = re.findall(r"Epoch: (\d+), using time: ([0-9.]+) second, ips: ([0-9.]+) example/sec. auc: ([0-9.]+)", log)
    assert [int(e[0]) for e in ep] == list(range(epochs)), log[-3000:]
    aucs = [float(e[3]) for e in ep]
    assert all(0.0 < a < 1.0 for a in aucs) and aucs[1] > aucs[0]          # the second pass ranks better than the first
    assert log.count("self.reader.load_into_memory cost") == epochs and log.count("begin_pass cost") == epochs
    assert "sync_mode = gpubox" in log                                     # get_strategy took the gpubox branch
    # the data file: 80 lines of `click:L dense_feature:... 1:id ... 26:id`
    data = open(os.path.join(REF, MODELS[model][1])).read().strip().split("\n")
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
    assert sum(1 for k in z.files if k.startswith("dense.")) == MODELS[model][2]      # Linear layers: weight + bias
    return log


def test_static_gpubox_trainer_runs_unmodified_cpu_backend(tmp_path):
    _check(tmp_path, gpu=False)


@pytest.mark.skipif(not os.path.isfile(os.path.join(REF, MODELS["wide_deep"][0])), reason="wide_deep not staged")
def test_static_gpubox_trainer_wide_deep_config_gpups_cpu_backend(tmp_path):
    """rank/wide_deep is the second net the reference wires for gpubox (wide_deep/net.py:73-100,
    config_gpups.yaml) — same unmodified trainer, same checks."""
    _check(tmp_path, gpu=False, model="wide_deep")


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.isfile(os.path.join(REF, MODELS["wide_deep"][0])), reason="wide_deep not staged")
def test_static_gpubox_trainer_wide_deep_config_gpups_on_the_hip_kernels(tmp_path, engine_lib):
    _check(tmp_path, gpu=True, model="wide_deep")


def test_profiler_options_do_not_crash_the_unmodified_trainer(tmp_path):
    """--profiler_options (static_gpubox_trainer.py:57,255 -> tools/profiler.py:82-110 -> paddle.utils.profiler.start /
    stop_profiler): the compat hooks open / close a ROCTX range (no-op without libroctx) instead of raising."""
    log = _check(tmp_path, gpu=False, args=("--profiler_options", "batch_range=[0,1];state=GPU;exit_on_finished=false;profile_path=%s" % tmp_path))
    assert "profiler window closed" in log


@pytest.mark.gpu
def test_static_gpubox_trainer_runs_unmodified_on_the_hip_kernels(tmp_path, engine_lib):
    _check(tmp_path, gpu=True)


def _check_two_ranks_equal_one(tmp_path, gpu):
    """tools/run_gpubox.sh:21 with two GPUs named: the UNMODIFIED script as 2 ranks (run_reference spawns them; gloo here —
    two processes on the CPU backend, or sharing cuda:0 with the HIP kernels), the table of static.nn.sparse_embedding
    row-sharded over the ranks, every global batch of 2 x 32 samples split between them — against ONE unsharded run of
    the same script at batch 64: the same samples per step, so the merged pass checkpoint must hold the same keys, the
    same counters (exactly), and weights / g2sums at the stated bar."""
    epochs = 2
    seed = {"REC_COMPAT_SEED": "1234"}
    log2, out2 = _run(tmp_path, gpu, epochs, gpus="0,1", batch=32, name="two", extra_env=seed)
    log1, out1 = _run(tmp_path, gpu, epochs, gpus="0", batch=64, name="one", extra_env=seed)
    assert log2.count("Run Success, Exit.") == 2 and log1.count("Run Success, Exit.") == 1
    assert os.path.isfile(os.path.join(str(out2), str(epochs - 1), "rec_gpubox.shard1of2.npz"))
    a = np.load(os.path.join(str(out2), str(epochs - 1), "rec_gpubox.npz"))
    b = np.load(os.path.join(str(out1), str(epochs - 1), "rec_gpubox.npz"))
    assert np.array_equal(a["table.embedding.rows"], b["table.embedding.rows"])
    ra, rb = a["table.embedding.records"], b["table.embedding.records"]
    D = 9
    for col, what in ((D, "show"), (D + 1, "click"), (D + 4, "state"), (D + 6, "unseen_days")):
        assert np.array_equal(ra[:, col], rb[:, col]), what
    np.testing.assert_allclose(ra[:, D + 5], rb[:, D + 5], rtol=1e-6, err_msg="delta_score")
    wscale = float(np.abs(rb[:, :D]).max())
    werr = np.abs(ra[:, :D] - rb[:, :D])
    assert float(np.mean(werr <= 1e-5 * wscale)) >= 0.98 and float(werr.max()) <= 1e-4 * wscale, (werr.max(), wscale)
    np.testing.assert_allclose(ra[:, D + 2:D + 4], rb[:, D + 2:D + 4], rtol=1e-4, atol=1e-5 * float(rb[:, D + 2:D + 4].max()))
    # the two shards hold disjoint keys: owner(row) = row % 2
    for r in range(2):
        z = np.load(os.path.join(str(out2), str(epochs - 1), "rec_gpubox.shard%dof2.npz" % r))
        assert len(z["table.embedding.rows"]) > 50 and np.all(z["table.embedding.rows"] % 2 == r)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from helpers import assert_adam_weights_close
    steps = epochs * 2                                         # 80 samples: one global batch of 64 + one of 16 per epoch
    for k in [k for k in b.files if k.startswith("dense.")]:
        assert_adam_weights_close(a[k], b[k], lr=1e-3, steps=steps, err_msg=k)


def test_static_gpubox_trainer_two_ranks_equal_one_unsharded_run_cpu_backend(tmp_path):
    _check_two_ranks_equal_one(tmp_path, gpu=False)


@pytest.mark.gpu
def test_static_gpubox_trainer_two_ranks_on_one_gpu_equal_one_unsharded_run(tmp_path, engine_lib):
    _check_two_ranks_equal_one(tmp_path, gpu=True)


def _oracle_replay(init, data_lines, epochs, batch, table_rows, D=9):
    """The same two passes on the oracles, independent of the compat namespace: feasigns -> rows (slot_dnn_ref.feasign_rows,
    the engine's published hash), rank/dnn forward / backward (oracle/dnn_ref.py, pinned to the reference's net.py
    golden), the PS accessor's push on the merged row gradients (oracle/ps_ref.py: show = 1 per occurrence, click =
    label, gradient of the SUMMED loss), Paddle's Adam on the dense parameters (oracle/deepfm_ref.adam_update).
    -> (record table [table_rows, 16], dense parameters, their Adam moments)."""
    from oracle import deepfm_ref as R
    from oracle import dnn_ref as DN
    from oracle import ps_ref, slot_dnn_ref
    acc = dict(lr=0.05, initial_g2sum=3.0, bounds=(-10.0, 10.0), initial_range=1e-4, embedx_threshold=10.0,
               nonclk_coeff=0.1, click_coeff=1.0, seed=2025)                    # ops.PsTable defaults = Paddle's defaults
    lay = dict(embed_off=0, embedx_off=1, embedx_dim=D - 1, stat_off=D)
    rec = np.zeros((table_rows, 16), np.float32)
    n_lin = len(init) // 2
    mlp_w = [init["dense.%d" % (2 * i)].copy() for i in range(n_lin)]
    mlp_b = [init["dense.%d" % (2 * i + 1)].copy() for i in range(n_lin)]
    mom = {("w", i): (np.zeros_like(mlp_w[i]), np.zeros_like(mlp_w[i])) for i in range(n_lin)}
    mom.update({("b", i): (np.zeros_like(mlp_b[i]), np.zeros_like(mlp_b[i])) for i in range(n_lin)})
    samples = []
    for ln in data_lines:
        tok = dict((t.split(":")[0], t.split(":")[1]) for t in ln.split() if ":" in t and not t.startswith("dense"))
        dense = [float(t.split(":")[1]) for t in ln.split() if t.startswith("dense_feature:")]
        samples.append((int(tok["click"]), [int(tok.get(str(s), 0)) for s in range(1, 27)], dense))
    step = 0
    for _ in range(epochs):
        for lo in range(0, len(samples), batch):
            part = samples[lo:lo + batch]
            B = len(part)
            label = np.int64([[s[0]] for s in part])
            keys = np.array([s[1] for s in part], np.uint64)
            dense = np.float32([s[2] for s in part])
            rows = slot_dnn_ref.feasign_rows(keys.reshape(-1), table_rows).reshape(B, 26)
            step += 1
            o = DN.loss_and_grads(rows, dense, label, dict(W=rec[:, :D].copy(), mlp_w=mlp_w, mlp_b=mlp_b))
            flat = rows.reshape(-1)
            uniq, merged, counts = R.merge_rows(flat, flat != 0, o["row_grad"])
            clicks = np.zeros(len(uniq), np.int64)
            lab_of_pos = np.repeat(label.reshape(-1), 26)
            idx = {int(r): i for i, r in enumerate(uniq)}
            for pos in np.nonzero(flat != 0)[0]:
                clicks[idx[int(flat[pos])]] += lab_of_pos[pos]
            ps_ref.push_rows(rec, lay, uniq, merged[:, 0], merged[:, 1:], counts, clicks, dict(acc, grad_scale=float(B)))
            for i in range(n_lin):
                R.adam_update(mlp_w[i], *mom[("w", i)], o["mlp_dw"][i].reshape(mlp_w[i].shape).astype(np.float32), step, lr=1e-3)
                R.adam_update(mlp_b[i], *mom[("b", i)], o["mlp_db"][i].reshape(mlp_b[i].shape).astype(np.float32), step, lr=1e-3)
    return rec, mlp_w, mlp_b


def _check_against_oracles(tmp_path, gpu, tree=REF):
    """the entry-point test compared log lines and counter sums only.  Here the pass checkpoint the
    UNMODIFIED script wrote — every record of the PS table (weights, both g2sums, counters, state) and the dense
    parameters — against an independent replay on the oracles, from the same initial dense parameters
    (REC_COMPAT_DUMP_INIT) and a small hashed table (REC_GPUBOX_TABLE_ROWS)."""
    epochs, rows_n = 2, 20011
    init_path = str(tmp_path / "init.npz")
    log, out = _run(tmp_path, gpu, epochs, name="orc", tree=tree,
                    extra_env={"REC_COMPAT_DUMP_INIT": init_path, "REC_GPUBOX_TABLE_ROWS": str(rows_n)})
    init = dict(np.load(init_path))
    assert len(init) == 10 and init["dense.0"].shape == (26 * 9 + 13, 512)
    data = open(os.path.join(REF, "models/rank/dnn/data/sample_data/train/sample_train.txt")).read().strip().split("\n")
    rec, mlp_w, mlp_b = _oracle_replay(init, data, epochs, 32, rows_n)
    z = np.load(os.path.join(str(out), str(epochs - 1), "rec_gpubox.npz"))
    got_rows, got = z["table.embedding.rows"], z["table.embedding.records"]
    D = 9
    want_rows = np.nonzero(rec[:, D + 4] != 0)[0]
    assert np.array_equal(got_rows, want_rows)                                 # the same keys exist
    want = rec[want_rows]
    for col, what in ((D, "show"), (D + 1, "click"), (D + 4, "state"), (D + 6, "unseen_days")):
        assert np.array_equal(got[:, col], want[:, col]), what
    assert (want[:, D + 4] == 2).any() and (want[:, D + 4] == 1).any()         # embedx created for some keys, not for all
    np.testing.assert_allclose(got[:, D + 5], want[:, D + 5], rtol=1e-6, err_msg="delta_score")
    # Weights and g2sums: 1e-5 of the tensor's scale, or 4 x the MEASURED sensitivity of this two-pass trajectory to fp32
    # round-off — the same oracle replay from initial dense parameters moved by ONE ulp (the dense Adam turns the sign
    # noise of ~eps-sized gradients into lr-sized steps, which reach the embedding gradients of later steps): what the
    # oracle cannot reproduce of itself, the kernels are not asked to reproduce of it
    init_ulp = {k: np.nextafter(v, np.float32(np.inf)) for k, v in init.items()}
    rec_ulp, _, _ = _oracle_replay(init_ulp, data, epochs, 32, rows_n)
    want_ulp = rec_ulp[want_rows]
    wscale = float(np.abs(want[:, :D]).max())
    wfloor = float(np.abs(want_ulp[:, :D] - want[:, :D]).max())
    werr = np.abs(got[:, :D] - want[:, :D])
    assert float(werr.max()) <= max(1e-5 * wscale, 4.0 * wfloor), (float(werr.max()), wscale, wfloor)
    g2 = want[:, D + 2:D + 4]
    gfloor = float(np.abs(want_ulp[:, D + 2:D + 4] - g2).max())
    gerr = float(np.abs(got[:, D + 2:D + 4] - g2).max())
    assert gerr <= max(1e-5 * float(g2.max()), 4.0 * gfloor), (gerr, float(g2.max()), gfloor)
    print("gpubox checkpoint vs oracles: weights err %.3e (1e-5 of scale %.3e, measured floor %.3e); g2sum err %.3e (floor %.3e)"
          % (float(werr.max()), 1e-5 * wscale, wfloor, gerr, gfloor))
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from helpers import assert_adam_weights_close
    steps = epochs * 3                                                          # 80 samples at batch 32: 3 steps per pass
    for i in range(5):
        assert_adam_weights_close(z["dense.%d" % (2 * i)], mlp_w[i], lr=1e-3, steps=steps, err_msg="weight %d" % i)
        assert_adam_weights_close(z["dense.%d" % (2 * i + 1)], mlp_b[i], lr=1e-3, steps=steps, err_msg="bias %d" % i)


def test_static_gpubox_trainer_checkpoint_equals_the_oracles_cpu_backend(tmp_path):
    _check_against_oracles(tmp_path, gpu=False)


@pytest.mark.gpu
def test_static_gpubox_trainer_checkpoint_equals_the_oracles_on_the_hip_kernels(tmp_path, engine_lib):
    _check_against_oracles(tmp_path, gpu=True)


# ------------------------------------------------------------------------------------------------------------------
# integration/dnn_net.patch: the gpubox branch of dnn/net.py calling the compiled custom operator rec_ps_pull (its gradient
# operator is the accessor's push) instead of static.nn.sparse_embedding + continuous_value_model — under the SAME
# unmodified tools/static_gpubox_trainer.py (patched == unpatched == oracle)
def _patched_equals_unpatched(tmp_path, gpu, tol):
    epochs = 2
    env = {"REC_GPUBOX_TABLE_ROWS": "20011", "REC_COMPAT_SEED": "99"}
    loga, outa = _run(tmp_path, gpu, epochs, name="plain", extra_env=env)
    logb, outb = _run(tmp_path, gpu, epochs, name="patched", extra_env=env, tree=PATCHED)
    assert "rec_ps_pull" not in loga and "custom operator rec_ps_pull" in logb and "Run Success, Exit." in logb
    a = np.load(os.path.join(str(outa), str(epochs - 1), "rec_gpubox.npz"))
    b = np.load(os.path.join(str(outb), str(epochs - 1), "rec_gpubox.npz"))
    assert sorted(a.files) == sorted(b.files)                      # same dense tensors (the anchor is no parameter of the
    assert np.array_equal(a["table.embedding.rows"], b["table.embedding.rows"])          # program), same table, same keys
    ra, rb = a["table.embedding.records"], b["table.embedding.records"]
    D = 9
    for col, what in ((D, "show"), (D + 1, "click"), (D + 4, "state"), (D + 6, "unseen_days")):
        assert np.array_equal(ra[:, col], rb[:, col]), what
    np.testing.assert_allclose(rb, ra, rtol=0, atol=tol * float(np.abs(ra[:, :D]).max()), err_msg="records")
    for k in [k for k in a.files if k.startswith("dense.")]:
        np.testing.assert_allclose(b[k], a[k], rtol=0, atol=tol * max(1.0, float(np.abs(a[k]).max())), err_msg=k)
    ea = re.findall(r"Epoch: \d+, .* auc: ([0-9.]+)", loga)
    eb = re.findall(r"Epoch: \d+, .* auc: ([0-9.]+)", logb)
    assert len(ea) == epochs and [round(float(x), 5) for x in ea] == [round(float(x), 5) for x in eb]


@needs_patched
def test_patched_dnn_net_trains_like_the_unpatched_one_cpu_backend(tmp_path):
    _patched_equals_unpatched(tmp_path, gpu=False, tol=1e-6)


@needs_patched
def test_patched_dnn_net_checkpoint_equals_the_oracles_cpu_backend(tmp_path):
    _check_against_oracles(tmp_path, gpu=False, tree=PATCHED)


@needs_patched
@pytest.mark.gpu
def test_patched_dnn_net_trains_like_the_unpatched_one_on_the_hip_kernels(tmp_path, engine_lib):
    # the same kernels on both sides (rec_feasign_rows, rec_emb_gather, rec_ids_group, rec_ps_push_rows): the pull is
    # bit-identical, the push sees the 26 slots' gradient rows in the same order
    _patched_equals_unpatched(tmp_path, gpu=True, tol=1e-6)


@needs_patched
@pytest.mark.gpu
def test_patched_dnn_net_checkpoint_equals_the_oracles_on_the_hip_kernels(tmp_path, engine_lib):
    _check_against_oracles(tmp_path, gpu=True, tree=PATCHED)
