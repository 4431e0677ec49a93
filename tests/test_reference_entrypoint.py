"""The reference's OWN driver, unmodified (row N1): `tools/trainer.py -m models/rank/deepfm/config.yaml` executed through
paddlerec_amd.run_reference over the product `paddle` compat namespace (paddlerec_amd/compat).  The loss lines the
reference prints and the checkpoint it writes must equal the oracle's trajectory from the same initial parameters
(dygraph Adam, lazy_mode=False: every row decays each step).
  * not gpu: the operator backend is the oracle-backed stand-in (REC_COMPAT_KERNELS) — the host logic of the namespace;
  * -m gpu : `runner.use_gpu=True`, the HIP kernels behind the C-ABI carry the lookup, GEMMs, merge and optimizers.
The reference tree is /root/reference in the build container; on the GPU box (no /root/reference) it is the byte copy
of the needed files that oracle/make_ref_tree.py stages under oracle/_ref/PaddleRec (git-ignored, travels with the
snapshot)."""
import os
import pickle
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO

STAGED = os.path.join(REPO, "oracle", "_ref", "PaddleRec")
REF = os.environ.get("PADDLEREC_REF") or (STAGED if os.path.isdir(os.path.join(STAGED, "tools")) else "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "tools")),
                                reason="reference tree not present (run oracle/make_ref_tree.py in the build container)")

INIT_SCRIPT = r"""
import os, sys, pickle
sys.path.insert(0, os.path.join(%(repo)r, "paddlerec_amd", "compat"))
sys.path.insert(0, %(repo)r)
sys.path.insert(0, os.path.join(%(ref)r, "models", "rank", "deepfm"))
import paddle
paddle.seed(12345)
paddle.set_device(%(dev)r)
import net
m = net.DeepFMLayer(1000001, 9, 13, 26, [512, 256, 128, 32])
pickle.dump({k: v.detach().numpy() for k, v in m.state_dict().items()}, open(sys.argv[1], "wb"))
"""


def _env(gpu=False):
    env = dict(os.environ, OMP_NUM_THREADS="4", PYTHONDONTWRITEBYTECODE="1")      # nothing is written next to the reference's files
    if not gpu:
        env["REC_COMPAT_KERNELS"] = "cpu_kernels"
    else:
        env.pop("REC_COMPAT_KERNELS", None)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(REPO, "tests"), REPO, env.get("PYTHONPATH", "")])
    return env


def test_staged_tree_is_a_byte_copy():
    """oracle/_ref/PaddleRec (what the GPU box runs) holds unmodified copies of the reference's files."""
    if not (os.path.isdir("/root/reference/tools") and os.path.isdir(STAGED)):
        pytest.skip("needs both /root/reference and the staged tree")
    n = 0
    for root, _, files in os.walk(STAGED):
        for f in files:
            if f in ("STAGED_FROM", "tmp.txt", "train_result_dict.txt") or f.endswith((".pyc", ".prototxt")) \
                    or "__pycache__" in root:
                continue
            rel = os.path.relpath(os.path.join(root, f), STAGED)
            if rel.startswith("output_model") or "output_model" in rel:
                continue
            with open(os.path.join(root, f), "rb") as a, open(os.path.join("/root/reference", rel), "rb") as b:
                assert a.read() == b.read(), rel
            n += 1
    assert n >= 30


def test_reference_trainer_runs_unmodified_and_matches_oracle(tmp_path):
    _run_trainer_and_check(tmp_path, gpu=False)


@pytest.mark.gpu
def test_reference_trainer_unmodified_on_the_hip_kernels(tmp_path, engine_lib):
    """Row N1 on the GPU: the reference's own tools/trainer.py, its loop, reader, dygraph_model.py and net.py, with
    `runner.use_gpu=True` — every paddle op of the step executes on the HIP kernels through the compat namespace."""
    _run_trainer_and_check(tmp_path, gpu=True)


def _run_trainer_and_check(tmp_path, gpu, tree=None):
    """tree: the PaddleRec tree whose tools/trainer.py + deepfm plugin run (default: the unmodified one; the custom-op
    tests pass the copy patched by integration/*.patch — initial parameters still come from the unmodified net.py)."""
    from oracle import deepfm_ref as R
    out_dir = tmp_path / "ckpt"
    tree = tree or REF
    cmd = [sys.executable, "-m", "paddlerec_amd.run_reference", os.path.join(tree, "tools", "trainer.py"),
           "-m", os.path.join(tree, "models", "rank", "deepfm", "config.yaml"),
           "-o", "runner.epochs=1", "runner.print_interval=5", "runner.model_save_path=%s" % out_dir,
           "runner.use_gpu=%s" % ("True" if gpu else "False")]
    r = subprocess.run(cmd, cwd=tree, env=_env(gpu), capture_output=True, text=True, timeout=900)
    log = r.stdout + r.stderr
    assert r.returncode == 0, log[-3000:]
    printed = [(int(m.group(1)), float(m.group(2))) for m in
               re.finditer(r"batch_id: (\d+), auc:[0-9.]+, loss:\s*([0-9.eE+-]+),", log)]
    assert len(printed) == 8 and printed[0][0] == 0 and printed[-1][0] == 35, log[-2000:]       # 40 batches of 2
    ips = re.findall(r"ips: ([0-9.]+) ins/s", log)
    assert ips, "the reference's own ips line is missing"
    # the same initial parameters: the reference's net.py constructed over the compat namespace with the trainer's seed
    init_file = tmp_path / "init.pkl"
    r2 = subprocess.run([sys.executable, "-c", INIT_SCRIPT % dict(repo=REPO, ref=REF, dev="gpu" if gpu else "cpu"),
                         str(init_file)], env=_env(gpu), capture_output=True, text=True, timeout=300)
    assert r2.returncode == 0, r2.stderr[-2000:]
    sd = pickle.load(open(init_file, "rb"))
    n_mlp = 5
    p = {"W": sd["fm.embedding.weight"].copy(), "W1": sd["fm.embedding_one.weight"].copy(),
         "dense_w": sd["fm.dense_w"].copy(), "dense_w_one": sd["fm.dense_w_one"].copy(),
         "mlp_w": [sd["dnn.linear_%d.weight" % i].copy() for i in range(n_mlp)],
         "mlp_b": [sd["dnn.linear_%d.bias" % i].copy() for i in range(n_mlp)]}
    lines = open(os.path.join(REF, "models/rank/deepfm/data/sample_data/train/sample_train.txt")).read().strip().split("\n")
    parsed = [R.parse_slot_line(ln) for ln in lines]
    lab = np.asarray([a for a, _, _ in parsed], np.int64).reshape(-1, 1)
    ids = np.stack([b for _, b, _ in parsed])
    dense = np.stack([c for _, _, c in parsed])
    st = {k: np.zeros_like(p[t]) for k, t in (("mW", "W"), ("vW", "W"), ("mW1", "W1"), ("vW1", "W1"))}
    dstate, want = {}, {}
    for step in range(40):
        lo = step * 2
        o = R.deepfm_loss_and_grads(ids[lo:lo + 2], dense[lo:lo + 2], lab[lo:lo + 2], p)
        want[step] = float(o["loss"])
        t = step + 1
        uniq, merged, _ = R.merge_rows(o["rows"], o["row_valid"], o["row_grad"])
        R.adam_update_dense_equivalent(p["W"], st["mW"], st["vW"], uniq, merged, t, lr=0.001)
        uniq1, merged1, _ = R.merge_rows(o["rows"], o["row_valid"], o["row_grad1"])
        R.adam_update_dense_equivalent(p["W1"], st["mW1"], st["vW1"], uniq1, merged1, t, lr=0.001)
        pairs = [("dense_w", o["d_dense_w"]), ("dense_w_one", o["d_dense_w_one"])]
        for i in range(n_mlp):
            pairs += [(("mlp_w", i), o["mlp_dw"][i]), (("mlp_b", i), o["mlp_db"][i])]
        for key, gr in pairs:
            arr = p[key] if not isinstance(key, tuple) else p[key[0]][key[1]]
            mm, vv = dstate.setdefault(key, (np.zeros_like(arr), np.zeros_like(arr)))
            R.adam_update(arr, mm, vv, gr.reshape(arr.shape).astype(arr.dtype), t, lr=0.001)
    for batch_id, loss in printed:
        np.testing.assert_allclose(loss, want[batch_id], rtol=2e-4, atol=1e-6)       # the log prints 6-8 digits
    saved = pickle.load(open(out_dir / "0" / "rec.pdparams", "rb"))
    np.testing.assert_allclose(saved["dnn.linear_4.weight"], p["mlp_w"][4], rtol=1e-3, atol=1e-5)
    touched = np.unique(ids)
    np.testing.assert_allclose(saved["fm.embedding.weight"][touched], p["W"][touched], rtol=1e-3, atol=2e-5)
    assert os.path.exists(out_dir / "0" / "rec.pdopt")


def _collective_equals_single(tmp_path, gpu):
    """runner.use_fleet=True (tools/trainer.py:112-118,212: fleet.init(is_collective=True), distributed_optimizer,
    distributed_model) — VERDICT r03 "missing" 3: the compat fleet raised NotImplementedError.  Two ranks of the
    UNMODIFIED trainer (torch.distributed.run, gloo: two CPU processes / two processes sharing cuda:0), each reading its own
    file as the reference's reader splits them (criteo_reader.py:30-42), batch 2 per rank, gradients averaged over the
    ranks — against ONE run without fleet on the interleaved file at batch 4: the same four samples per step, so the
    checkpoint rank 0 writes must equal the single run's."""
    import socket
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from helpers import assert_adam_weights_close
    lines = open(os.path.join(REF, "models/rank/deepfm/data/sample_data/train/sample_train.txt")).read().strip().split("\n")
    two, one = tmp_path / "two", tmp_path / "one"
    two.mkdir()
    one.mkdir()
    (two / "part-0").write_text("\n".join(lines[:40]) + "\n")
    (two / "part-1").write_text("\n".join(lines[40:80]) + "\n")
    inter = []
    for k in range(20):
        inter += lines[2 * k:2 * k + 2] + lines[40 + 2 * k:40 + 2 * k + 2]
    (one / "part-0").write_text("\n".join(inter) + "\n")
    common = ["-m", os.path.join(REF, "models", "rank", "deepfm", "config.yaml"), "-o", "runner.epochs=1",
              "runner.print_interval=5", "runner.use_gpu=%s" % ("True" if gpu else "False")]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd2 = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
            "127.0.0.1", "--master-port", str(port), "-m", "paddlerec_amd.run_reference",
            os.path.join(REF, "tools", "trainer.py")] + common + \
           ["runner.use_fleet=True", "runner.train_data_dir=%s" % two, "runner.model_save_path=%s" % (tmp_path / "ck2")]
    r = subprocess.run(cmd2, cwd=REF, env=_env(gpu), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    cmd1 = [sys.executable, "-m", "paddlerec_amd.run_reference", os.path.join(REF, "tools", "trainer.py")] + common + \
           ["runner.train_batch_size=4", "runner.train_data_dir=%s" % one, "runner.model_save_path=%s" % (tmp_path / "ck1")]
    r1 = subprocess.run(cmd1, cwd=REF, env=_env(gpu), capture_output=True, text=True, timeout=900)
    assert r1.returncode == 0, (r1.stdout + r1.stderr)[-4000:]
    a = pickle.load(open(tmp_path / "ck2" / "0" / "rec.pdparams", "rb"))
    b = pickle.load(open(tmp_path / "ck1" / "0" / "rec.pdparams", "rb"))
    assert set(a) == set(b) and len(a) >= 14
    for k in b:
        assert_adam_weights_close(a[k], b[k], lr=1e-3, steps=20, err_msg=k)
    assert not np.array_equal(b["fm.embedding.weight"], np.zeros_like(b["fm.embedding.weight"]))


def test_reference_trainer_use_fleet_two_ranks_equal_one_run_cpu_backend(tmp_path):
    _collective_equals_single(tmp_path, gpu=False)


@pytest.mark.gpu
def test_reference_trainer_use_fleet_two_ranks_on_one_gpu(tmp_path, engine_lib):
    _collective_equals_single(tmp_path, gpu=True)
