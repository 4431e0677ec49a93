"""Train / infer loops (paddlerec_amd/trainer.py; reference: tools/trainer.py:40-223, tools/infer.py) on
BASELINE configs[0]: DeepFM, the reference's own 80-line Criteo sample, small batches.

Host-logic tests run here on the CPU with the oracle-backed operator backend (tests/cpu_kernels.py) injected
through `kernels=` — config handling, the reader, the epoch loop, AUC accumulation, checkpoints per epoch and
the per-epoch infer loop are exactly the product code; the `-m gpu` test runs the same thing on the HIP kernels."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from helpers import OracleTrainer

YAML = """
runner:
  train_data_dir: "data/train"
  train_reader_path: "criteo_reader" # importlib format
  use_gpu: False
  use_auc: True
  train_batch_size: 16
  epochs: 2
  print_interval: 2
  model_save_path: "{out}"
  test_data_dir: "data/train"
  infer_batch_size: 20
  infer_load_path: "{out}"
  infer_start_epoch: 0
  infer_end_epoch: 2
hyper_parameters:
  optimizer:
    class: Adam
    learning_rate: 0.01
    strategy: async
    lazy_mode: True     # engine key: the static-graph optimizer (deepfm/static_model.py:101-107); default False = dygraph Adam
  sparse_inputs_slots: 27
  sparse_feature_number: 1000001
  sparse_feature_dim: 9
  dense_input_dim: 13
  fc_sizes: [16, 8]
  distributed_embedding: 0
"""


def _sample_lines(n=80, seed=7):
    """Slot-text lines in the format of models/rank/deepfm/data/sample_data/train/sample_train.txt (the golden
    fixture holds its first 6 lines; the other 74 are synthetic: a small id range so that rows repeat across
    batches and Adam's moments matter, a few missing slots -> padding id 0, criteo_reader.py:80-91)."""
    rng = np.random.default_rng(seed)
    lines = open(os.path.join(GOLDEN, "criteo_slot_sample.txt")).read().strip().split("\n")
    while len(lines) < n:
        parts = ["click:%d" % int(rng.random() < 0.3)]
        parts += ["dense_feature:%s" % repr(round(float(rng.random()), 6)) for _ in range(13)]
        for slot in range(1, 27):
            if rng.random() < 0.04:
                continue
            parts.append("%d:%d" % (slot, int(rng.integers(1, 400)) + 1000 * slot))
        lines.append(" ".join(parts))
    return lines


@pytest.fixture()
def workdir(tmp_path):
    d = tmp_path / "models" / "rank" / "deepfm"
    (d / "data" / "train").mkdir(parents=True)
    (d / "data" / "train" / "sample_train.txt").write_text("\n".join(_sample_lines()) + "\n")
    (d / "config.yaml").write_text(YAML.format(out=str(tmp_path / "output_model_deepfm")))
    return d


def test_load_yaml_flattens_and_overrides(workdir):
    from paddlerec_amd import trainer
    cfg = trainer.load_yaml(str(workdir / "config.yaml"),
                            ["runner.epochs=5", "runner.use_auc=false", "hyper_parameters.optimizer.learning_rate=0.5",
                             "runner.model_save_path=elsewhere"])
    assert cfg["runner.train_batch_size"] == 16 and cfg["hyper_parameters.fc_sizes"] == [16, 8]
    assert cfg["hyper_parameters.optimizer.class"] == "Adam"
    assert cfg["runner.epochs"] == 5 and cfg["runner.use_auc"] is False          # coerced to the old value's type
    assert cfg["hyper_parameters.optimizer.learning_rate"] == 0.5 and cfg["runner.model_save_path"] == "elsewhere"
    assert cfg["config_abs_dir"] == str(workdir)
    assert trainer.guess_model(str(workdir / "config.yaml")) == "deepfm"
    with pytest.raises(ValueError):
        trainer.create_data_loader({"runner.train_data_dir": "nope", "config_abs_dir": str(workdir),
                                    "runner.train_batch_size": 2}, "deepfm", "cpu")


def _oracle_params(sd, n_mlp):
    p = {"W": sd["fm.embedding.weight"], "W1": sd["fm.embedding_one.weight"], "dense_w": sd["fm.dense_w"],
         "dense_w_one": sd["fm.dense_w_one"], "mlp_w": [sd["dnn.linear_%d.weight" % i] for i in range(n_mlp)],
         "mlp_b": [sd["dnn.linear_%d.bias" % i] for i in range(n_mlp)]}
    return p


def _hist_auc(R, preds, labels):
    pos, neg = np.zeros(4096, np.int64), np.zeros(4096, np.int64)
    for p, t in zip(preds, labels):
        dp, dn = R.auc_histogram(p, t)
        pos += dp
        neg += dn
    return R.auc_from_buckets(pos, neg)


def _run(workdir, device, kernels, tol):
    """tol = (loss rtol, AUC atol, max |param diff|, mean |param diff|)"""
    l_rtol, auc_atol, p_max, p_mean = tol
    from oracle import deepfm_ref as R
    from paddlerec_amd import trainer
    from paddlerec_amd.deepfm import DygraphModel
    cfg = trainer.load_yaml(str(workdir / "config.yaml"))
    # the same initial parameters train() will draw (runner.seed default 12345, trainer.py:83-84)
    torch.manual_seed(12345)
    m0 = DygraphModel().create_model(cfg, device, **({"kernels": kernels} if kernels else {}))
    sd0 = {k: v.detach().cpu().numpy().copy() for k, v in m0.state_dict().items()}
    del m0
    summaries, model = trainer.train(cfg, "deepfm", device, kernels)
    # oracle: the same 2 x 5 batches of 16 in file order (drop_last: 80 = 5 * 16)
    parsed = [R.parse_slot_line(ln) for ln in _sample_lines()]
    lab = np.asarray([a for a, _, _ in parsed], np.int64).reshape(-1, 1)
    ids = np.stack([b for _, b, _ in parsed])
    dense = np.stack([c for _, _, c in parsed])
    tr = OracleTrainer(_oracle_params(sd0, 3), lr=0.01)
    snaps, aucs, losses = [], [], []
    for _ in range(2):
        preds = []
        for lo in range(0, 80, 16):
            loss, pred = tr.train_step(ids[lo:lo + 16], dense[lo:lo + 16], lab[lo:lo + 16])
            preds.append(pred)
        losses.append(float(loss))
        aucs.append(_hist_auc(R, preds, [lab[lo:lo + 16] for lo in range(0, 80, 16)]))
        snaps.append({k: (v.copy() if not isinstance(v, list) else [x.copy() for x in v]) for k, v in tr.p.items()})
    assert [s["epoch"] for s in summaries] == [0, 1] and all(s["batches"] == 5 and s["samples"] == 80 for s in summaries)
    for s, ol, oa in zip(summaries, losses, aucs):
        np.testing.assert_allclose(s["loss"], ol, rtol=l_rtol)
        np.testing.assert_allclose(s["auc"], oa, rtol=1e-9, atol=auc_atol)   # integer buckets: equal unless a pred sits on an edge
        assert os.path.exists(os.path.join(s["model_dir"], "rec.pdparams"))
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    touched = np.unique(ids)
    for got, want in ((sd["fm.embedding.weight"][touched], tr.p["W"][touched]),
                      (sd["dnn.linear_0.weight"], tr.p["mlp_w"][0])):
        d = np.abs(got - want)          # 10 Adam steps at lr 1e-2: parameters moved by up to 0.1
        assert d.max() <= p_max and d.mean() <= p_mean, (d.max(), d.mean())
    untouched = np.setdiff1d(np.arange(0, 5000), touched)
    assert np.array_equal(sd["fm.embedding.weight"][untouched], sd0["fm.embedding.weight"][untouched])   # lazy rows
    # infer loop: checkpoint of each epoch -> AUC over the test set in batches of 20
    res = trainer.infer(cfg, "deepfm", device, kernels)
    assert [r["epoch"] for r in res] == [0, 1] and all(r["batches"] == 4 and r["samples"] == 80 for r in res)
    for r, snap in zip(res, snaps):
        preds = [R.deepfm_forward(ids[lo:lo + 20], dense[lo:lo + 20], snap)[0] for lo in range(0, 80, 20)]
        np.testing.assert_allclose(r["auc"], _hist_auc(R, preds, [lab[lo:lo + 20] for lo in range(0, 80, 20)]),
                                   rtol=1e-9, atol=auc_atol)
    # resume: model_init_path + last_epoch continue from the epoch-0 checkpoint and reproduce epoch 1
    cfg2 = dict(cfg)
    cfg2["runner.model_init_path"] = summaries[0]["model_dir"]
    cfg2["last_epoch"] = 0
    cfg2["runner.model_save_path"] = str(workdir / "resumed")
    s2, _ = trainer.train(cfg2, "deepfm", device, kernels)
    assert [s["epoch"] for s in s2] == [1]
    np.testing.assert_allclose(s2[0]["loss"], summaries[1]["loss"], rtol=1e-6)
    np.testing.assert_allclose(s2[0]["auc"], summaries[1]["auc"], rtol=1e-9)


def test_train_infer_resume_cpu_backend(workdir):
    import cpu_kernels
    _run(workdir, "cpu", cpu_kernels, (2e-5, 1e-12, 1e-4, 1e-6))


@pytest.mark.gpu
def test_train_infer_resume_gpu(workdir, engine_lib):
    # fp32 kernels vs the NumPy oracle over 10 optimizer steps: a prediction next to a bucket edge may move one
    # AUC bucket, an entry with an ~eps-sized gradient may take a different Adam step (see test_dcn_v2_gpu.py)
    _run(workdir, "cuda", None, (1e-3, 5e-3, 2e-2, 1e-4))


REF_RANK = "/root/reference/models/rank"


@pytest.mark.skipif(not os.path.isdir(REF_RANK), reason="reference tree not mounted (only in the build container)")
@pytest.mark.parametrize("model,samples,batches", [("deepfm", 80, 40), ("fm", None, None), ("wide_deep", None, None),
                                                    ("dnn", None, None), ("dcn_v2", None, None), ("din", None, None),
                                                    ("xdeepfm", None, None), ("dlrm", None, None)])
def test_reference_yaml_and_sample_data_run_unchanged(model, samples, batches, tmp_path):
    """BASELINE configs[0] and its siblings: the reference's OWN models/rank/<model>/config.yaml and sample data
    directory drive the loops as they are (deepfm: bs 2, 80 lines, D 9, fc 512-256-128-32; dcn_v2: D 40, CrossNetMix
    with 4 experts; din: bs 32, 100 lines) — only the output directory and the number of epochs are redirected."""
    import cpu_kernels
    from paddlerec_amd import trainer
    yaml_path = os.path.join(REF_RANK, model, "config.yaml")
    cfg = trainer.load_yaml(yaml_path, ["runner.epochs=1", "runner.model_save_path=" + str(tmp_path / "out"),
                                        "runner.infer_load_path=" + str(tmp_path / "out"),
                                        "runner.infer_start_epoch=0", "runner.infer_end_epoch=1"])
    assert trainer.guess_model(yaml_path) == model
    s, net = trainer.train(cfg, model, "cpu", cpu_kernels)
    assert len(s) == 1 and np.isfinite(s[0]["loss"]) and 0.0 <= s[0]["auc"] <= 1.0
    if samples is not None:
        assert s[0]["samples"] == samples and s[0]["batches"] == batches
    else:
        assert s[0]["samples"] > 0 and s[0]["samples"] % cfg["runner.train_batch_size"] == 0
    assert int(net.status.item()) == 0                          # no id of the sample files is out of range
    r = trainer.infer(cfg, model, "cpu", cpu_kernels)
    assert r[0]["samples"] > 0 and 0.0 <= r[0]["auc"] <= 1.0


def _other_mirrors(model, tmp_path, device, kernels):
    import shutil
    from paddlerec_amd import trainer
    d = tmp_path / "models" / "rank" / model
    (d / "data").mkdir(parents=True)
    src = "criteo_slot_sample.txt" if model == "dcn_v2" else "din_sample.txt"
    shutil.copy(os.path.join(GOLDEN, src), d / "data" / "part-0")
    cfg = {"config_abs_dir": str(d), "runner.train_data_dir": "data", "runner.test_data_dir": "data",
           "runner.train_batch_size": 2 if model == "dcn_v2" else 8, "runner.infer_batch_size": 2 if model == "dcn_v2" else 8,
           "runner.epochs": 2, "runner.print_interval": 1, "runner.use_auc": True,
           "runner.model_save_path": str(tmp_path / "out"), "runner.infer_load_path": str(tmp_path / "out"),
           "runner.infer_start_epoch": 0, "runner.infer_end_epoch": 2,
           "hyper_parameters.sparse_feature_number": 1000001, "hyper_parameters.sparse_feature_dim": 8,
           "hyper_parameters.dense_input_dim": 13, "hyper_parameters.sparse_inputs_slots": 27,
           "hyper_parameters.fc_sizes": [32, 16], "hyper_parameters.cross_num": 2, "hyper_parameters.is_Stacked": True,
           "hyper_parameters.use_low_rank_mixture": False, "hyper_parameters.optimizer.learning_rate": 0.001,
           "hyper_parameters.item_emb_size": 64, "hyper_parameters.cat_emb_size": 64,
           "hyper_parameters.item_count": 63001, "hyper_parameters.cat_count": 801,
           "hyper_parameters.optimizer.learning_rate_base_lr": 0.85}
    s, net = trainer.train(cfg, model, device, kernels)
    assert [x["epoch"] for x in s] == [0, 1] and all(np.isfinite(x["loss"]) and x["batches"] >= 3 for x in s)
    assert all(os.path.exists(os.path.join(x["model_dir"], "rec.pdparams")) for x in s)
    r = trainer.infer(cfg, model, device, kernels)
    assert [x["epoch"] for x in r] == [0, 1] and all(0.0 <= x["auc"] <= 1.0 and x["samples"] > 0 for x in r)
    # the epoch-1 checkpoint holds the trained parameters: a fresh model loaded from it predicts like `net`
    from paddlerec_amd import checkpoint
    dm = trainer._dygraph_model(model)
    fresh = dm.create_model(cfg, device, **({"kernels": kernels} if kernels is not None else {}))
    checkpoint.load_model(s[-1]["model_dir"], fresh, load_optimizer=False)
    for k, v in net.state_dict().items():
        assert torch.equal(v.detach().cpu(), fresh.state_dict()[k].detach().cpu()), k


@pytest.mark.parametrize("model", ["dcn_v2", "din"])
def test_other_mirrors_through_the_loops_cpu_backend(model, tmp_path):
    """DCN-v2 (slot text, log1p dense, ClipGradByGlobalNorm) and DIN (dinReader format, SGD) through the same
    train / checkpoint / infer loops on the reference's own sample lines (tests/golden), host logic only."""
    import cpu_kernels
    _other_mirrors(model, tmp_path, "cpu", cpu_kernels)


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["dcn_v2", "din"])
def test_other_mirrors_through_the_loops_gpu(model, tmp_path, engine_lib):
    _other_mirrors(model, tmp_path, "cuda", None)


def test_collective_mode_two_ranks_gloo(tmp_path):
    """tools/trainer.py `use_fleet` mode on the engine: 2 processes (gloo), the data files split over the ranks, tables
    row-sharded, routing prefetched a step ahead, global loss / AUC, one checkpoint shard per rank, sharded infer —
    equal to ONE oracle run on the concatenated batches."""
    import socket
    import subprocess
    import sys
    from conftest import REPO
    from oracle import deepfm_ref as R
    world, B = 2, 8
    d = tmp_path / "deepfm"
    (d / "data" / "train").mkdir(parents=True)
    lines = _sample_lines(96)
    for r in range(world):                                         # rank r reads file r (criteo_reader.py:30-43)
        mine = lines[r * 40:(r + 1) * 40] + (lines[80:] if r == 1 else [])     # rank 1 holds 2 more batches: unused,
        (d / "data" / "train" / ("part-%d" % r)).write_text("\n".join(mine) + "\n")   # every rank runs min = 5 steps
    lines = lines[:80]
    (d / "config.yaml").write_text(YAML.format(out=str(tmp_path / "out")).replace("train_batch_size: 16", "train_batch_size: %d" % B)
                                   .replace("infer_batch_size: 20", "infer_batch_size: 10"))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    worker = os.path.join(REPO, "tests", "_trainer_dist_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), port, str(d)], env=dict(os.environ, OMP_NUM_THREADS="2"),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = [p.communicate(timeout=600)[0].decode("utf-8", "replace") for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, outs[r][-3000:])
    ranks = [dict(np.load(str(d / ("rank%d.npz" % r)))) for r in range(world)]
    # every rank issued the same sequence of collectives (one communicator: any divergence is a hang on real GPUs)
    assert ranks[0]["trace"].tolist() == ranks[1]["trace"].tolist() and len(ranks[0]["trace"]) > 50
    # both ranks report the same GLOBAL numbers
    for k in ("loss", "auc", "samples", "batches", "infer_auc", "W", "mlp_w0"):
        assert np.array_equal(ranks[0][k], ranks[1][k]), k
    assert ranks[0]["samples"].tolist() == [80, 80] and ranks[0]["batches"].tolist() == [5, 5]
    assert ranks[0]["shard_files"].all() and ranks[1]["shard_files"].all()
    # oracle: one unsharded run, step i = rank 0's batch i followed by rank 1's batch i
    g = ranks[0]
    p = {"W": g["init.W"].copy(), "W1": g["init.W1"].copy(), "dense_w": g["init.fm.dense_w"].copy(),
         "dense_w_one": g["init.fm.dense_w_one"].copy(),
         "mlp_w": [g["init.dnn.linear_%d.weight" % i].copy() for i in range(3)],
         "mlp_b": [g["init.dnn.linear_%d.bias" % i].copy() for i in range(3)]}
    parsed = [R.parse_slot_line(ln) for ln in lines]
    lab = np.asarray([a for a, _, _ in parsed], np.int64).reshape(-1, 1)
    ids = np.stack([b for _, b, _ in parsed])
    dense = np.stack([c for _, _, c in parsed])
    tr = OracleTrainer(p, lr=0.01)
    losses, aucs = [], []
    for _ in range(2):
        preds, labs = [], []
        for i in range(5):
            sel = np.concatenate([np.arange(r * 40 + i * B, r * 40 + (i + 1) * B) for r in range(world)])
            loss, pred = tr.train_step(ids[sel], dense[sel], lab[sel])
            preds.append(pred)
            labs.append(lab[sel])
        losses.append(float(loss))
        aucs.append(_hist_auc(R, preds, labs))
    np.testing.assert_allclose(g["loss"], losses, rtol=2e-5)
    np.testing.assert_allclose(g["auc"], aucs, rtol=1e-9, atol=1e-12)
    touched = np.unique(ids)
    assert np.abs(g["W"][touched] - tr.p["W"][touched]).max() <= 1e-4
    assert np.abs(g["mlp_w0"] - tr.p["mlp_w"][0]).max() <= 1e-4
    assert g["infer_samples"].tolist() == [80, 80]
    final_pred = R.deepfm_forward(ids, dense, tr.p)[0]
    np.testing.assert_allclose(g["infer_auc"][-1], _hist_auc(R, [final_pred], [lab]), rtol=1e-9, atol=1e-12)
