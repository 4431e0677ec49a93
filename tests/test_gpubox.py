"""The gpubox pass loop (paddlerec_amd/gpubox.py; reference: tools/static_gpubox_trainer.py:85-260 with the PSGPU
begin_pass / train_from_dataset / end_pass surface) on the reference's own multi-value fixture (the first lines of
models/rank/slot_dnn/data/demo_10) with the accessor of slot_dnn/config_online.yaml:57-89.

Checked against a straight replay of the same passes with the oracle (oracle/slot_dnn_ref.py + oracle/ps_ref.py):
per-pass loss, AUC buckets, the table after every end_pass (shrink: decay + delete), the pass checkpoint round trip."""
import os
import shutil

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import deepfm_ref as R
from oracle import ps_ref
from oracle import slot_dnn_ref as M

SLOTS, N, D = 300, 20011, 9


def _config(tmp_path, epochs=2):
    d = tmp_path / "slot_dnn"
    (d / "data").mkdir(parents=True, exist_ok=True)
    shutil.copy(os.path.join(GOLDEN, "slot_dnn_demo_4.txt"), d / "data" / "part-0")
    acc = {"embedx_threshold": 0.5,      # low enough that some features grow their embedx within two passes
           "embedx_sgd_param": {"adagrad": {"learning_rate": 0.05, "initial_g2sum": 3.0, "initial_range": 1e-2,
                                            "weight_bounds": [-10.0, 10.0]}},
           "ctr_accessor_param": {"nonclk_coeff": 0.1, "click_coeff": 1.0, "show_click_decay_rate": 0.98,
                                  "delete_threshold": 0.15}}
    return {"config_abs_dir": str(d), "runner.train_data_dir": "data", "runner.train_batch_size": 2,
            "runner.epochs": epochs, "runner.use_auc": True, "runner.model_save_path": str(tmp_path / "out"),
            "hyper_parameters.dict_dim": N, "hyper_parameters.emb_dim": D, "hyper_parameters.slot_num": SLOTS,
            "hyper_parameters.layer_sizes": [16, 8], "hyper_parameters.optimizer.learning_rate": 1e-3,
            "table_parameters.embedding.accessor": acc}


def _run(tmp_path, device, kernels, loss_rtol=1e-5):
    from paddlerec_amd import gpubox, reader
    cfg = _config(tmp_path)
    torch.manual_seed(7)
    main = gpubox.Main(cfg, device, kernels)
    with pytest.raises(RuntimeError):
        gpubox.PSGPU(main.k).begin_pass()                       # begin_pass before init_gpu_ps / bind
    main.network()
    mw = [w.cpu().numpy().copy() for w in main.net.mlp_w]      # the initial dense weights the replay starts from
    mb = [b.cpu().numpy().copy() for b in main.net.mlp_b]
    res = main.run_worker()
    net, L = main.net, main.net.table.layout
    assert len(res["loss"]) == 2 and all(np.isfinite(x) for x in res["loss"]) and len(res["auc"]) == 2
    assert main.PSGPU.passes == 2 and main.PSGPU.device_pass is None
    # ---- replay with the oracle: the same initial dense weights, the same accessor, the same passes
    acc = dict(lr=0.05, initial_g2sum=3.0, bounds=(-10.0, 10.0), initial_range=1e-2, embedx_threshold=0.5,
               nonclk_coeff=0.1, click_coeff=1.0, seed=net.table.accessor.seed)
    lay = dict(embed_off=L.embed_off, embedx_off=L.embedx_off, embedx_dim=L.embedx_dim, stat_off=L.stat_off)
    def replay(mw, mb):
        """The two passes on the oracles from the dense weights (mw, mb) -> (record table, mean loss per pass, rows deleted
        by the end-of-pass shrink)."""
        mw, mb = [w.copy() for w in mw], [b_.copy() for b_ in mb]
        st = [[np.zeros_like(w), np.zeros_like(w)] for w in mw], [[np.zeros_like(b), np.zeros_like(b)] for b in mb]
        rec = np.zeros((N, L.row_stride), np.float32)
        data = open(os.path.join(GOLDEN, "slot_dnn_demo_4.txt"), "rb").read().split(b"\n")
        lines = [ln for ln in data if ln.strip()]
        step, want_loss, want_deleted = 0, [], []
        for epoch in range(2):
            losses = []
            for b0 in range(0, len(lines) - 1, 2):
                chunk = b"\n".join(lines[b0:b0 + 2]) + b"\n"
                values, lod, base, n = reader.parse_feasign_slots(chunk, 2, SLOTS, 0)
                lv, llod, _, _ = reader.parse_feasign_slots(chunk, 1, 1, 0)
                label = lv[llod[0, :-1]].reshape(n, 1).clamp(0, 1).numpy()
                values, lod, base = values.numpy(), lod.numpy(), base.numpy()
                # what a pull shows (PullSparse + Select): the stored W of every key, zeros for keys that do not exist
                Wv = np.stack([ps_ref.pull_value(rec, lay, r, acc, D) for r in range(N)])
                Wv[0] = 0
                o = M.loss_and_grads(values, lod, base, label, Wv, mw, mb, 0, 1, N)
                losses.append(float(o["loss"]))
                U = len(o["uniq"])
                dshow, dclick = np.zeros(U), np.zeros(U)
                pos = {int(r): i for i, r in enumerate(o["uniq"])}
                for k in np.nonzero(values != 0)[0]:
                    dshow[pos[int(o["rows"][k])]] += 1
                    dclick[pos[int(o["rows"][k])]] += int(label[o["seg"][k] // SLOTS, 0])
                ps_ref.push_rows(rec, lay, o["uniq"], o["merged"][:, 0], o["merged"][:, 1:], dshow, dclick,
                                 dict(acc, grad_scale=float(n)))      # the pushed gradient is that of the SUMMED loss
                step += 1
                for i in range(len(mw)):
                    R.adam_update(mw[i], st[0][i][0], st[0][i][1], o["dws"][i], step, lr=1e-3)
                    R.adam_update(mb[i], st[1][i][0], st[1][i][1], o["dbs"][i], step, lr=1e-3)
            want_loss.append(float(np.mean(losses)))
            want_deleted.append(ps_ref.shrink_rows(rec, lay, acc, 0.98, 0.15))

        return rec, want_loss, want_deleted

    rec, want_loss, want_deleted = replay(mw, mb)
    # the same replay from dense weights moved by ONE ulp: the sensitivity of this two-pass trajectory to fp32 round-off
    # (the dense Adam turns the sign noise of ~eps-sized gradients into lr-sized steps) — the measured floor of the bars below
    rec_ulp, loss_ulp, _ = replay([np.nextafter(w, np.float32(np.inf)) for w in mw],
                                  [np.nextafter(b_, np.float32(np.inf)) for b_ in mb])

    # losses: 1e-5, or 4 x the measured sensitivity floor (no looser bar for the second pass)
    for i in range(2):
        lf = abs(loss_ulp[i] - want_loss[i])
        assert abs(res["loss"][i] - want_loss[i]) <= max(loss_rtol * abs(want_loss[i]), 4.0 * lf), \
            (i, res["loss"][i], want_loss[i], lf)
    assert res["deleted"] == want_deleted and want_deleted[0] > 0            # the shrink really deletes rows
    got = net.rec.cpu().numpy()
    so = L.stat_off
    assert np.array_equal(got[:, so + 4], rec[:, so + 4]), "feature states after two passes"
    np.testing.assert_allclose(got[:, so:so + 2], rec[:, so:so + 2], rtol=1e-6, atol=0)
    # weights: 1e-5 of the tensor's scale, or 4 x the measured floor of the oracle against itself (one-ulp replay above)
    wscale = float(np.abs(rec[:, :D]).max())
    wfloor = float(np.abs(rec_ulp[:, :D] - rec[:, :D]).max())
    werr = np.abs(got[:, :D] - rec[:, :D])
    assert float(werr.max()) <= max(1e-5 * wscale, 4.0 * wfloor), (float(werr.max()), wscale, wfloor)
    states = rec[:, so + 4]
    assert (states == 1).any() and (states == 2).any()
    # ---- pass checkpoint: born rows only, round trip
    z = np.load(os.path.join(cfg["runner.model_save_path"], "1", "rec_gpubox.npz"))
    assert 0 < len(z["rows"]) < N // 2
    before = net.rec.clone()
    net.rec.zero_()
    main.load_pass(os.path.join(cfg["runner.model_save_path"], "1"))
    # the checkpoint of epoch 1 is written after that epoch's end_pass: identical table
    assert torch.equal(before, net.rec)


def test_gpubox_pass_loop_cpu_backend(tmp_path):
    import cpu_kernels
    _run(tmp_path, "cpu", cpu_kernels)


@pytest.mark.gpu
def test_gpubox_pass_loop_gpu(tmp_path, engine_lib):
    _run(tmp_path, "cuda", None)


REF_CFG = "/root/reference/models/rank/slot_dnn/config_online.yaml"


@pytest.mark.skipif(not os.path.exists(REF_CFG), reason="reference tree not mounted (only in the build container)")
def test_reference_online_yaml_and_data_run_unchanged(tmp_path):
    """The reference's OWN slot_dnn/config_online.yaml (300 slots, D 9, accessor block :57-89) and its data directory
    (demo_10) through the pass loop — only the batch size (its 32 exceeds the 10 demo lines), the pass count and the
    output directory are overridden."""
    import cpu_kernels
    from paddlerec_amd import gpubox
    cfg = gpubox.load_config(REF_CFG, ["runner.train_batch_size=5", "runner.epochs=2",
                                       "runner.model_save_path=" + str(tmp_path / "out")])
    acc = cfg["table_parameters.embedding.accessor"]
    assert acc["embedx_threshold"] == 10 and acc["ctr_accessor_param"]["show_click_decay_rate"] == 0.98
    main = gpubox.Main(cfg, "cpu", cpu_kernels)
    res = main.run_worker()
    assert len(res["loss"]) == 2 and all(np.isfinite(x) for x in res["loss"]) and all(0 <= a <= 1 for a in res["auc"])
    assert main.net.slot_num == 300 and main.net.emb_dim == 9 and main.net.table.accessor.embedx_threshold == 10
    assert os.path.exists(tmp_path / "out" / "1" / "rec_gpubox.npz")


def test_in_memory_reader_cuts_the_batches_a_per_batch_parse_gives(tmp_path):
    """InMemoryReader.load_into_memory parses each file once (whole, mmap'ed) and cuts batches by index arithmetic:
    every batch equals what rec_parse_feasign_slots gives for exactly those lines parsed on their own — across file
    boundaries, with blank lines, an empty file, and batch sizes that do / do not divide the line count (drop_last)."""
    import random
    from paddlerec_amd import gpubox, reader as rd
    lines = [ln for ln in open(os.path.join(GOLDEN, "slot_dnn_demo_4.txt"), "rb").read().split(b"\n") if ln.strip()]
    random.seed(1)
    pool = []
    for k in range(11):                     # 11 lines with 50..420 feasigns each, labels alternating
        feat = lines[k % 4].split(b" ")[1:]
        random.shuffle(feat)
        feat = sorted(feat[: 50 + 37 * k], key=lambda t: int(t.split(b":")[1]))
        pool.append(b" ".join([b"%d:1" % (k % 2)] + feat))
    files = []
    for i, (a, z) in enumerate(((0, 3), (3, 4), (4, 9), (9, 11))):
        p = tmp_path / ("part-%d" % i)
        p.write_bytes(b"\n".join(pool[a:z]) + (b"\n \n\n" if i % 2 else b"\n"))      # blank lines after odd files
        files.append(str(p))
    (tmp_path / "empty").write_bytes(b"")
    files.insert(2, str(tmp_path / "empty"))
    assert rd.blank_lines(b"a:1\n\n \t\nb:2\n").tolist() == [1, 2]
    for B in (1, 2, 3, 4, 5, 11, 12):
        r = gpubox.InMemoryReader(files, B, 300)
        assert r.load_into_memory() == 11 // B
        for i, (values, lod, base, label) in enumerate(r.batches):
            data = b"\n".join(pool[i * B:(i + 1) * B]) + b"\n"
            v2, l2, b2, n = rd.parse_feasign_slots(data, 2, 300, 0, 0)
            lv, llod, _, _ = rd.parse_feasign_slots(data, 1, 1, 0, 0)
            assert torch.equal(values, v2) and torch.equal(lod, l2) and torch.equal(base, b2), (B, i)
            assert torch.equal(label, lv[llod[0, :-1]].reshape(n, 1).clamp_(0, 1)), (B, i)
