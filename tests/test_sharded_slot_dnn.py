"""BenchmarkDNNLayer (slot_dnn, the gpubox model: tools/run_gpubox.sh:24, slot_dnn/net.py:55-85) on the ROW-SHARDED PS
accessor table (paddlerec_amd/sharded_slot_dnn.py): 2 gloo ranks == ONE unsharded oracle run on the concatenated
batches of the reference's own multi-value lines — per-value owner routing, pooled sums on the requester, pushed
show / click / summed-loss gradients merged on the owner."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO
from oracle import deepfm_ref as R
from oracle import ps_ref
from oracle import slot_dnn_ref as M

sys.path.insert(0, os.path.join(REPO, "tests"))


def _run_world(world, mode, tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
    worker = os.path.join(REPO, "tests", "_sharded_slot_worker.py")
    env = dict(os.environ, OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), port, mode, str(tmp_path)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, outs[r][-3000:])
    return [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]


def _expected(world):
    from _sharded_slot_worker import CFG as c, global_batches, parse
    N, D, S = c["N"], c["D"], c["S"]
    acc = dict(c["accessor"])
    lay = dict(embed_off=0, embedx_off=1, embedx_dim=D - 1, stat_off=D)
    rec = np.zeros((N, 16), np.float32)
    rng = np.random.default_rng(5)
    sizes = [D * S] + list(c["layers"]) + [1]
    mw, mb = [], []
    for i in range(len(sizes) - 1):
        mw.append((rng.standard_normal((sizes[i], sizes[i + 1])) * 0.05).astype(np.float32))
        mb.append((rng.standard_normal((sizes[i + 1],)) * 0.05).astype(np.float32))
    st = [[np.zeros_like(w), np.zeros_like(w)] for w in mw], [[np.zeros_like(b), np.zeros_like(b)] for b in mb]
    losses, preds, rec0 = [], [], None
    for step, lines in enumerate(global_batches(world)):
        values, lod, base, label = (a.numpy() for a in parse(lines, S))
        # what a pull shows (PullSparse + Select): the stored W of every key, zeros for keys that do not exist
        Wv = np.stack([ps_ref.pull_value(rec, lay, r, acc, D) for r in range(N)])
        Wv[0] = 0
        o = M.loss_and_grads(values, lod, base, label, Wv, mw, mb, 0, 1, N)
        losses.append(float(o["loss"]))
        preds.append(o["pred"])
        U = len(o["uniq"])
        dshow, dclick = np.zeros(U), np.zeros(U)
        pos = {int(r): i for i, r in enumerate(o["uniq"])}
        for k in np.nonzero(values != 0)[0]:
            dshow[pos[int(o["rows"][k])]] += 1
            dclick[pos[int(o["rows"][k])]] += int(label[o["seg"][k] // S, 0])
        ps_ref.push_rows(rec, lay, o["uniq"], o["merged"][:, 0], o["merged"][:, 1:], dshow, dclick,
                         dict(acc, grad_scale=float(len(label))))
        if step == 0:
            rec0 = rec.copy()
        for i in range(len(mw)):
            R.adam_update(mw[i], st[0][i][0], st[0][i][1], o["dws"][i], step + 1, lr=c["lr"])
            R.adam_update(mb[i], st[1][i][0], st[1][i][1], o["dbs"][i], step + 1, lr=c["lr"])
    return c, rec, losses, preds, mw, rec0, st


def _check(world, ranks):
    c, rec, losses, preds, mw, rec0, st = _expected(world)
    D, B = c["D"], c["B"]
    so = D
    for r, out in enumerate(ranks):
        assert int(out["status"][0]) == 0
        assert out["trace"].tolist() == ranks[0]["trace"].tolist() and len(out["trace"]) >= 7 * c["steps"]
        # The dense learning rate of this test is 1e-6: Adam moves an entry whose gradient is ~eps-sized by lr * sign(noise),
        # and with a real lr that sign noise of the fp32 summation order feeds back into every later step's embedding
        # gradients (tests/test_gpubox.py) — here the multi-step object under test is the SHARDED TABLE, and the dense
        # path is checked through its all-reduced Adam moments instead.
        for s_, want in enumerate(losses):
            np.testing.assert_allclose(out["loss%d" % s_][0], want, rtol=2e-5)
            np.testing.assert_allclose(out["pred%d" % s_], preds[s_][r * B:(r + 1) * B], rtol=2e-5, atol=1e-6)
        from helpers import assert_close_scaled
        assert_close_scaled(out["m_w0"], st[0][0][0])
        assert_close_scaled(out["v_w0"], st[0][0][1])
        mine = rec[r::world]                                    # owner(row) = row % world, local = row // world
        got = out["rec"][: mine.shape[0]]
        assert np.array_equal(got[:, so:so + 2], mine[:, so:so + 2]), "show / click counters of rank %d" % r
        assert np.array_equal(got[:, so + 4], mine[:, so + 4]), "feature states of rank %d" % r
        assert np.array_equal(got[:, so + 6], mine[:, so + 6]), "unseen_days of rank %d" % r
        np.testing.assert_allclose(got[:, so + 5], mine[:, so + 5], rtol=1e-6, err_msg="delta_score")
        # the table after the FIRST step at the stated bar (1e-5 of the weight scale) ...
        mine0, got0 = rec0[r::world], out["rec_step0"][: mine.shape[0]]
        w0 = float(np.abs(rec0[:, :D]).max())
        np.testing.assert_allclose(got0[:, :D], mine0[:, :D], rtol=1e-5, atol=1e-5 * w0)
        np.testing.assert_allclose(got0[:, so + 2:so + 4], mine0[:, so + 2:so + 4], rtol=2e-5, atol=1e-12)
        # ... and after three
        wscale = float(np.abs(rec[:, :D]).max())
        g2 = float(np.abs(rec[:, so + 2:so + 4]).max())
        np.testing.assert_allclose(got[:, :D], mine[:, :D], rtol=2e-5, atol=2e-5 * wscale)
        np.testing.assert_allclose(got[:, so + 2:so + 4], mine[:, so + 2:so + 4], rtol=2e-5, atol=2e-5 * g2)
        assert np.array_equal(out["mlp_w0"], ranks[0]["mlp_w0"])          # replicas stay identical
    st = rec[:, so + 4]
    assert (st == 0).any() and (st == 1).any() and (st == 2).any()
    assert any(rec[r::world][:, so + 4].any() for r in range(world))


def test_sharded_slot_dnn_world2_cpu(tmp_path):
    _check(2, _run_world(2, "cpu", tmp_path))


@pytest.mark.gpu
def test_sharded_slot_dnn_two_ranks_one_gpu(tmp_path, engine_lib):
    _check(2, _run_world(2, "gpu", tmp_path))
