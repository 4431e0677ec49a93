"""Row-sharded DeepFM (paddlerec_amd/sharded.py): a G-rank sharded run must equal ONE unsharded
oracle run on the concatenated batch.

  * test_shard_route_oracle_*      : the routing contract itself (CPU, NumPy oracle).
  * test_sharded_orchestration_cpu : world_size-2 gloo on CPU, operator backend = oracle stand-in —
                                     checks split bookkeeping / exchange order / unscrambling.
  * test_shard_route_gpu, test_sharded_two_ranks_one_gpu (gpu): the HIP partition kernel bit-exact
    against the oracle, and 2 processes sharing cuda:0 running the real kernels end to end.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO
from helpers import OracleTrainer, make_deepfm_problem
from oracle import shard_ref

WORKER = os.path.join(REPO, "tests", "_sharded_worker.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return str(p)


def _run_world(world, mode, outdir, tables=False, worker=None, extra_env=None):
    port = _free_port()
    env = dict(os.environ, OMP_NUM_THREADS="2", **(extra_env or {}))
    procs = [subprocess.Popen([sys.executable, worker or WORKER, str(r), str(world), port, mode, str(outdir)]
                              + (["tables"] if tables else []), env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode("utf-8", "replace"))
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, outs[r][-3000:])
    return [dict(np.load(os.path.join(outdir, "rank%d.npz" % r))) for r in range(world)]


def _expected(world, tables, zipf=False):
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from _sharded_worker import CFG as c, later_ids
    pr = make_deepfm_problem(B=c["B"] * world, N=c["N"], D=c["D"], fc=c["fc"], seed=c["seed"],
                             pad_frac=c["pad_frac"], tables=tables, zipf=zipf)
    tr = OracleTrainer(pr["params"], pr["slot_offsets"], lr=c["lr"])
    rng = np.random.default_rng(c["seed"] + 1)
    res = []
    for step in range(c["steps"]):
        if step == 0:
            ids, dense, label = pr["ids"], pr["dense"], pr["label"]
        else:
            ids = later_ids(rng, c, world, zipf)
            dense = rng.random((c["B"] * world, 13), dtype=np.float32)
            label = (rng.random((c["B"] * world, 1)) < 0.3).astype(np.int64)
        res.append(tr.train_step(ids, dense, label))
    from oracle import deepfm_ref as R
    pred_eval, _, _ = R.deepfm_forward(pr["ids"], pr["dense"], tr.p, slot_offsets=pr["slot_offsets"])
    return c, tr, res, pred_eval


def _check(world, ranks, tables, zipf=False):
    c, tr, res, pred_eval = _expected(world, tables, zipf)
    B = c["B"]
    for r, out in enumerate(ranks):
        assert int(out["status"][0]) == 0
        # one communicator: every rank must issue the same sequence of collectives (a divergence hangs on real GPUs)
        assert out["trace"].tolist() == ranks[0]["trace"].tolist() and len(out["trace"]) > 10
        for step, (loss, pred) in enumerate(res):
            np.testing.assert_allclose(out["loss%d" % step][0], loss, rtol=2e-5)
            np.testing.assert_allclose(out["pred%d" % step], pred[r * B:(r + 1) * B], rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(out["pred_eval"], pred_eval[r * B:(r + 1) * B], rtol=1e-4, atol=1e-5)
        from helpers import assert_adam_weights_close
        for k, w in (("W", tr.p["W"]), ("W1", tr.p["W1"]), ("mlp_w0", tr.p["mlp_w"][0]), ("dense_w", tr.p["dense_w"])):
            assert_adam_weights_close(out[k], w, lr=c["lr"], steps=c["steps"], err_msg="rank %d %s" % (r, k))
        # the optimizer state at the stated bar: Adam's moments within 1e-5 of their scale (the weights above carry
        # lr-sized differences where a gradient is ~eps-sized, helpers.assert_moments_close)
        from helpers import assert_close_scaled
        assert_close_scaled(out["m_mlp_w0"], tr.dstate[("mlp_w", 0)][0])
        assert_close_scaled(out["v_mlp_w0"], tr.dstate[("mlp_w", 0)][1])
        assert_close_scaled(out["m_dense_w"], tr.dstate["dense_w"][0])
        assert_close_scaled(out["v_dense_w"], tr.dstate["dense_w"][1])
        mine = tr.st["mW"][r::world]
        assert_close_scaled(out["mW_local"][: mine.shape[0]], mine)
        assert_close_scaled(out["vW_local"][: mine.shape[0]], tr.st["vW"][r::world])
    # every rank holds the same replicated dense parameters and sees the same global tables
    for out in ranks[1:]:
        assert np.array_equal(out["mlp_w0"], ranks[0]["mlp_w0"])
        assert np.array_equal(out["W"], ranks[0]["W"])


# ------------------------------------------------------------------------------ routing contract (CPU)
@pytest.mark.parametrize("G", [1, 2, 3, 8])
def test_shard_route_oracle_properties(G):
    pr = make_deepfm_problem(B=37, N=101, seed=G, pad_frac=0.1, tables=(G == 3))
    r = shard_ref.shard_route(pr["ids"], G, 0, pr["slot_offsets"])
    ids = pr["ids"].reshape(-1)
    n = ids.size
    nv = int((ids != 0).sum())
    assert r["send_counts"][:G].sum() == nv and r["send_counts"][G] == n - nv
    rows = (pr["ids"] + (pr["slot_offsets"][None] if pr["slot_offsets"] is not None else 0)).reshape(-1)
    # send order: grouped by owner, ascending position inside a group
    owner = rows[r["send_pos"]] % G
    assert np.all(np.diff(owner) >= 0)
    for d in range(G):
        assert np.all(np.diff(r["send_pos"][owner == d]) > 0)
    assert np.array_equal(r["send_local_row"] * G + owner, rows[r["send_pos"]])
    assert np.array_equal(r["send_sample"], r["send_pos"] // 26)
    # slot_of_pos inverts send order and is 0 exactly on padding
    assert np.array_equal(r["slot_of_pos"][r["send_pos"]], np.arange(1, nv + 1))
    assert np.array_equal(r["slot_of_pos"] == 0, ids == 0)


@pytest.mark.parametrize("tables", [False, True])
def test_sharded_orchestration_cpu(tmp_path, tables):
    _check(2, _run_world(2, "cpu", tmp_path, tables), tables)


def test_sharded_orchestration_cpu_world3(tmp_path):
    _check(3, _run_world(3, "cpu", tmp_path), False)


# ---- the deduplicated exchange (REC_SHARD_DEDUP=1): distinct rows per owner in fixed-capacity slots, replies expanded and
# gradients merged locally, no host read on the step path — the same checks as above must hold with the switch on
DEDUP = {"REC_SHARD_DEDUP": "1"}


@pytest.mark.parametrize("world,tables", [(2, False), (2, True), (3, False)])
def test_sharded_dedup_exchange_cpu(tmp_path, world, tables):
    _check(world, _run_world(world, "cpu", tmp_path, tables, extra_env=DEDUP), tables)


def test_sharded_dedup_exchange_zipf_ids_cpu(tmp_path):
    """Zipf ids (one row owning a large share of a batch's lookups): the case fixed-capacity buckets could not take without
    deduplication — a hot row is ONE slot here, whatever its share."""
    env = dict(DEDUP, REC_TEST_ZIPF="1")
    _check(2, _run_world(2, "cpu", tmp_path, extra_env=env), False, zipf=True)


@pytest.mark.gpu
@pytest.mark.parametrize("world,tables", [(2, False), (2, True), (1, False)])
def test_sharded_dedup_exchange_ranks_share_one_gpu(engine_lib, tmp_path, world, tables):
    """The HIP kernels + the device-side plan (ops.dedup_plan / dedup_merge), and no .item() / .tolist() on the step path."""
    env = dict(DEDUP, REC_TEST_NO_HOST_READS="1")
    _check(world, _run_world(world, "gpu", tmp_path, tables, extra_env=env), tables)


@pytest.mark.gpu
def test_dedup_plan_device_equals_the_oracle_statement(engine_lib):
    """ops.dedup_plan (rec_ids_group + device index arithmetic) == tests/cpu_kernels.dedup_plan on every integer it emits;
    ops.dedup_merge == the position-ordered merge, bit for bit."""
    import torch
    import cpu_kernels as K
    from paddlerec_amd import ops
    rng = np.random.default_rng(8)
    for B, S, N, G, cap, so in ((300, 26, 5003, 8, 2000, False), (64, 26, 100, 3, 30, False), (128, 4, 997, 2, 512, True)):
        local = ((N * S if so else N) + G - 1) // G
        ids = np.minimum(rng.zipf(1.2, size=(B, S)), N - 1).astype(np.int64)
        ids[rng.random((B, S)) < 0.05] = 0
        slot_off = (np.arange(S, dtype=np.int64) * N) if so else None
        rows_total = N * S if so else N
        st_d, st_h = ops.new_status("cuda"), K.new_status("cpu")
        ws = ops.Workspace("cuda")
        pd, _ = ops.dedup_plan(torch.from_numpy(ids).cuda(), rows_total, 0, G, local, cap, ws,
                               None if slot_off is None else torch.from_numpy(slot_off).cuda(), st_d)
        ph, _ = K.dedup_plan(torch.from_numpy(ids), rows_total, 0, G, local, cap, None,
                             None if slot_off is None else torch.from_numpy(slot_off), st_h)
        assert np.array_equal(pd.send_rows.cpu().numpy(), ph.send_rows.numpy())
        assert np.array_equal(pd.slot_of_pos[: B * S].cpu().numpy(), ph.slot_of_pos.numpy()[: B * S])
        assert np.array_equal(pd.counts.cpu().numpy(), ph.counts.numpy())
        assert int(st_d.item()) == int(st_h.item())
        grad = rng.standard_normal((B * S, 16)).astype(np.float32)
        md = ops.dedup_merge(pd, torch.from_numpy(grad).cuda(), 16)[: G * cap].cpu().numpy()
        mh = K.dedup_merge(ph, torch.from_numpy(grad), 16)[: G * cap].numpy()
        assert np.array_equal(md, mh)
        dz = rng.standard_normal((B, 1)).astype(np.float32)
        m1d = ops.dedup_merge(pd, torch.from_numpy(dz).cuda(), 1, grad_div=S)[: G * cap].cpu().numpy()
        m1h = K.dedup_merge(ph, torch.from_numpy(dz), 1, grad_div=S)[: G * cap].numpy()
        assert np.array_equal(m1d, m1h)


def test_dedup_plan_properties():
    """ops.dedup_plan's contract on the oracle-side statement (tests/cpu_kernels.py; the GPU test compares the device
    implementation with it): owner-major ascending distinct rows, slots inside the capacity, padding -> slot 0, every
    position's slot holds its row, overflow flagged."""
    import torch
    import cpu_kernels as K
    rng = np.random.default_rng(4)
    B, S, N, G = 40, 26, 997, 3
    local = (N + G - 1) // G
    ids = rng.integers(0, N, (B, S), dtype=np.int64)
    ids[rng.random((B, S)) < 0.1] = 0
    ids[:, 3] = 17                                             # a hot row
    st = K.new_status("cpu")
    plan, _ = K.dedup_plan(torch.from_numpy(ids), N, 0, G, local, B * S, None, None, st)
    send = plan.send_rows.numpy().reshape(G, -1)
    sop = plan.slot_of_pos.numpy()
    assert int(st.item()) == 0
    for o in range(G):
        live = send[o][send[o] != local]
        want = np.unique(ids[(ids != 0) & (ids % G == o)] // G)
        assert np.array_equal(live, want)                      # distinct, ascending, nothing else
        assert np.array_equal(send[o][: len(live)], live) and int(plan.counts[o]) == len(want)
    flat = ids.reshape(-1)
    assert np.array_equal(sop == 0, flat == 0)
    slot = sop[flat != 0] - 1
    assert np.array_equal(send.reshape(-1)[slot], flat[flat != 0] // G) and np.array_equal(slot // send.shape[1], flat[flat != 0] % G)
    # a capacity below the need: flagged, the rows behind it read slot 0
    st2 = K.new_status("cpu")
    plan2, _ = K.dedup_plan(torch.from_numpy(ids), N, 0, G, local, 8, None, None, st2)
    assert int(st2.item()) & 2 and (plan2.slot_of_pos.numpy()[flat != 0] == 0).any()


# ------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("B,N,G,pad,tables", [(1, 50, 2, 0.0, False), (300, 1000, 8, 0.05, False),
                                               (1000, 5000, 3, 0.03, True), (64, 10, 2, 1.0, False),
                                               (4096, 100000, 8, 0.03, True), (100, 7, 64, 0.2, False)])
def test_shard_route_gpu(engine_lib, B, N, G, pad, tables):
    import torch
    from paddlerec_amd import ops
    pr = make_deepfm_problem(B=B, N=N, seed=B + G, pad_frac=pad, tables=tables)
    want = shard_ref.shard_route(pr["ids"], G, 0, pr["slot_offsets"])
    dev = "cuda"
    so = None if pr["slot_offsets"] is None else torch.as_tensor(pr["slot_offsets"]).to(dev)
    route, status = ops.shard_route(torch.as_tensor(pr["ids"]).to(dev), pr["N"], 0, G, ops.Workspace(dev), so)
    assert int(status.item()) == 0
    k = len(want["send_pos"])
    assert np.array_equal(route.send_counts.cpu().numpy(), want["send_counts"])
    assert np.array_equal(route.send_local_row[:k].cpu().numpy(), want["send_local_row"])
    assert np.array_equal(route.send_pos[:k].cpu().numpy(), want["send_pos"])
    assert np.array_equal(route.send_sample[:k].cpu().numpy(), want["send_sample"])
    assert np.array_equal(route.slot_of_pos[:B * 26].cpu().numpy(), want["slot_of_pos"])


@pytest.mark.gpu
def test_shard_route_flags_out_of_range(engine_lib):
    import torch
    from paddlerec_amd import ops
    ids = torch.tensor([[1, 2, 99, -3]], dtype=torch.int64, device="cuda")
    route, status = ops.shard_route(ids, 10, 0, 2, ops.Workspace("cuda"))
    assert int(status.item()) & 1
    assert route.send_counts.tolist() == [1, 1, 2]


@pytest.mark.gpu
@pytest.mark.parametrize("world,tables", [(2, False), (2, True), (1, False)])
def test_sharded_ranks_share_one_gpu(engine_lib, tmp_path, world, tables):
    """2 processes on cuda:0 (gloo transport, host-staged) running the real HIP kernels."""
    _check(world, _run_world(world, "gpu", tmp_path, tables), tables)


# ------------------------------------------------------------------------------- PS / gpubox table (configs[4])
def _expected_ps(world):
    """ONE unsharded run on the concatenated batches: hashed rows, lazily born accessor table, AdaGrad push."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from _sharded_ps_worker import CFG as c, make_batches
    from oracle import deepfm_ref as R
    from oracle import ps_ref
    from oracle import slot_dnn_ref
    D, N = c["D"], c["N"]
    pr = make_deepfm_problem(B=4, N=N, D=D, fc=c["fc"], seed=c["seed"])
    p = {k: (v.copy() if not isinstance(v, list) else [x.copy() for x in v]) for k, v in pr["params"].items()}
    acc = dict(c["accessor"])
    lay = dict(embed_off=D, embedx_off=0, embedx_dim=D, stat_off=D + 1)
    rec = np.zeros((N, 32), np.float32)
    dstate, losses = {}, []
    for step, (keys, dense, label) in enumerate(make_batches(world)):
        rows = slot_dnn_ref.feasign_rows(keys.astype(np.uint64), N)
        p["W"], w1 = ps_ref.pull_deepfm(rec, lay, np.arange(N), acc)
        p["W1"] = w1.reshape(-1, 1)
        o = R.deepfm_loss_and_grads(rows, dense, label, p)
        losses.append(float(o["loss"]))
        uniq, merged, counts = R.merge_rows(o["rows"], o["row_valid"], o["row_grad"])
        _, merged1, _ = R.merge_rows(o["rows"], o["row_valid"], o["row_grad1"])
        lab = np.repeat(label.reshape(-1), rows.shape[1])
        flat = o["rows"].reshape(-1)
        clicks = np.array([lab[(flat == u) & o["row_valid"].reshape(-1)].sum() for u in uniq], np.float64)
        # the layer pushes the gradient of the SUMMED loss: grad_scale = the global batch (scale_sparse_grad)
        ps_ref.push_rows(rec, lay, uniq, merged1[:, 0], merged, counts, clicks,
                         dict(acc, grad_scale=float(label.shape[0])))
        pairs = [("dense_w", o["d_dense_w"]), ("dense_w_one", o["d_dense_w_one"])]
        for i in range(len(p["mlp_w"])):
            pairs += [(("mlp_w", i), o["mlp_dw"][i]), (("mlp_b", i), o["mlp_db"][i])]
        for key, gr in pairs:
            arr = p[key] if not isinstance(key, tuple) else p[key[0]][key[1]]
            mm, vv = dstate.setdefault(key, (np.zeros_like(arr), np.zeros_like(arr)))
            R.adam_update(arr, mm, vv, gr.reshape(arr.shape).astype(arr.dtype), step + 1, lr=c["lr"])
    return c, rec, losses, p


def _check_ps(world, ranks):
    c, rec, losses, p = _expected_ps(world)
    D = c["D"]
    for r, out in enumerate(ranks):
        assert int(out["status"][0]) == 0
        assert out["trace"].tolist() == ranks[0]["trace"].tolist()
        for s, want in enumerate(losses):
            np.testing.assert_allclose(out["loss%d" % s][0], want, rtol=2e-5)
        mine = rec[r::world]                                    # owner(row) = row % world, local = row // world
        got = out["rec"][: mine.shape[0]]
        so = D + 1
        assert np.array_equal(got[:, so:so + 2], mine[:, so:so + 2]), "show / click counters of rank %d" % r
        assert np.array_equal(got[:, so + 4], mine[:, so + 4]), "feature states of rank %d" % r
        assert np.array_equal(got[:, so + 6], mine[:, so + 6]), "unseen_days of rank %d" % r
        np.testing.assert_allclose(got[:, so + 5], mine[:, so + 5], rtol=1e-6, err_msg="delta_score")
        wscale = float(np.abs(rec[:, :D + 1]).max())
        np.testing.assert_allclose(got[:, :D + 1], mine[:, :D + 1], rtol=1e-4, atol=1e-5 * wscale)
        np.testing.assert_allclose(got[:, so + 2:so + 4], mine[:, so + 2:so + 4], rtol=1e-4, atol=1e-12)
        from helpers import assert_adam_weights_close
        assert_adam_weights_close(out["mlp_w0"], p["mlp_w"][0], lr=c["lr"], steps=c["steps"], err_msg="rank %d mlp_w0" % r)
    st = rec[:, D + 5]
    assert (st == 0).any() and (st == 1).any() and (st == 2).any()       # unborn, embed-only and full features occur


def test_sharded_ps_table_world2_cpu(tmp_path):
    """configs[4] in miniature: uint64 feasigns hashed on the device, the AdaGrad accessor table row-sharded over 2
    gloo ranks == one unsharded oracle run on the concatenated batch (rows born with the same values on either)."""
    worker = os.path.join(REPO, "tests", "_sharded_ps_worker.py")
    _check_ps(2, _run_world(2, "cpu", tmp_path, worker=worker))


@pytest.mark.gpu
def test_sharded_ps_table_two_ranks_one_gpu(tmp_path):
    worker = os.path.join(REPO, "tests", "_sharded_ps_worker.py")
    _check_ps(2, _run_world(2, "gpu", tmp_path, worker=worker))


def test_sharded_ps_table_dedup_world2_cpu(tmp_path):
    """The accessor table behind the deduplicated exchange: a distinct row's occurrence and click counts travel with its
    merged gradient; counters stay exact."""
    worker = os.path.join(REPO, "tests", "_sharded_ps_worker.py")
    _check_ps(2, _run_world(2, "cpu", tmp_path, worker=worker, extra_env=DEDUP))


@pytest.mark.gpu
def test_sharded_ps_table_dedup_two_ranks_one_gpu(tmp_path):
    worker = os.path.join(REPO, "tests", "_sharded_ps_worker.py")
    _check_ps(2, _run_world(2, "gpu", tmp_path, worker=worker, extra_env=DEDUP))
