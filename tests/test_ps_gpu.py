"""PS / gpubox row: CVM lookup (skip the show/click columns) and the AdaGrad accessor rule on a record table."""
import numpy as np
import pytest
import torch

from oracle import deepfm_ref as R
from oracle import ps_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize("D,B,N", [(9, 200, 50), (16, 1000, 5000), (1, 64, 10)])
def test_cvm_lookup_and_adagrad_rule(engine_lib, D, B, N):
    from paddlerec_amd import ops
    rng = np.random.default_rng(D + B)
    S = 26
    rec = np.zeros((N, 32), np.float32)
    rec[:, 4:4 + D] = rng.uniform(-1e-4, 1e-4, (N, D))                 # initial_range 1e-4
    rec[:, 2:4] = rng.uniform(0, 2, (N, 2))                            # some accumulated g2sum
    ids = rng.integers(0, N, (B, S), dtype=np.int64)
    label = (rng.random(B) < 0.3).astype(np.int64)
    trec = T(rec)
    # lookup: table view = rec[:, 4:4+D] (continuous_value_model(use_cvm=False): no CVM columns)
    got, status = ops.emb_gather(T(ids), trec[:, 4:4 + D], None)
    assert np.array_equal(got.cpu().numpy(), ps_ref.cvm_lookup(rec, ids, D))
    # push: per-position gradients, merged per row, AdaGrad rule + show/click
    grad = (rng.standard_normal((B * S, D)) * 0.1).astype(np.float32)
    ws = ops.Workspace(DEV)
    groups, _ = ops.ids_group(T(ids), N, None, ws)
    ops.sparse_adagrad_rows(groups, T(grad), trec, D, S, label=T(label))
    rows = ids.reshape(-1)
    uniq, merged, counts = R.merge_rows(rows, np.ones_like(rows, bool), grad)
    clicks = np.zeros(len(uniq))
    lab_pos = np.repeat(label, S)
    for u, row in enumerate(uniq):
        clicks[u] = lab_pos[rows == row].sum()
    want = rec.copy()
    ps_ref.adagrad_rows(want, D, uniq, merged, counts, clicks)
    got = trec.cpu().numpy()
    assert np.array_equal(got[:, :2], want[:, :2])                      # show / click counters: exact
    np.testing.assert_allclose(got[:, 2:4 + D], want[:, 2:4 + D], rtol=1e-5, atol=1e-9)
    untouched = np.setdiff1d(np.arange(N), uniq)
    assert np.array_equal(got[untouched], rec[untouched])
    assert np.all(np.abs(got[:, 4:4 + D]) <= 10.0)


def test_multi_value_slots_file_to_sum_pool(engine_lib):
    """Row P end to end on the reference's own multi-value fixture (first lines of slot_dnn/data/demo_10):
    host parser (queuedataset_reader.py:56-82) -> hashed rows -> one rec_emb_gather_sumpool launch per slot
    == sparse_embedding(padding_idx=0) + sequence_pool('sum') of slot_dnn/net.py:63-75 (oracle), counts bit-exact."""
    import os
    from conftest import GOLDEN
    from paddlerec_amd import ops, reader
    data = open(os.path.join(GOLDEN, "slot_dnn_demo_4.txt"), "rb").read()
    N, D = 100003, 9
    values, lod, base, n = reader.parse_feasign_slots(data, 2, 300, N)            # slots "2".."301" (slot "1" = click)
    rng = np.random.default_rng(1)
    W = rng.standard_normal((N, D)).astype(np.float32)
    W[0] = 0
    tW, tv, tl = T(W), T(values.numpy()), T(lod.numpy())
    nnz_per_slot = (base[1:] - base[:-1]).numpy()
    picked = list(np.argsort(-nnz_per_slot)[:6]) + [0, 5, 299]                     # the fattest slots + some thin ones
    for s in picked:
        ids = tv[int(base[s]): int(base[s + 1])]
        out, counts, status = ops.emb_gather_sumpool(ids.contiguous(), tl[s].contiguous(), tW, 0)
        want, wcnt = R.sequence_pool_sum(W, values.numpy()[int(base[s]): int(base[s + 1])], lod.numpy()[s], 0)
        assert np.array_equal(counts.cpu().numpy(), wcnt.astype(np.int32))
        np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-5, atol=1e-6)
        assert int(status.item()) == 0
    assert nnz_per_slot.max() > 4                                                  # really multi-valued


# ------------------------------------------------------------------------------------------------------------------
# The hand-computed known answers of tests/test_ps_accessor_kat.py through rec_ps_push_rows (both kernels: the
# lane-per-feature one for the 'slot' layout, the row-group one for a float4 'deepfm' layout).
def _push(table, rows_pos, grad, S=1, grad1=None, click=None, show=None):
    """rows_pos: the table row of every lookup position (0 = padding)."""
    from paddlerec_amd import ops
    ids = T(np.asarray(rows_pos, np.int64).reshape(-1, S))
    groups, status = ops.ids_group(ids, table.num_rows, 0, ops.Workspace(DEV))
    ops.ps_push_rows(table, groups, T(np.asarray(grad, np.float32)), S, grad1=None if grad1 is None else T(grad1),
                     click=None if click is None else T(np.asarray(click, np.int64)),
                     show=None if show is None else T(np.asarray(show, np.int64)))
    assert int(status.item()) == 0
    return table.rec.cpu().numpy()


KAT = dict(lr=0.05, initial_g2sum=3.0, bounds=(-10.0, 10.0), initial_range=1e-2, embedx_threshold=1.0,
           nonclk_coeff=0.1, click_coeff=1.0, seed=7)


def test_kat_three_duplicate_occurrences_slot_layout(engine_lib):
    """tests/test_ps_accessor_kat.py::test_existing_value_three_duplicate_occurrences on the device: three lookup
    positions of row 2 with per-occurrence gradients summing to (0.6 | 0.3, -0.9), one of the three samples clicked."""
    from paddlerec_amd import ops
    t = ops.PsTable(4, 3, DEV, kind="slot", **KAT)
    so = t.layout.stat_off
    t.rec[2, :3] = T(np.float32([0.5, 0.1, -0.2]))
    t.rec[2, so:so + 7] = T(np.float32([5, 1, 0.0, 1.0, 2, 0.3, 4]))
    grad = np.float32([[0.1, 0.1, -0.3], [0.2, 0.1, -0.3], [0.3, 0.1, -0.3]])        # sums: 0.6 | 0.3, -0.9
    rec = _push(t, [2, 2, 2], grad, click=[0, 1, 0])
    np.testing.assert_allclose(rec[2, :3], [0.49, 0.09566987, -0.18700962], rtol=0, atol=3e-8)
    np.testing.assert_allclose(rec[2, so:so + 7], [8, 2, 0.04, 1.05, 2, 1.5, 0], rtol=0, atol=1e-7)
    assert not rec[[0, 1, 3]].any()


def test_kat_new_key_creation_sequence_slot_layout(engine_lib):
    """tests/test_ps_accessor_kat.py::test_new_key_then_embedx_creation_then_update on the device."""
    from paddlerec_amd import ops
    t = ops.PsTable(3, 3, DEV, kind="slot", **KAT)
    so = t.layout.stat_off
    rec = _push(t, [1, 1], np.float32([[-0.2, 9, 9], [-0.2, 0, 9]]), click=[0, 0])          # show 2, g_embed -0.4
    np.testing.assert_allclose(rec[1, :3], [0.01, 0, 0], atol=1e-9)
    np.testing.assert_allclose(rec[1, so:so + 7], [2, 0, 0.04, 0, 1, 0.2, 0], atol=1e-8)
    rec = _push(t, [1], np.float32([[0.3, 9, 9]]), click=[1])
    init = np.float32([ps_ref.init_value(7, 1, 1, 1e-2), ps_ref.init_value(7, 1, 2, 1e-2)])
    np.testing.assert_allclose(rec[1, 0], -0.00490099, atol=2e-9)
    assert np.array_equal(rec[1, 1:3], init)
    np.testing.assert_allclose(rec[1, so:so + 7], [3, 1, 0.13, 0, 2, 1.2, 0], atol=1e-7)
    rec = _push(t, [1], np.float32([[0.0, 0.2, 0.4]]), click=[0])
    np.testing.assert_allclose(rec[1, 1:3], init - np.float32([0.01, 0.02]), atol=2e-9)
    np.testing.assert_allclose(rec[1, so + 3], 0.1, atol=1e-8)
    assert rec[1, so] == 4 and rec[1, so + 4] == 2


def test_kat_k_duplicates_move_like_one_deepfm_layout(engine_lib):
    """The row-group kernel (float4 lanes, 'deepfm' record, D 16): a key hit 4 times with the same per-occurrence
    gradient ends where a key hit once with it ends — except for its counters; and show_scale off moves it 4x."""
    from paddlerec_amd import ops
    rng = np.random.default_rng(5)
    D = 16
    g = (rng.standard_normal(D) * 0.1).astype(np.float32)
    g1 = np.float32(0.25)
    out = {}
    for name, k, kw in (("one", 1, {}), ("four", 4, {}), ("four_noscale", 4, dict(show_scale=False))):
        t = ops.PsTable(8, D, DEV, kind="deepfm", **dict(KAT, embedx_threshold=0.0, **kw))
        so = t.layout.stat_off
        w0 = (rng.standard_normal(D + 1) * 0).astype(np.float32) + np.float32(0.5)
        t.rec[3, :D + 1] = T(w0)
        t.rec[3, so:so + 7] = T(np.float32([5, 1, 0.2, 0.7, 2, 0.0, 3]))
        dz = np.full((k, 1), g1, np.float32)          # first-order gradient per sample (one slot per sample)
        out[name] = _push(t, [3] * k, np.tile(g, (k, 1)), grad1=dz, click=[0] * k)[3].copy()
    so = D + 1
    assert np.array_equal(out["one"][:D + 1], out["four"][:D + 1])
    assert np.array_equal(out["one"][so + 2:so + 4], out["four"][so + 2:so + 4])
    assert out["one"][so] == 6 and out["four"][so] == 9
    np.testing.assert_allclose(0.5 - out["four_noscale"][:D + 1], 4 * (0.5 - out["one"][:D + 1]), rtol=2e-5, atol=2e-7)
    # and against the oracle, element by element
    lay = dict(embed_off=D, embedx_off=0, embedx_dim=D, stat_off=D + 1)
    want = np.zeros((8, 32), np.float32)
    want[3, :D + 1] = 0.5
    want[3, so:so + 7] = [5, 1, 0.2, 0.7, 2, 0.0, 3]
    ps_ref.push_rows(want, lay, [3], [4 * g1], [4 * g], [4], [0], dict(KAT, embedx_threshold=0.0))
    np.testing.assert_allclose(out["four"][:D + 1], want[3, :D + 1], rtol=2e-7, atol=0)
    np.testing.assert_allclose(out["four"][so:so + 7], want[3, so:so + 7], rtol=2e-7, atol=0)


def test_kat_float_division_typing_on_the_device(engine_lib):
    """tests/test_ps_accessor_kat.py::test_float_division_of_the_float_pushed_gradient through rec_ps_push_rows, both
    kernels: six occurrences of the key (pushed show 6) whose gradients sum to 0.82f, grad_scale 4 — the float division
    of the C++ text ends on 0.47266665 / 0.29884446, a double division on 0.47266668 / 0.29884443."""
    from paddlerec_amd import ops
    g = np.float32(0.82)
    parts = np.float32([0.82, 0, 0, 0, 0, 0])                        # merged in ascending position: exactly 0.82f
    # lane-per-feature kernel ('slot' layout, D 3)
    t = ops.PsTable(3, 3, DEV, kind="slot", **dict(KAT, embedx_threshold=1e9))
    so = t.layout.stat_off
    t.rec[1, 0] = 0.5
    t.rec[1, so:so + 7] = T(np.float32([6, 0, 0, 0, 1, 0, 0]))
    t.accessor.grad_scale = 4.0
    grad = np.zeros((6, 3), np.float32)
    grad[:, 0] = parts
    ids = T(np.full((6, 1), 1, np.int64))
    groups, _ = ops.ids_group(ids, t.num_rows, 0, ops.Workspace(DEV))
    ops.ps_push_rows(t, groups, T(grad), 1, click=T(np.zeros(6, np.int64)))
    rec = t.rec.cpu().numpy()
    assert rec[1, 0] == np.float32(0.47266665) and rec[1, so + 2] == np.float32(0.29884446)
    assert rec[1, so] == 12 and g == np.float32(grad[:, 0].sum())
    # row-group kernel ('deepfm' layout, D 16): embed_w = the first-order weight at column D
    D = 16
    t = ops.PsTable(4, D, DEV, kind="deepfm", **dict(KAT, embedx_threshold=1e9))
    so = t.layout.stat_off
    t.rec[2, D] = 0.5
    t.rec[2, so:so + 7] = T(np.float32([6, 0, 0, 0, 1, 0, 0]))
    t.accessor.grad_scale = 4.0
    ids = T(np.full((6, 1), 2, np.int64))
    groups, _ = ops.ids_group(ids, t.num_rows, 0, ops.Workspace(DEV))
    ops.ps_push_rows(t, groups, T(np.zeros((6, D), np.float32)), 1, grad1=T(parts.reshape(6, 1)),
                     click=T(np.zeros(6, np.int64)))
    rec = t.rec.cpu().numpy()
    assert rec[2, D] == np.float32(0.47266665) and rec[2, so + 2] == np.float32(0.29884446)
