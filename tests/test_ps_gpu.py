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


@pytest.mark.first_hw_run
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
