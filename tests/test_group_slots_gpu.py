"""GPU parity tests of the slot-local SelectedRows grouping (rec_ids_group_slots, csrc/ids_group_slots.hip), the rank it
emits, the sorted-order row gradient of rec_deepfm_fm_bwd_sorted and the row-update kernels reading it
(rec_grad_layout.sorted) — against the NumPy oracle (oracle/deepfm_ref.py group_ids / merge_rows / adam_update_rows) and
against the general path (rec_ids_group with slot_offset), bit for bit.

Reference behaviour: the duplicate-row merge of a `sparse=True` embedding gradient and lazy Adam on the merged rows
(/root/reference/models/rank/deepfm/net.py:62-86, deepfm/static_model.py:101-107)."""
import numpy as np
import pytest
import torch

from oracle import deepfm_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops(engine_lib):
    from paddlerec_amd import ops as o
    assert torch.cuda.is_available()
    return o


def T(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV)


def N_(t):
    return t.detach().cpu().numpy()


def make_ids(B, S, R_, seed, pad_frac=0.03, zipf=False, hot=False):
    rng = np.random.default_rng(seed)
    if zipf:
        ranks = np.minimum(rng.zipf(1.05, size=(B, S)), R_ - 1)
        perm = rng.permutation(R_)
        ids = np.clip(perm[ranks], 1, R_ - 1).astype(np.int64)
    else:
        ids = rng.integers(1, R_, size=(B, S), dtype=np.int64)
    if hot:                       # one row of slot 3 owns a fifth of the batch, one of slot 0 a run of 127 and one of 128
        ids[rng.random(B) < 0.2, 3] = 77
        ids[:127, 0] = 5
        ids[200:328, 0] = 6
    ids[rng.random((B, S)) < pad_frac] = 0
    return ids


def oracle_group(ids, R_):
    B, S = ids.shape
    so = np.arange(S, dtype=np.int64) * R_
    rows, valid = R.effective_rows(ids, 0, so)
    return R.group_ids(rows.reshape(-1), valid.reshape(-1))


def check_against_oracle(groups, ids, R_):
    spos, uniq, offs = groups.host()
    rspos, runiq, roffs = oracle_group(ids, R_)
    assert np.array_equal(spos, rspos) and np.array_equal(uniq, runiq) and np.array_equal(offs, roffs)
    U, nv, has_long, _ = groups.n_uniq.tolist()
    assert U == len(runiq) and nv == len(rspos)
    assert has_long == int((np.diff(roffs) >= 128).any())
    if groups.rank is not None:
        rank = N_(groups.rank)
        want = np.full(ids.size, -1, np.int32)
        want[rspos] = np.arange(len(rspos), dtype=np.int32)
        assert np.array_equal(rank, want)


# slot-local path (B >= 8192, R <= 2^20, S <= 60): one and two digits, ragged last chunk, hot rows, all-padding slot
@pytest.mark.parametrize("B,S,R_,kw", [
    (8192, 26, 1000, dict()),                        # one 10-bit digit
    (8192, 3, 2, dict(pad_frac=0.5)),                # ids in {0 (padding), 1}: one huge segment per slot
    (10000, 26, 1_000_000, dict()),                  # two digits, ragged second chunk
    (16384, 7, 1 << 20, dict(zipf=True)),            # the widest key (20 bits), Zipf ids
    (12345, 26, 5000, dict(hot=True)),               # 13-bit keys: digits 7 + 6; hot rows, 127 / 128 runs
    (65536, 26, 1_000_000, dict()),                  # BASELINE configs[1]
    (8192, 60, 1025, dict(pad_frac=0.3)),            # most slots the transpose tile takes; 11-bit keys: 6 + 5
])
def test_group_slots_bit_exact(ops, B, S, R_, kw):
    ids = make_ids(B, S, R_, seed=B + S, **kw)
    if S >= 7:
        ids[:, S - 2] = 0                             # a slot without a single key
    ws = ops.Workspace(DEV)
    groups, status = ops.ids_group_slots(T(ids), R_, 0, ws, want_rank=True)
    assert int(status.item()) == 0
    check_against_oracle(groups, ids, R_)
    # ... and the general path (one 25-bit key sort over slot_offset) gives the same grouping
    g2, _ = ops.ids_group(T(ids), S * R_, 0, ops.Workspace(DEV), T(np.arange(S, dtype=np.int64) * R_))
    for a, b in zip(groups.host(), g2.host()):
        assert np.array_equal(a, b)
    assert groups.n_uniq.tolist() == g2.n_uniq.tolist()
    assert np.array_equal(N_(ops.ids_rank(g2)), N_(groups.rank))


# shapes the slot-local path does not cover take the general sort inside the same entry point
@pytest.mark.parametrize("B,S,R_", [(1, 26, 50), (333, 26, 1000), (4096, 4, 3_000_000), (9000, 2, 2_000_000_000)])
def test_group_slots_fallback_shapes(ops, B, S, R_):
    ids = make_ids(B, S, min(R_, 100000), seed=B)
    groups, status = ops.ids_group_slots(T(ids), R_, 0, ops.Workspace(DEV), want_rank=True)
    assert int(status.item()) == 0
    check_against_oracle(groups, ids, R_)


def test_group_slots_edge_cases(ops):
    ws = ops.Workspace(DEV)
    B, S, R_ = 8192, 26, 100000
    groups, _ = ops.ids_group_slots(torch.zeros(B, S, dtype=torch.int64, device=DEV), R_, 0, ws, want_rank=True)
    assert groups.n_uniq.tolist() == [0, 0, 0, 0] and int(groups.seg_offset[0]) == 0      # nothing but padding
    assert bool((groups.rank == -1).all())
    groups, _ = ops.ids_group_slots(torch.zeros(0, S, dtype=torch.int64, device=DEV), R_, 0, ws)
    assert groups.n_uniq.tolist() == [0, 0, 0, 0]
    ids = make_ids(B, S, R_, seed=5)
    ids[17, 3] = R_            # the first id of the NEXT slot's range: with one table per slot it is out of range
    ids[99, 0] = -4
    groups, status = ops.ids_group_slots(T(ids), R_, 0, ws, want_rank=True)
    assert int(status.item()) & 1
    ok = ids.copy()
    ok[17, 3] = 0
    ok[99, 0] = 0
    check_against_oracle(groups, ok, R_)
    # no padding id at all (padding_idx None): id 0 is a row like any other
    ids = make_ids(B, 4, 64, seed=9, pad_frac=0.2)
    groups, _ = ops.ids_group_slots(T(ids), 64, None, ws)
    so = np.arange(4, dtype=np.int64) * 64
    rspos, runiq, roffs = R.group_ids((ids + so).reshape(-1), np.ones(ids.size, bool))
    spos, uniq, offs = groups.host()
    assert np.array_equal(spos, rspos) and np.array_equal(uniq, runiq) and np.array_equal(offs, roffs)


def test_group_slots_is_stable_under_a_busy_chip(ops):
    """The kernels hold no inter-block waits; the same batch grouped while another stream keeps every CU busy must
    give the same bits (uneven load is where an ordering bug between the passes would show)."""
    B, S, R_ = 32768, 26, 1_000_000
    ids = T(make_ids(B, S, R_, seed=3, zipf=True))
    ws = ops.Workspace(DEV)
    ref, _ = ops.ids_group_slots(ids, R_, 0, ws, want_rank=True)
    want = [t.clone() for t in (ref.sorted_pos, ref.uniq_rows, ref.seg_offset, ref.n_uniq, ref.rank)]
    nv, U = int(ref.n_uniq[1]), int(ref.n_uniq[0])
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=DEV)
    for _ in range(5):
        with torch.cuda.stream(side):
            for _ in range(4):
                a = torch.tanh(a @ a * 1e-3)
        g, _ = ops.ids_group_slots(ids, R_, 0, ws, want_rank=True)
        torch.cuda.synchronize()
        assert torch.equal(g.sorted_pos[:nv], want[0][:nv]) and torch.equal(g.uniq_rows[:U], want[1][:U])
        assert torch.equal(g.seg_offset[:U + 1], want[2][:U + 1]) and torch.equal(g.n_uniq, want[3])
        assert torch.equal(g.rank, want[4])


# ------------------------------------------------------------------------------ sorted-order row gradients
@pytest.mark.parametrize("B,R_,zipf", [(8192, 50000, False), (8192, 3000, True)])
def test_fm_bwd_sorted_and_record_update_match_position_order(ops, B, R_, zipf):
    """rec_deepfm_fm_bwd_sorted writes row k of the gradient where rec_deepfm_fm_bwd's row sorted_pos[k] is, and
    rec_sparse_adam_record / rec_segment_partials with rec_grad_layout.sorted give the bits of the position-order
    path (same summation order: ascending position inside a row's segment)."""
    S, Dn, D = 26, 13, 16
    rng = np.random.default_rng(B + R_)
    ids = make_ids(B, S, R_, seed=1, zipf=zipf, hot=True)
    dense = rng.random((B, Dn), dtype=np.float32)
    N = S * R_
    std = 0.1 / np.sqrt(D)
    rec = torch.zeros(N, 32, device=DEV)
    rec[:, :D + 1] = torch.randn(N, D + 1, device=DEV) * std
    dense_w = T((rng.standard_normal((1, Dn, D)) * std).astype(np.float32))
    dense_w1 = T((rng.standard_normal(Dn) * std).astype(np.float32))
    so = T(np.arange(S, dtype=np.int64) * R_)
    tids, tdense = T(ids), T(dense)
    y1, y2, feat, sum_emb, status = ops.deepfm_fm_fwd(tids, tdense, rec[:, :D], rec[:, D:D + 1], dense_w, dense_w1, 0, so,
                                                      compact=True)
    dfeat = T((rng.standard_normal((B, S + 1, D)) * 1e-2).astype(np.float32))
    dz = T((rng.standard_normal((B, 1)) * 1e-2).astype(np.float32))
    ws, wsg = ops.Workspace(DEV), ops.Workspace(DEV)
    groups, _ = ops.ids_group_slots(tids, R_, 0, wsg, want_rank=True)
    nv = int(groups.n_uniq[1])
    rg, ddw, ddw1 = ops.deepfm_fm_bwd(tdense, feat, sum_emb, dfeat, dz, dz, S, ws, dense_w=dense_w, compact=True)
    rg, ddw, ddw1 = rg.clone(), ddw.clone(), ddw1.clone()
    rgs = torch.full((B * S, D), float("nan"), device=DEV)
    _, ddw_s, ddw1_s = ops.deepfm_fm_bwd(tdense, feat, sum_emb, dfeat, dz, dz, S, ws, out=(rgs, torch.empty_like(ddw),
                                         torch.empty_like(ddw1)), dense_w=dense_w, compact=True, row_rank=groups.rank)
    assert torch.equal(ddw, ddw_s) and torch.equal(ddw1, ddw1_s)
    assert torch.equal(rgs[:nv], rg[groups.sorted_pos[:nv].long()])
    assert bool(torch.isnan(rgs[nv:]).all())                          # dropped lookups are not written
    # the record update from both layouts, two steps, hot rows through the tile partials
    mv = torch.zeros(N, 32, device=DEV)
    rec2, mv2 = rec.clone(), mv.clone()
    for step in (1, 2):
        pp = ops.segment_partials(groups, rg, D)
        pp1 = ops.segment_partials(groups, dz, 1, grad_div=S)
        ops.sparse_adam_record(groups, rg, dz, S, rec, mv, D, step, 1e-3, v_offset=16, partials=pp, partials1=pp1)
        pps = ops.segment_partials(groups, rgs, D, grad_sorted=True)
        ops.sparse_adam_record(groups, rgs, dz, S, rec2, mv2, D, step, 1e-3, v_offset=16, partials=pps, partials1=pp1,
                               grad_sorted=True)
    assert torch.equal(rec, rec2) and torch.equal(mv, mv2)
    # ... and against the oracle's merge + lazy Adam on W (moments: they do not amplify an eps-sized sign flip)
    rows, valid = R.effective_rows(ids, 0, N_(so))
    uniq, merged, _ = R.merge_rows(rows.reshape(-1), valid.reshape(-1), N_(rg))
    M = np.zeros((N, D), np.float32)
    V = np.zeros((N, D), np.float32)
    P = np.zeros((N, D), np.float32)
    for step in (1, 2):
        R.adam_update_rows(P, M, V, uniq, merged, step, lr=1e-3)
    np.testing.assert_allclose(N_(mv2[:, :D]), M, rtol=1e-5, atol=1e-5 * float(np.abs(M).max()))
    np.testing.assert_allclose(N_(mv2[:, 16:16 + D]), V, rtol=1e-5, atol=1e-5 * float(np.abs(V).max()))


def test_deepfm_layer_slot_path_equals_general_path(ops, monkeypatch):
    """DeepFMLayer with 26 tables as one (BASELINE configs[1] layout) at a batch the slot-local grouping takes: three train
    steps through rec_ids_group_slots (+ rec_deepfm_fm_bwd_sorted + rec_grad_layout.sorted with REC_DEEPFM_SORTED_GRAD=1)
    leave every parameter, moment and prediction bit-identical to the general path (REC_DEEPFM_GROUP=general: one 25-bit
    sort, gradients in position order)
    — and the first step's loss / dense gradient agree with the NumPy oracle."""
    from paddlerec_amd.deepfm import DeepFMLayer
    from helpers import make_deepfm_problem
    B, S, R_, D, fc = 8192, 26, 2000, 16, [64, 32]
    pr = make_deepfm_problem(B=B, N=R_, D=D, fc=fc, seed=11, tables=True, zipf=True)
    p = pr["params"]
    sd = {"fm.embedding.weight": p["W"], "fm.embedding_one.weight": p["W1"], "fm.dense_w": p["dense_w"],
          "fm.dense_w_one": p["dense_w_one"]}
    for i in range(len(fc) + 1):
        sd["dnn.linear_%d.weight" % i] = p["mlp_w"][i]
        sd["dnn.linear_%d.bias" % i] = p["mlp_b"][i]
    batches = [make_deepfm_problem(B=B, N=R_, D=D, fc=fc, seed=20 + i, tables=True, zipf=(i == 1)) for i in range(3)]

    def run(flag):
        monkeypatch.setenv("REC_DEEPFM_GROUP", "slots" if flag != "0" else "general")
        monkeypatch.setenv("REC_DEEPFM_SORTED_GRAD", "1" if flag == "1" else "0")
        monkeypatch.setenv("REC_STEP_PLAN", "0")
        m = DeepFMLayer(R_ * S, D, 13, S, fc, device=DEV, slot_offset=pr["slot_offsets"])
        assert m.slot_rows == R_
        m.set_dict(sd)
        outs = []
        for b in batches:
            loss, pred = m.train_step(T(b["ids"]), T(b["dense"]), T(b["label"]), lr=1e-2)
            outs.append((loss.clone(), pred.clone()))
        torch.cuda.synchronize()
        assert int(m.status.item()) == 0
        return m, outs

    m1, o1 = run("1")
    assert m1._groups.rank is not None                    # slot-local grouping + sorted row gradients
    m0, o0 = run("0")
    assert m0._groups.rank is None                        # the general 25-bit sort, gradients in position order
    m2, o2 = run("2")
    assert m2._groups.rank is None                        # the default: slot-local grouping, gradients in position order
    for mx, ox in ((m0, o0), (m2, o2)):
        for (l1, p1), (l0, p0) in zip(o1, ox):
            assert torch.equal(l1, l0) and torch.equal(p1, p0)
        assert torch.equal(m1.fm.rec, mx.fm.rec) and torch.equal(m1.sparse_state["mv"], mx.sparse_state["mv"])
        assert torch.equal(m1.dense.data, mx.dense.data) and torch.equal(m1.dense.m, mx.dense.m)
    o = R.deepfm_loss_and_grads(batches[0]["ids"], batches[0]["dense"], batches[0]["label"], p,
                                slot_offsets=pr["slot_offsets"])
    np.testing.assert_allclose(float(o1[0][0].item()), o["loss"], rtol=1e-5)
