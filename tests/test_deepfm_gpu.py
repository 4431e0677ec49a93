"""GPU parity tests: every HIP kernel on the DeepFM path, called through the C-ABI, against the
oracle (oracle/deepfm_ref.py, oracle/deepfm_oracle.c) and the golden fixtures.

Bars (north star): indices / pooling counts / histograms bit-exact; fp32 within 1e-5 relative.
`ATOL` bounds the fp32 summation-order noise of values that are sums of O(40..1e5) terms of
magnitude <= ~0.1 (measured against the float64 oracle in test_fp32_noise_floor).
"""
import numpy as np
import pytest
import torch

from helpers import deepfm_params_from_golden, load_golden, make_deepfm_problem
from oracle import deepfm_ref as R

pytestmark = pytest.mark.gpu
RTOL = 1e-5
ATOL = 2e-7
DEV = "cuda"


def assert_close_scaled(got, want, rel=1e-5):
    """1e-5 relative, with the absolute floor tied to the TENSOR's scale (1e-5 x max|want|): elements that are sums
    of opposite-sign terms carry the fp32 summation noise of the terms, not of the (cancelled) result."""
    want = np.asarray(want)
    np.testing.assert_allclose(got, want, rtol=rel, atol=rel * float(np.abs(want).max()))


@pytest.fixture(scope="module")
def ops(engine_lib):
    from paddlerec_amd import ops as o
    assert torch.cuda.is_available()
    return o


def T(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV)


def N_(t):
    return t.detach().cpu().numpy()


def run_fm_fwd(ops, pr):
    p = pr["params"]
    so = T(pr["slot_offsets"]) if pr.get("slot_offsets") is not None else None
    y1, y2, feat, sum_emb, status = ops.deepfm_fm_fwd(
        T(pr["ids"]), T(pr["dense"]), T(p["W"]), T(p["W1"]), T(p["dense_w"]), T(p["dense_w_one"]),
        0, so)
    assert int(status.item()) == 0
    return y1, y2, feat, sum_emb


# ------------------------------------------------------------------------------ forward
@pytest.mark.parametrize("name", ["deepfm_D9", "deepfm_D16"])
def test_fm_fwd_golden(ops, name):
    g = load_golden(name)
    pr = dict(ids=g["ids"], dense=g["dense"], params=deepfm_params_from_golden(g))
    y1, y2, feat, sum_emb = run_fm_fwd(ops, pr)
    assert np.array_equal(N_(feat), g["feat"])                      # gather + multiply: bit-exact
    np.testing.assert_allclose(N_(y1), g["y1"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(N_(y2), g["y2"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(N_(sum_emb), g["feat"].sum(1), rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("B,D,zipf,tables", [
    (1, 16, False, False), (63, 16, False, False), (1000, 16, True, False), (513, 16, False, True),
    (257, 9, False, False), (300, 10, True, False), (130, 40, False, False), (65, 64, False, False),
    (70, 7, False, False), (40, 128, False, False)])
def test_fm_fwd_vs_oracle(ops, B, D, zipf, tables):
    pr = make_deepfm_problem(B=B, D=D, N=3000, seed=B + D, zipf=zipf, tables=tables)
    p = pr["params"]
    y1, y2, feat, sum_emb = run_fm_fwd(ops, pr)
    ry1, ry2, rfeat = R.fm_forward(pr["ids"], pr["dense"], p["W1"], p["W"], p["dense_w_one"],
                                   p["dense_w"], 0, pr["slot_offsets"])
    assert np.array_equal(N_(feat), rfeat)
    np.testing.assert_allclose(N_(y1), ry1, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(N_(y2), ry2, rtol=RTOL, atol=ATOL * D)
    np.testing.assert_allclose(N_(sum_emb), rfeat.sum(1), rtol=RTOL, atol=ATOL)


def test_fm_fwd_edge_cases(ops):
    # all-padding batch: zero sparse rows, y1/y2 come from the dense fields only
    pr = make_deepfm_problem(B=32, N=100, seed=1)
    pr["ids"][:] = 0
    y1, y2, feat, _ = run_fm_fwd(ops, pr)
    assert np.all(N_(feat)[:, :26] == 0)
    ry1, ry2, _ = R.fm_forward(pr["ids"], pr["dense"], pr["params"]["W1"], pr["params"]["W"],
                               pr["params"]["dense_w_one"], pr["params"]["dense_w"])
    np.testing.assert_allclose(N_(y1), ry1, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(N_(y2), ry2, rtol=RTOL, atol=ATOL)
    # empty batch is a no-op
    e = ops.deepfm_fm_fwd(torch.zeros(0, 26, dtype=torch.int64, device=DEV),
                          torch.zeros(0, 13, device=DEV), T(pr["params"]["W"]), T(pr["params"]["W1"]),
                          T(pr["params"]["dense_w"]), T(pr["params"]["dense_w_one"]))
    assert e[2].shape == (0, 39, 16)
    # out-of-range id: flagged in the device status word, row treated as zero, no crash
    pr2 = make_deepfm_problem(B=8, N=100, seed=2)
    pr2["ids"][3, 5] = 100
    pr2["ids"][4, 6] = -7
    p = pr2["params"]
    _, _, feat2, _, status = ops.deepfm_fm_fwd(T(pr2["ids"]), T(pr2["dense"]), T(p["W"]), T(p["W1"]),
                                               T(p["dense_w"]), T(p["dense_w_one"]))
    assert int(status.item()) & 1
    assert np.all(N_(feat2)[3, 5] == 0) and np.all(N_(feat2)[4, 6] == 0)
    with pytest.raises(Exception):
        ops.raise_on_status(status)


def test_no_cpu_fallback(ops):
    pr = make_deepfm_problem(B=4, N=50, seed=0)
    p = pr["params"]
    with pytest.raises(Exception, match="device tensor"):
        ops.deepfm_fm_fwd(torch.as_tensor(pr["ids"]), torch.as_tensor(pr["dense"]),
                          torch.as_tensor(p["W"]), torch.as_tensor(p["W1"]),
                          torch.as_tensor(p["dense_w"]), torch.as_tensor(p["dense_w_one"]))


def test_fp32_noise_floor():
    """Justifies ATOL: float32 oracle vs float64 oracle on the same inputs."""
    pr = make_deepfm_problem(B=512, N=3000, seed=77)
    p = pr["params"]
    p64 = {k: (v.astype(np.float64) if not isinstance(v, list) else [x.astype(np.float64) for x in v])
           for k, v in p.items()}
    y1, y2, _ = R.fm_forward(pr["ids"], pr["dense"], p["W1"], p["W"], p["dense_w_one"], p["dense_w"])
    z1, z2, _ = R.fm_forward(pr["ids"], pr["dense"].astype(np.float64), p64["W1"], p64["W"],
                             p64["dense_w_one"], p64["dense_w"])
    assert np.abs(y1 - z1).max() < ATOL and np.abs(y2 - z2).max() < ATOL * 16


# ------------------------------------------------------------------------------ backward
@pytest.mark.parametrize("B,D,tables", [(1, 16, False), (777, 16, False), (2100, 16, True),
                                        (257, 9, False), (129, 40, False)])
def test_fm_bwd_vs_oracle(ops, oracle_lib, B, D, tables):
    from helpers import c_fm_bwd
    pr = make_deepfm_problem(B=B, D=D, N=3000, seed=B, tables=tables)
    y1, y2, feat, sum_emb = run_fm_fwd(ops, pr)
    rng = np.random.default_rng(B)
    dfeat = (rng.standard_normal(feat.shape) * 1e-3).astype(np.float32)
    dz = (rng.standard_normal((B, 1)) * 1e-3).astype(np.float32)
    ws = ops.Workspace(DEV)
    rg, ddw, ddw1 = ops.deepfm_fm_bwd(T(pr["dense"]), feat, sum_emb, T(dfeat), T(dz), T(dz), 26, ws)
    ref = R.fm_backward(pr["ids"], pr["dense"], N_(feat), dfeat, dz, dz, 0, pr["slot_offsets"])
    np.testing.assert_allclose(N_(rg), ref["row_grad"], rtol=RTOL, atol=1e-9)
    # batch sums: compare against the float64 oracle (summation-order independent truth)
    ref64 = R.fm_backward(pr["ids"], pr["dense"].astype(np.float64), N_(feat).astype(np.float64),
                          dfeat.astype(np.float64), dz.astype(np.float64), dz.astype(np.float64))
    scale = np.abs(ref64["d_dense_w"]).max()
    np.testing.assert_allclose(N_(ddw), ref64["d_dense_w"][0], rtol=RTOL, atol=RTOL * scale)
    np.testing.assert_allclose(N_(ddw1), ref64["d_dense_w_one"], rtol=RTOL,
                               atol=RTOL * np.abs(ref64["d_dense_w_one"]).max())
    # C oracle agrees as well (independent restatement)
    crg, _, cddw, _ = c_fm_bwd(oracle_lib, 26, pr["dense"], N_(feat), N_(sum_emb), dfeat, dz, dz)
    np.testing.assert_allclose(N_(rg), crg, rtol=RTOL, atol=1e-9)
    # deterministic: same bits on a second run
    rg2, ddw2, _ = ops.deepfm_fm_bwd(T(pr["dense"]), feat, sum_emb, T(dfeat), T(dz), T(dz), 26, ws)
    assert torch.equal(rg, rg2) and torch.equal(ddw, ddw2)
    # recomputing the dense part of feat from dense_w instead of re-reading it: same bits
    rg3, ddw3, ddw13 = ops.deepfm_fm_bwd(T(pr["dense"]), feat, sum_emb, T(dfeat), T(dz), T(dz), 26, ws,
                                         dense_w=T(pr["params"]["dense_w"]))
    assert torch.equal(rg, rg3) and torch.equal(ddw, ddw3) and torch.equal(ddw1, ddw13)


@pytest.mark.parametrize("B,D,tables", [(1, 16, False), (515, 16, True), (130, 40, False), (77, 13, False)])
def test_compact_dense_mode(ops, B, D, tables):
    """compact feat = S embedding rows + one row of raw dense values; FM outputs unchanged; the backward takes
    d_feat_dnn in the same layout and returns the FM part of d_dense_w; the folded layer-0 weights reproduce
    feat @ W0 and its gradients."""
    pr = make_deepfm_problem(B=B, D=D, N=2000, seed=B + D, tables=tables)
    p = pr["params"]
    so = T(pr["slot_offsets"]) if tables else None
    args = (T(pr["ids"]), T(pr["dense"]), T(p["W"]), T(p["W1"]), T(p["dense_w"]), T(p["dense_w_one"]), 0, so)
    y1, y2, feat, sum_emb, st = ops.deepfm_fm_fwd(*args)
    c1, c2, cfeat, csum, st2 = ops.deepfm_fm_fwd(*args, compact=True)
    assert int(st2.item()) == 0 and cfeat.shape == (B, 27, D)
    assert torch.equal(y1, c1) and torch.equal(y2, c2) and torch.equal(sum_emb, csum)
    assert torch.equal(cfeat[:, :26], feat[:, :26])
    assert np.array_equal(N_(cfeat[:, 26, :13]), pr["dense"]) and float(cfeat[:, 26, 13:].abs().sum()) == 0.0
    # folded layer 0: feat' @ [W0_sparse; M; 0] == feat @ W0
    rng = np.random.default_rng(B)
    n_out = 48
    W0 = (rng.standard_normal((39 * D, n_out)) / np.sqrt(39 * D)).astype(np.float32)
    M = torch.zeros(13, n_out, device=DEV)
    ops.dense_fold_fwd(26, T(p["dense_w"]).view(13, D), T(W0), M)
    np.testing.assert_allclose(N_(M), np.einsum("jd,jdn->jn", p["dense_w"][0], W0[26 * D:].reshape(13, D, n_out)),
                               rtol=1e-5, atol=1e-7)
    W0p = np.zeros((27 * D, n_out), np.float32)
    W0p[:26 * D] = W0[:26 * D]
    W0p[26 * D:26 * D + 13] = N_(M)
    np.testing.assert_allclose(N_(cfeat).reshape(B, -1).astype(np.float64) @ W0p,
                               N_(feat).reshape(B, -1).astype(np.float64) @ W0, rtol=1e-5, atol=1e-6)
    # backward: same row gradients when the dnn gradient of the dense fields is zero; d_dense_w = FM part
    dfeat = (rng.standard_normal((B, 39, D)) * 1e-3).astype(np.float32)
    dfeat[:, 26:] = 0
    dz = (rng.standard_normal((B, 1)) * 1e-3).astype(np.float32)
    ws = ops.Workspace(DEV)
    rg, ddw, ddw1 = ops.deepfm_fm_bwd(T(pr["dense"]), feat, sum_emb, T(dfeat), T(dz), T(dz), 26, ws)
    rg, ddw, ddw1 = rg.clone(), ddw.clone(), ddw1.clone()
    cd = np.zeros((B, 27, D), np.float32)
    cd[:, :26] = dfeat[:, :26]
    crg, cddw, cddw1 = ops.deepfm_fm_bwd(T(pr["dense"]), cfeat, csum, T(cd), T(dz), T(dz), 26, ws,
                                         dense_w=T(p["dense_w"]), compact=True)
    assert torch.equal(rg, crg) and torch.equal(ddw, cddw) and torch.equal(ddw1, cddw1)
    # fold backward vs NumPy
    dM = (rng.standard_normal((13, n_out)) * 1e-2).astype(np.float32)
    dW0 = torch.zeros(39 * D, n_out, device=DEV)
    gdw = torch.ones(13, D, device=DEV)
    ops.dense_fold_bwd(26, T(p["dense_w"]).view(13, D), T(W0), T(dM), dW0, gdw, accumulate=True)
    np.testing.assert_allclose(N_(dW0)[26 * D:], (p["dense_w"][0][:, :, None] * dM[:, None, :]).reshape(13 * D, n_out),
                               rtol=1e-6, atol=1e-9)
    assert float(dW0[:26 * D].abs().max()) == 0.0
    np.testing.assert_allclose(N_(gdw), 1 + np.einsum("jn,jdn->jd", dM, W0[26 * D:].reshape(13, D, n_out)),
                               rtol=1e-5, atol=1e-7)
    # the one-launch forms: the same bits as the copy + the separate kernels
    W0f = torch.zeros(27 * D, n_out, device=DEV)
    ops.dense_fold_fwd_full(26, T(p["dense_w"]).view(13, D), T(W0), W0f)
    assert torch.equal(W0f, T(W0p))
    dW0f = T((rng.standard_normal((27 * D, n_out)) * 1e-2).astype(np.float32))
    dW0f[26 * D + 13:] = 0
    want_dw0 = torch.zeros(39 * D, n_out, device=DEV)
    want_dw0[:26 * D] = dW0f[:26 * D]
    want_gdw = torch.ones(13, D, device=DEV)
    ops.dense_fold_bwd(26, T(p["dense_w"]).view(13, D), T(W0), dW0f[26 * D:26 * D + 13].clone(), want_dw0, want_gdw,
                       accumulate=True)
    got_dw0 = torch.full((39 * D, n_out), 7.0, device=DEV)
    got_gdw = torch.ones(13, D, device=DEV)
    ops.dense_fold_bwd_full(26, T(p["dense_w"]).view(13, D), T(W0), dW0f, got_dw0, got_gdw, accumulate=True)
    assert torch.equal(got_dw0, want_dw0) and torch.equal(got_gdw, want_gdw)


# ------------------------------------------------------------------------------ ids grouping
@pytest.mark.parametrize("B,N,pad_frac,zipf,tables", [
    (1, 50, 0.0, False, False), (64, 50, 0.2, False, False), (1000, 100000, 0.03, True, False),
    (4096, 1000, 0.03, False, True), (333, 7, 0.5, False, False)])
def test_ids_group_bit_exact(ops, B, N, pad_frac, zipf, tables):
    pr = make_deepfm_problem(B=B, N=N, seed=B, pad_frac=pad_frac, zipf=zipf, tables=tables)
    so = T(pr["slot_offsets"]) if tables else None
    ws = ops.Workspace(DEV)
    groups, status = ops.ids_group(T(pr["ids"]), pr["N"], 0, ws, so)
    spos, uniq, offs = groups.host()
    rows, valid = R.effective_rows(pr["ids"], 0, pr["slot_offsets"])
    rspos, runiq, roffs = R.group_ids(rows.reshape(-1), valid.reshape(-1))
    assert int(status.item()) == 0
    assert np.array_equal(spos, rspos) and np.array_equal(uniq, runiq) and np.array_equal(offs, roffs)


@pytest.mark.parametrize("N", [3000, 5_000_000_000])
def test_ids_group_payload_travels_with_the_lookup(ops, N):
    """rec_ids_group_payload: sorted_pos[k] = payload[position] — same rows, same segments, stable in the positions
    (32- and 64-bit key paths)."""
    rng = np.random.default_rng(17)
    n = 20011
    ids = rng.integers(0, min(N, 4000), size=n).astype(np.int64)            # ~15 % duplicates per row, some padding 0
    payload = rng.integers(0, 2 ** 31 - 1, size=n).astype(np.int32)
    ws = ops.Workspace(DEV)
    g0, _ = ops.ids_group(T(ids), N, 0, ws)
    spos0, uniq0, offs0 = g0.host()
    g1, st = ops.ids_group(T(ids), N, 0, ws, payload=T(payload))
    spos1, uniq1, offs1 = g1.host()
    assert int(st.item()) == 0
    assert np.array_equal(uniq0, uniq1) and np.array_equal(offs0, offs1)
    assert np.array_equal(spos1, payload[spos0])
    rspos, runiq, roffs = R.group_ids(ids, ids != 0)
    assert np.array_equal(spos0, rspos) and np.array_equal(uniq0, runiq) and np.array_equal(offs0, roffs)


def test_ids_group_edge_cases(ops):
    ws = ops.Workspace(DEV)
    ids = torch.zeros(16, 26, dtype=torch.int64, device=DEV)          # nothing but padding
    groups, _ = ops.ids_group(ids, 100, 0, ws)
    assert groups.n_uniq.tolist() == [0, 0, 0, 0]                      # {rows, positions, has a long segment, -}
    groups, _ = ops.ids_group(torch.zeros(0, 26, dtype=torch.int64, device=DEV), 100, 0, ws)
    assert groups.n_uniq.tolist() == [0, 0, 0, 0]
    ids = torch.full((5, 26), 7, dtype=torch.int64, device=DEV)       # one row, 130 duplicates
    groups, _ = ops.ids_group(ids, 100, 0, ws)
    spos, uniq, offs = groups.host()
    assert uniq.tolist() == [7] and offs.tolist() == [0, 130] and spos.tolist() == list(range(130))
    ids[2, 3] = 1000                                                   # out of range -> flagged, dropped
    groups, status = ops.ids_group(ids, 100, 0, ws)
    assert int(status.item()) & 1 and groups.n_uniq.tolist() == [1, 129, 1, 0]
    big = torch.randint(1, 3_000_000_000, (64, 26), dtype=torch.int64, device=DEV)   # 64-bit key path
    groups, _ = ops.ids_group(big, 5_000_000_000, 0, ws)
    spos, uniq, offs = groups.host()
    rspos, runiq, roffs = R.group_ids(N_(big).reshape(-1), np.ones(64 * 26, bool))
    assert np.array_equal(spos, rspos) and np.array_equal(uniq, runiq) and np.array_equal(offs, roffs)


# ------------------------------------------------------------------------------ optimizers
@pytest.mark.parametrize("D,zipf", [(16, False), (16, True), (9, False), (1, False), (40, True)])
def test_sparse_adam_rows_vs_oracle(ops, D, zipf):
    B, S, N = 600, 26, 500
    pr = make_deepfm_problem(B=B, N=N, D=max(D, 2), seed=D, zipf=zipf, pad_frac=0.05)
    rng = np.random.default_rng(D)
    rows, valid = R.effective_rows(pr["ids"])
    P = (rng.standard_normal((N, D)) * 0.1).astype(np.float32)
    M = (rng.standard_normal((N, D)) * 0.01).astype(np.float32)
    V = (np.abs(rng.standard_normal((N, D))) * 1e-4).astype(np.float32)
    grad = (rng.standard_normal((B * S, D)) * 1e-2).astype(np.float32)
    ws = ops.Workspace(DEV)
    groups, _ = ops.ids_group(T(pr["ids"]), N, 0, ws)
    tP, tM, tV = T(P), T(M), T(V)
    uniq, merged, _ = R.merge_rows(rows.reshape(-1), valid.reshape(-1), grad)
    for step in (1, 2, 7):
        ops.sparse_adam_rows(groups, T(grad), 1, tP, tM, tV, step, lr=1e-3)
        R.adam_update_rows(P, M, V, uniq, merged, step, lr=1e-3)
    # atol: M ~ 1e-2 and V ~ 1e-4 are sums of opposite-sign terms; 1e-5 of their scale
    np.testing.assert_allclose(N_(tM), M, rtol=RTOL, atol=1e-7)
    np.testing.assert_allclose(N_(tV), V, rtol=RTOL, atol=1e-9)
    np.testing.assert_allclose(N_(tP), P, rtol=RTOL, atol=1e-6)
    untouched = np.setdiff1d(np.arange(N), uniq)
    assert np.array_equal(N_(tP)[untouched], P[untouched])            # lazy: other rows untouched


def _hot_row_ids(rng, pad=37):
    """Flattened ids whose sorted segments hit every tile case of rec_segment_partials (REC_SEG_TILE = 64,
    REC_SEG_LONG = 128): long segments that start / end on and off tile boundaries, a 127-long (short) one,
    long-short-long inside neighbouring tiles, a ragged tail, padding ids in between."""
    lens = [5, 128, 59, 128, 1, 127, 300, 64, 1000, 2, 2500, 17, 129, 3, 640, 1]
    ids = np.concatenate([np.full(L, r + 1, np.int64) for r, L in enumerate(lens)] + [np.zeros(pad, np.int64)])
    return rng.permutation(ids), len(lens) + 1


@pytest.mark.parametrize("D,layout", [(16, "rows"), (1, "div"), (9, "rows"), (40, "group"), (64, "rows")])
def test_hot_rows_segment_partials(ops, D, layout):
    """Merged gradient through the tile partials == position-by-position merge == float64 sum (1e-5)."""
    rng = np.random.default_rng(D)
    ids, N = _hot_row_ids(rng)
    n = ids.size
    S = 3 if layout == "div" else 1
    if layout == "div":
        ids = ids[: n // S * S]                                   # gradient row of position p is grad[p // S]
        n = ids.size
        grad = rng.standard_normal((n // S, 1)).astype(np.float32)
        kw = dict(grad_div=S)
        rows_of = lambda g: np.repeat(g, S, 0)
    elif layout == "group":
        stride = D + 24
        buf = rng.standard_normal((n, stride)).astype(np.float32)
        grad, kw = buf, dict(grad_group=1, grad_group_stride=stride)
        rows_of = lambda g: g[:, :D]
    else:
        grad, kw = rng.standard_normal((n, D)).astype(np.float32), {}
        rows_of = lambda g: g
    ws = ops.Workspace(DEV)
    groups, _ = ops.ids_group(T(ids.reshape(-1, 1)), N, 0, ws)
    tg = T(grad)
    pp = ops.segment_partials(groups, tg, D, **kw)
    out = []
    for partials in (None, pp):
        P = torch.zeros(N, D, dtype=torch.float32, device=DEV)
        ops.sparse_sgd_rows(groups, tg, P, -1.0, partials=partials, **kw)      # P = merged gradient
        out.append(N_(P))
    ref = np.zeros((N, D), np.float64)
    np.add.at(ref, ids[ids != 0], rows_of(grad).astype(np.float64)[ids != 0])
    mag = np.zeros((N, D), np.float64)
    np.add.at(mag, ids[ids != 0], np.abs(rows_of(grad).astype(np.float64))[ids != 0])
    for got in out:
        assert np.all(np.abs(got - ref) <= 1e-6 * mag + 1e-6)                # fp32 summation bound, any order
        assert np.array_equal(got[0], np.zeros(D, np.float32))
    # the other consumers take the same partials: Adam (lazy + all rows) and the clipping norm agree both ways
    res = []
    for partials in (None, pp):
        Pm = [torch.zeros(N, D, dtype=torch.float32, device=DEV) for _ in range(3)]
        ops.sparse_adam_rows(groups, tg, kw.get("grad_div", 1), *Pm, 1, lr=1e-3, partials=partials,
                             **{k_: v for k_, v in kw.items() if k_ != "grad_div"})
        Pa = [torch.zeros(N, D, dtype=torch.float32, device=DEV) for _ in range(3)]
        ops.adam_rows_all(groups, tg, kw.get("grad_div", 1), *Pa, 1, lr=1e-3, partials=partials,
                          **{k_: v for k_, v in kw.items() if k_ != "grad_div"})
        ss = torch.zeros(1, dtype=torch.float32, device=DEV)
        ops.sparse_rows_sumsq(groups, tg, D, ss, ws, partials=partials, **kw)
        res.append((N_(Pm[1]), N_(Pa[1]), float(ss.item())))
    for j in (0, 1):                                                           # first moment = 0.1 * merged g
        assert np.all(np.abs(res[1][j] - res[0][j]) <= 1e-7 * mag + 1e-7)
        assert np.all(np.abs(res[1][j] - 0.1 * ref) <= 1e-7 * mag + 1e-7)
    assert abs(res[1][2] - res[0][2]) <= 1e-5 * res[0][2]
    np.testing.assert_allclose(res[1][2], float((ref ** 2).sum()), rtol=1e-5)


def test_sparse_adam_first_order_via_grad_div(ops):
    """embedding_one: SelectedRows.value is dy1[b] for all S slots of sample b (grad_div = S)."""
    B, S, N = 300, 26, 200
    pr = make_deepfm_problem(B=B, N=N, seed=4, pad_frac=0.1)
    rng = np.random.default_rng(4)
    dz = (rng.standard_normal((B, 1)) * 1e-2).astype(np.float32)
    rows, valid = R.effective_rows(pr["ids"])
    P = (rng.standard_normal((N, 1)) * 0.1).astype(np.float32)
    M, V = np.zeros_like(P), np.zeros_like(P)
    ws = ops.Workspace(DEV)
    groups, _ = ops.ids_group(T(pr["ids"]), N, 0, ws)
    tP, tM, tV = T(P), T(M), T(V)
    ops.sparse_adam_rows(groups, T(dz), S, tP, tM, tV, 1)
    uniq, merged, _ = R.merge_rows(rows.reshape(-1), valid.reshape(-1), np.repeat(dz, S, 1).reshape(-1, 1))
    R.adam_update_rows(P, M, V, uniq, merged, 1)
    np.testing.assert_allclose(N_(tP), P, rtol=RTOL, atol=1e-7)
    np.testing.assert_allclose(N_(tM), M, rtol=RTOL, atol=1e-10)


@pytest.mark.parametrize("D", [16, 9, 1])
def test_adam_rows_all_nonlazy_vs_oracle(ops, D):
    """lazy_mode=False (dygraph default, App. B-3): absent rows decay too — oracle adam_update_dense_equivalent."""
    pr = make_deepfm_problem(B=300, N=700, D=D, seed=D + 3, pad_frac=0.1)
    rng = np.random.default_rng(D)
    n = pr["ids"].size
    grad = (rng.standard_normal((n, D)) * 1e-2).astype(np.float32)
    P = pr["params"]["W"].copy()
    M = (rng.standard_normal(P.shape) * 1e-3).astype(np.float32)
    V = (rng.random(P.shape) * 1e-5).astype(np.float32)
    tp, tm, tv = T(P), T(M), T(V)
    ws = ops.Workspace(DEV)
    groups, _ = ops.ids_group(T(pr["ids"]), 700, 0, ws)
    ops.adam_rows_all(groups, T(grad), 1, tp, tm, tv, 3, lr=1e-2)
    rows, valid = R.effective_rows(pr["ids"], 0)
    uniq, merged, _ = R.merge_rows(rows.reshape(-1), valid.reshape(-1), grad)
    R.adam_update_dense_equivalent(P, M, V, uniq, merged, 3, lr=1e-2)
    np.testing.assert_allclose(N_(tp), P, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(N_(tm), M, rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(N_(tv), V, rtol=1e-5, atol=1e-12)


@pytest.mark.parametrize("D,stride", [(16, 32), (9, 16), (10, 16), (40, 64)])
def test_adam_record_all_equals_two_sweeps(ops, D, stride):
    """rec_adam_record_all: the dygraph-default (non-lazy) Adam on both embeddings of every row of the record layout in ONE
    sweep == rec_adam_rows_all on W and again on W1 / m1 / v1 (the oracle-checked path above), bit for bit; the pad floats
    of the record stay untouched."""
    rng = np.random.default_rng(D)
    N, B, S = 1500, 200, 26
    ids = rng.integers(0, N, (B, S)).astype(np.int64)
    ids[rng.random((B, S)) < 0.1] = 0
    ids[:, 0] = 7
    grad = T((rng.standard_normal((B * S, D)) * 1e-2).astype(np.float32))
    dz = T((rng.standard_normal(B) * 1e-2).astype(np.float32))
    Dp = (D + 3) // 4 * 4
    rec = T(rng.standard_normal((N, stride)).astype(np.float32))
    rec[:, D + 1:D + 3] = rec[:, D + 1:D + 3].abs() * 1e-3
    mv = T((rng.random((N, -(-2 * Dp // 32) * 32)) * 1e-3).astype(np.float32))
    rec2, mv2 = rec.clone(), mv.clone()
    ws = ops.Workspace(DEV)
    groups, _ = ops.ids_group(T(ids), N, 0, ws)
    pp = ops.segment_partials(groups, grad, D)
    pp1 = ops.segment_partials(groups, dz, 1, grad_div=S)
    for step in (1, 2):
        ops.adam_record_all(groups, grad, dz, S, rec, mv, D, step, lr=1e-2, v_offset=Dp, partials=pp, partials1=pp1)
        ops.adam_rows_all(groups, grad, 1, rec2[:, :D], mv2[:, :D], mv2[:, Dp:Dp + D], step, lr=1e-2, partials=pp)
        ops.adam_rows_all(groups, dz, S, rec2[:, D:D + 1], rec2[:, D + 1:D + 2], rec2[:, D + 2:D + 3], step, lr=1e-2,
                          partials=pp1)
    assert torch.equal(rec, rec2) and torch.equal(mv, mv2)


def test_adam_dense_vs_oracle(ops):
    rng = np.random.default_rng(0)
    n = 100003
    p = rng.standard_normal(n).astype(np.float32)
    m = np.zeros(n, np.float32)
    v = np.zeros(n, np.float32)
    tp, tm, tv = T(p), T(m), T(v)
    for step in range(1, 5):
        g = (rng.standard_normal(n) * 0.1).astype(np.float32)
        ops.adam_dense(tp, tm, tv, T(g), step)
        R.adam_update(p, m, v, g, step)
    np.testing.assert_allclose(N_(tp), p, rtol=RTOL, atol=1e-7)
    np.testing.assert_allclose(N_(tv), v, rtol=RTOL, atol=1e-12)


# ------------------------------------------------------------------------------ loss head / AUC
def test_sigmoid_logloss_and_auc(ops):
    rng = np.random.default_rng(3)
    B = 10007
    y1, y2, y3 = [(rng.standard_normal((B, 1)) * 2).astype(np.float32) for _ in range(3)]
    y1[:3] = [[40.0], [-40.0], [0.0]]                                   # saturated logits
    y2[:3] = 0; y3[:3] = 0
    label = (rng.random((B, 1)) < 0.3).astype(np.int64)
    ws = ops.Workspace(DEV)
    pred, dz, loss = ops.sigmoid_logloss(T(y1), T(y2), T(y3), T(label), ws)
    z = (y1 + y2 + y3).astype(np.float64)
    p64 = R.sigmoid(z)
    np.testing.assert_allclose(N_(pred), p64, rtol=RTOL, atol=1e-7)
    np.testing.assert_allclose(N_(loss)[0], R.log_loss_mean(p64, label), rtol=RTOL)
    # dz is evaluated from the float32 pred, as the reference does (sigmoid output feeds log_loss): the reference value is
    # the formula in float64 ON that float32 pred; the bar is 1e-5 of it, or 4 x what evaluating the SAME formula in
    # float32 costs the oracle itself (the 1 - p cancellation), element by element — a measured floor, not a looser rtol
    p32 = N_(pred)
    want64 = R.log_loss_mean_grad_z(p32.astype(np.float64), label)
    want32 = R.log_loss_mean_grad_z(p32, label).astype(np.float64)
    err = np.abs(N_(dz).astype(np.float64) - want64)
    bound = np.maximum(1e-5 * np.abs(want64), 4.0 * np.abs(want32 - want64)) + 1e-12
    assert np.all(err <= bound), (float((err / bound).max()), float(err.max()))
    ok = np.abs(z) < 4                      # away from saturation the float64 truth on the float64 pred agrees too:
    truth = R.log_loss_mean_grad_z(p64, label)            # what is left is the float32 rounding of pred itself,
    slack = np.abs(R.log_loss_mean_grad_z(p32.astype(np.float64) + 6e-8, label) - want64)     # one ulp of p around 0.5
    assert np.all(np.abs(N_(dz).astype(np.float64) - truth)[ok] <= (bound + 2.0 * slack)[ok])
    # AUC histogram: integer counts, bit-exact against the oracle on the SAME float32 predictions
    pos = torch.zeros(4096, dtype=torch.int64, device=DEV)
    neg = torch.zeros(4096, dtype=torch.int64, device=DEV)
    ops.auc_histogram(pred, T(label), pos, neg)
    ops.auc_histogram(pred, T(label), pos, neg)                         # accumulates
    rpos, rneg = R.auc_histogram(N_(pred), label)
    assert np.array_equal(N_(pos), 2 * rpos) and np.array_equal(N_(neg), 2 * rneg)
    from paddlerec_amd.deepfm import auc_from_buckets
    assert auc_from_buckets(N_(pos), N_(neg)) == R.auc_from_buckets(2 * rpos, 2 * rneg)


# ------------------------------------------------------------------------------ gather / sum-pool
@pytest.mark.parametrize("D", [9, 16, 64])
def test_gather_and_sumpool(ops, D):
    rng = np.random.default_rng(D)
    N, B = 1000, 300
    W = rng.standard_normal((N, D)).astype(np.float32)
    lens = rng.integers(0, 20, B)
    lens[0] = 0; lens[1] = 150                                          # empty and long segments
    lod = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    ids = rng.integers(0, N, lod[-1], dtype=np.int64)                   # id 0 = padding, occurs
    out, counts, status = ops.emb_gather_sumpool(T(ids), T(lod), T(W), 0)
    rout, rcnt = R.sequence_pool_sum(W, ids, lod)
    assert int(status.item()) == 0
    assert np.array_equal(N_(counts), rcnt)                             # pooling counts bit-exact
    assert np.array_equal(N_(out), rout)                                # same ascending-k order
    g, _ = ops.emb_gather(T(ids), T(W), 0)
    assert np.array_equal(N_(g), R.embedding_lookup(W, ids, 0))
    d_out = rng.standard_normal((B, D)).astype(np.float32)
    rg = ops.emb_sumpool_bwd(T(lod), T(d_out), int(lod[-1]))
    assert np.array_equal(N_(rg), np.repeat(d_out, lens, axis=0))


# ------------------------------------------------------------------------------ end to end
@pytest.mark.parametrize("name", ["deepfm_D9", "deepfm_D16"])
def test_deepfm_layer_forward_and_grads_golden(ops, name):
    """Host mirror + kernels reproduce the reference net.py outputs and autograd gradients."""
    from paddlerec_amd.deepfm import DeepFMLayer
    g = load_golden(name)
    D = int(g["D"])
    m = DeepFMLayer(1001, D, 13, 26, [32, 16], device=DEV)
    sd = {"fm.embedding.weight": g["W"], "fm.embedding_one.weight": g["W1"],
          "fm.dense_w": g["dense_w"], "fm.dense_w_one": g["dense_w_one"]}
    for i in range(int(g["n_mlp"])):
        sd["dnn.linear_%d.weight" % i] = g["mlp_w%d" % i]
        sd["dnn.linear_%d.bias" % i] = g["mlp_b%d" % i]
    m.set_dict(sd)
    sparse_inputs = [T(g["ids"][:, s:s + 1]) for s in range(26)]        # list of [B,1], as the reader feeds
    pred = m.forward(sparse_inputs, T(g["dense"]))
    np.testing.assert_allclose(N_(pred), g["pred"], rtol=RTOL, atol=ATOL)
    W0 = g["W"].copy()
    loss, _ = m.train_step(sparse_inputs, T(g["dense"]), T(g["label"]), lr=1e-3)
    np.testing.assert_allclose(N_(loss)[0], g["loss"], rtol=RTOL)
    for i in range(int(g["n_mlp"])):
        assert_close_scaled(N_(m.dense.g["dnn.linear_%d.weight" % i]), g["g_mlp_w%d" % i])
    assert_close_scaled(N_(m.dense.g["fm.dense_w"]).reshape(g["g_dense_w"].shape), g["g_dense_w"])
    assert_close_scaled(N_(m.dense.g["fm.dense_w_one"]), g["g_dense_w_one"])
    # first Adam step moves each touched row by -lr*sign(g) (m/(sqrt(v)+eps) = sign at t=1)
    moved = N_(m.fm.embedding) - W0
    touched = np.abs(g["gW"]) > 1e-4                 # update = lr*g/(|g|+eps): eps=1e-8 matters for tiny g
    np.testing.assert_allclose(moved[touched], -1e-3 * np.sign(g["gW"][touched]), rtol=1e-3)
    assert np.all(moved[g["gW"] == 0] == 0)


def test_train_steps_vs_oracle(ops):
    """3 full steps (fwd + bwd + lazy Adam on tables + dense Adam) against the NumPy oracle."""
    from paddlerec_amd.deepfm import DeepFMLayer
    B, N, D, fc = 256, 400, 16, [32, 16]
    pr = make_deepfm_problem(B=B, N=N, D=D, fc=fc, seed=11, pad_frac=0.05)
    p = pr["params"]
    m = DeepFMLayer(N, D, 13, 26, fc, device=DEV)
    sd = {"fm.embedding.weight": p["W"], "fm.embedding_one.weight": p["W1"],
          "fm.dense_w": p["dense_w"], "fm.dense_w_one": p["dense_w_one"]}
    for i in range(len(fc) + 1):
        sd["dnn.linear_%d.weight" % i] = p["mlp_w"][i]
        sd["dnn.linear_%d.bias" % i] = p["mlp_b"][i]
    m.set_dict(sd)
    # oracle state
    op = {k: (v.copy() if not isinstance(v, list) else [x.copy() for x in v]) for k, v in p.items()}
    st = {k: np.zeros_like(v) for k, v in (("mW", p["W"]), ("vW", p["W"]), ("mW1", p["W1"]), ("vW1", p["W1"]))}
    dstate = {}
    rng = np.random.default_rng(5)
    for step in (1, 2, 3):
        ids = rng.integers(0, N, (B, 26), dtype=np.int64)
        dense = rng.random((B, 13), dtype=np.float32)
        label = (rng.random((B, 1)) < 0.3).astype(np.int64)
        loss, pred = m.train_step(T(ids), T(dense), T(label), lr=1e-2)
        o = R.deepfm_loss_and_grads(ids, dense, label, op)
        np.testing.assert_allclose(N_(loss)[0], o["loss"], rtol=RTOL)
        np.testing.assert_allclose(N_(pred), o["pred"], rtol=RTOL, atol=ATOL)
        uniq, merged, _ = R.merge_rows(o["rows"], o["row_valid"], o["row_grad"])
        R.adam_update_rows(op["W"], st["mW"], st["vW"], uniq, merged, step, lr=1e-2)
        uniq1, merged1, _ = R.merge_rows(o["rows"], o["row_valid"], o["row_grad1"])
        R.adam_update_rows(op["W1"], st["mW1"], st["vW1"], uniq1, merged1, step, lr=1e-2)
        dense_pairs = [("dense_w", o["d_dense_w"]), ("dense_w_one", o["d_dense_w_one"])]
        for i in range(len(fc) + 1):
            dense_pairs += [(("mlp_w", i), o["mlp_dw"][i]), (("mlp_b", i), o["mlp_db"][i])]
        for key, gr in dense_pairs:
            arr = op[key] if not isinstance(key, tuple) else op[key[0]][key[1]]
            mm, vv = dstate.setdefault(key, (np.zeros_like(arr), np.zeros_like(arr)))
            R.adam_update(arr, mm, vv, gr.reshape(arr.shape).astype(arr.dtype), step, lr=1e-2)
    assert int(m.status.item()) == 0
    # The Adam MOMENTS are linear (m) / quadratic (v) in the gradients: they must agree at the 1e-5 bar.
    m._ensure_sparse_state()
    assert_close_scaled(N_(m.sparse_state["m"]), st["mW"])
    assert_close_scaled(N_(m.sparse_state["v"]), st["vW"])
    assert_close_scaled(N_(m.sparse_state["m1"]), st["mW1"])
    assert_close_scaled(N_(m.sparse_state["v1"]), st["vW1"])
    assert_close_scaled(N_(m.dense.pm["dnn.linear_0.weight"]), dstate[("mlp_w", 0)][0])
    assert_close_scaled(N_(m.dense.pv["dnn.linear_0.weight"]), dstate[("mlp_w", 0)][1])
    # The WEIGHTS go through m / (sqrt(v) + eps): where a gradient is ~0 the ratio amplifies fp32 noise up to a sign
    # flip, i.e. up to lr per step for a handful of elements — so: all but a vanishing fraction agree at 1e-6
    # absolute, and nothing moved further than the 3 steps allow.
    for got, want in ((N_(m.fm.embedding), op["W"]), (N_(m.fm.embedding_one), op["W1"]),
                      (N_(m.dense.p["dnn.linear_0.weight"]), op["mlp_w"][0])):
        d = np.abs(got - want)
        assert d.max() <= 3 * 1e-2 and (d > 2e-6).mean() < 2e-3, (d.max(), (d > 2e-6).mean())


def test_bench_batch_loss_vs_c_oracle(ops, oracle_lib):
    """The bench's batch size against the ORACLE (not torch-on-the-same-GPU): B = 65536, the bench's MLP, 26 slot
    tables (50 k rows each keeps the host side small; the reduction over 65536 samples is what is being checked):
    loss and predictions of one train step == C oracle (embedding + FM) + NumPy MLP."""
    from helpers import c_fm_fwd
    from paddlerec_amd.deepfm import DeepFMLayer
    B, S, D, NT, fc = 65536, 26, 16, 50_000, [400, 400, 400]
    pr = make_deepfm_problem(B=B, N=NT, D=D, fc=fc, seed=20250404, pad_frac=0.03, tables=True)
    p = pr["params"]
    so = pr["slot_offsets"]
    m = DeepFMLayer(NT * S, D, 13, S, fc, device=DEV, slot_offset=T(so))
    sd = {"fm.embedding.weight": p["W"], "fm.embedding_one.weight": p["W1"],
          "fm.dense_w": p["dense_w"], "fm.dense_w_one": p["dense_w_one"]}
    for i in range(len(fc) + 1):
        sd["dnn.linear_%d.weight" % i] = p["mlp_w"][i]
        sd["dnn.linear_%d.bias" % i] = p["mlp_b"][i]
    m.set_dict(sd)
    loss, pred = m.train_step(T(pr["ids"]), T(pr["dense"]), T(pr["label"]), lr=1e-3)
    y1, y2, feat, _ = c_fm_fwd(oracle_lib, pr["ids"], pr["dense"], p["W"], p["W1"], p["dense_w"], p["dense_w_one"], 0, so)
    y_dnn = R.dnn_forward(feat, p["mlp_w"], p["mlp_b"])
    want_pred = R.sigmoid(y1[:, None] + y2[:, None] + y_dnn).astype(np.float32)
    want_loss = R.log_loss_mean(want_pred, pr["label"])
    np.testing.assert_allclose(N_(loss)[0], want_loss, rtol=RTOL)
    np.testing.assert_allclose(N_(pred), want_pred, rtol=RTOL, atol=ATOL)
    assert int(m.status.item()) == 0


# ------------------------------------------------------------------------------ full size (BASELINE config 2)
def test_full_size_properties(ops):
    """B=65536, 26 tables x 1M rows x D=16: size-independent properties instead of an oracle run."""
    from paddlerec_amd.deepfm import DeepFMLayer
    B, S, D, NT = 65536, 26, 16, 1_000_000
    N = NT * S
    gen = torch.Generator(device=DEV).manual_seed(20250404)
    W = (torch.randn(N, D, device=DEV, generator=gen) * 0.025)
    W1 = (torch.randn(N, 1, device=DEV, generator=gen) * 0.025)
    dw = torch.randn(1, 13, D, device=DEV, generator=gen) * 0.025
    dw1 = torch.randn(13, device=DEV, generator=gen) * 0.025
    ids = torch.randint(1, NT, (B, S), device=DEV, generator=gen)
    ids[torch.rand(B, S, device=DEV, generator=gen) < 0.03] = 0
    dense = torch.rand(B, 13, device=DEV, generator=gen)
    so = torch.arange(S, device=DEV, dtype=torch.int64) * NT
    y1, y2, feat, sum_emb, status = ops.deepfm_fm_fwd(ids, dense, W, W1, dw, dw1, 0, so)
    assert int(status.item()) == 0
    # (1) feat is exactly the gathered rows / dense products
    rows = ids + so[None, :]
    want = W[rows] * (ids != 0).unsqueeze(-1)
    assert torch.equal(feat[:, :S], want)
    assert torch.equal(feat[:, S:], dense.unsqueeze(2) * dw)
    # (2) FM identity in float64 from the kernel's own feat
    f64 = feat.double()
    y2_64 = 0.5 * (f64.sum(1) ** 2 - (f64 ** 2).sum(1)).sum(1, keepdim=True)
    torch.testing.assert_close(y2.double(), y2_64, rtol=RTOL, atol=1e-6)
    y1_64 = (W1[rows].double() * (ids != 0).unsqueeze(-1)).sum(1) + (dense.double() * dw1.double()).sum(1, keepdim=True)
    torch.testing.assert_close(y1.double(), y1_64, rtol=RTOL, atol=1e-6)
    # (3) grouping invariants: permutation of the non-padding positions, sorted rows, counts add up
    ws = ops.Workspace(DEV)
    groups, _ = ops.ids_group(ids, N, 0, ws, so)
    U, nv = groups.n_uniq.tolist()[:2]
    assert nv == int((ids != 0).sum())
    spos = groups.sorted_pos[:nv].long()
    assert torch.equal(torch.sort(spos).values, torch.nonzero((ids != 0).reshape(-1)).reshape(-1))
    srows = rows.reshape(-1)[spos]
    assert bool((srows[1:] >= srows[:-1]).all())
    uniq = groups.uniq_rows[:U]
    assert torch.equal(uniq, torch.unique(srows))
    offs = groups.seg_offset[:U + 1].long()
    assert int(offs[0]) == 0 and int(offs[-1]) == nv and bool((offs[1:] > offs[:-1]).all())
    assert torch.equal(srows[offs[:-1]], uniq)
    # (4) backward is linear in the incoming gradients; merged row-gradient mass is conserved
    dfeat = torch.randn(B, S + 13, D, device=DEV, generator=gen) * 1e-3
    dz = torch.randn(B, 1, device=DEV, generator=gen) * 1e-3
    rg1, ddw_a, _ = ops.deepfm_fm_bwd(dense, feat, sum_emb, dfeat, dz, dz, S, ws)
    rg1 = rg1.clone(); ddw_a = ddw_a.clone()
    rg2, ddw_b, _ = ops.deepfm_fm_bwd(dense, feat, sum_emb, 2 * dfeat, 2 * dz, 2 * dz, S, ws)
    torch.testing.assert_close(rg2, 2 * rg1, rtol=1e-6, atol=1e-12)
    torch.testing.assert_close(ddw_b, 2 * ddw_a, rtol=1e-5, atol=1e-9)
    want_rg = dfeat[:, :S] + dz.unsqueeze(2) * (sum_emb.unsqueeze(1) - feat[:, :S])
    torch.testing.assert_close(rg1.view(B, S, D), want_rg, rtol=1e-5, atol=1e-9)
    # (5) lazy Adam at t=1 from zero moments moves exactly the touched rows, by lr*sign(merged grad)
    M = torch.zeros_like(W); V = torch.zeros_like(W); P = W.clone()
    ops.sparse_adam_rows(groups, rg1, 1, P, M, V, 1, lr=1e-3)
    delta = P - W
    touched = torch.zeros(N, dtype=torch.bool, device=DEV); touched[uniq] = True
    assert bool((delta[~touched] == 0).all())
    merged = torch.zeros(N, D, device=DEV, dtype=torch.float64)
    merged.index_add_(0, rows.reshape(-1)[(ids != 0).reshape(-1)], rg1.double()[(ids != 0).reshape(-1)])
    torch.testing.assert_close(M[uniq].double(), 0.1 * merged[uniq], rtol=1e-4, atol=1e-10)
    big = merged[uniq].abs() > 1e-6
    torch.testing.assert_close(delta[uniq][big].double(), (-1e-3 * torch.sign(merged[uniq]))[big],
                               rtol=2e-2, atol=0)
    del W, P, M, V, merged
    torch.cuda.empty_cache()
    # (6) the host mirror runs a whole step at this size and the loss is finite and decreases
    m = DeepFMLayer(N, D, 13, S, [400, 400, 400], device=DEV, slot_offset=so)
    label = (torch.rand(B, 1, device=DEV, generator=gen) < 0.25).long()
    losses = [float(m.train_step(ids, dense, label, lr=1e-3)[0].item()) for _ in range(4)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert int(m.status.item()) == 0


@pytest.mark.parametrize("zipf,tables", [(False, False), (False, True), (True, False)])
def test_small_merge_step_equals_sorted_merge_step(ops, monkeypatch, zipf, tables):
    """At the reference's batch size (config_bigdata.yaml: 512 x 26 lookups) the record update merges the duplicate rows
    inside its one launch (rec_sparse_adam_record_small); REC_SMALL_MERGE=0 keeps the grouping sort + partials + record
    update.  Three steps of two identical layers: with uniform ids both sum a row's gradients in ascending position,
    so every parameter and moment is bit-identical; with Zipf ids the sorted path pre-reduces its hot rows per 64-position
    tile (another association): equal at 1e-5 of each tensor's scale."""
    from paddlerec_amd.deepfm import DeepFMLayer
    B, N, D, fc = 512, 3000, 16, [64, 32]
    pr = make_deepfm_problem(B=B, N=N, D=D, fc=fc, seed=3, zipf=zipf, tables=tables)
    so = pr["slot_offsets"]
    a = DeepFMLayer(pr["N"], D, 13, 26, fc, device=DEV, slot_offset=so)
    b = DeepFMLayer(pr["N"], D, 13, 26, fc, device=DEV, slot_offset=so)
    b.fm.rec.copy_(a.fm.rec)
    b.dense.data.copy_(a.dense.data)
    rng = np.random.default_rng(8)
    for step in range(3):
        pb = make_deepfm_problem(B=B, N=N, D=D, fc=fc, seed=100 + step, zipf=zipf, tables=tables)
        ids, dense = T(pb["ids"]), T(pb["dense"])
        label = T((rng.random((B, 1)) < 0.3).astype(np.int64))
        monkeypatch.setenv("REC_SMALL_MERGE", "1")
        la, _ = a.train_step(ids, dense, label, lr=1e-2)
        monkeypatch.setenv("REC_SMALL_MERGE", "0")
        lb, _ = b.train_step(ids, dense, label, lr=1e-2)
        if not zipf:
            assert torch.equal(la, lb)
        else:
            np.testing.assert_allclose(N_(la), N_(lb), rtol=1e-5)
    pairs = [(a.fm.rec, b.fm.rec), (a.sparse_state["mv"], b.sparse_state["mv"]), (a.dense.data, b.dense.data)]
    if not zipf:
        for x, y in pairs:
            assert torch.equal(x, y)
    else:       # (after the first step the tables differ in the last bits, so everything downstream does too)
        assert_close_scaled(N_(a.sparse_state["mv"]), N_(b.sparse_state["mv"]))
        assert_close_scaled(N_(a.dense.m), N_(b.dense.m))
    assert int(a.status.item()) == 0 and int(b.status.item()) == 0


@pytest.mark.gpu
def test_small_merge_with_ids_outside_their_slot_span(ops, monkeypatch):
    """The one-launch merge compares a lookup with the lookups of its own slot only — unless some id leaves its slot's
    span of rows, where rows of different slots can coincide: slot offsets 100 apart, ids up to 300, so most rows are hit
    from three slots.  Must equal the sort-based merge bit for bit (both add a row's gradients in ascending position)."""
    from paddlerec_amd.deepfm import DeepFMLayer
    B, D, S, fc = 320, 16, 26, [32, 16]
    so = torch.arange(S, dtype=torch.int64) * 100
    N = int(so[-1]) + 300
    torch.manual_seed(5)
    a = DeepFMLayer(N, D, 13, S, fc, device=DEV, slot_offset=so)
    b = DeepFMLayer(N, D, 13, S, fc, device=DEV, slot_offset=so)
    b.fm.rec.copy_(a.fm.rec)
    b.dense.data.copy_(a.dense.data)
    g = torch.Generator(device=DEV).manual_seed(9)
    for step in range(3):
        ids = torch.randint(0, 300, (B, S), device=DEV, generator=g)
        dense = torch.rand(B, 13, device=DEV, generator=g)
        label = (torch.rand(B, 1, device=DEV, generator=g) < 0.3).to(torch.int64)
        monkeypatch.setenv("REC_SMALL_MERGE", "1")
        monkeypatch.setenv("REC_STEP_PLAN", "0")
        la, _ = a.train_step(ids, dense, label, lr=1e-2)
        monkeypatch.setenv("REC_SMALL_MERGE", "0")
        lb, _ = b.train_step(ids, dense, label, lr=1e-2)
        assert torch.equal(la, lb)
    assert torch.equal(a.fm.rec, b.fm.rec) and torch.equal(a.sparse_state["mv"], b.sparse_state["mv"])
    assert torch.equal(a.dense.data, b.dense.data)
    assert int(a.status.item()) == 0 and int(b.status.item()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("B", [96, 700])        # the one-launch merge / the grouping sort + record update (18200 lookups)
def test_planned_step_equals_eager_step(engine_lib, monkeypatch, B):
    """Launch-bound batches: the step replayed from its recorded call list (paddlerec_amd/plan.py) leaves the SAME bits
    in every parameter, moment, loss and prediction as the eager step — over several steps with fresh inputs each (a
    torch kernel hidden in the step, a stale pointer or a frozen Adam step count would all show)."""
    from paddlerec_amd.deepfm import DeepFMLayer
    N, D = 5000, 16
    runs = {}
    for mode in ("1", "0", "default"):
        # "1": the recorded call list; "0": eager; "default": launch-bound sizes go through rec_deepfm_train_step (the
        # one-launch tail of csrc/tail_roles.h), the others through the call list — all three leave the same bits
        monkeypatch.setenv("REC_STEP_PLAN", "0" if mode == "0" else "1")
        monkeypatch.setenv("REC_SMALL_C_STEP", "1" if mode == "default" else "0")
        torch.manual_seed(3)
        m = DeepFMLayer(N, D, 13, 26, [64, 32], device=DEV)
        auc = (torch.zeros(4096, dtype=torch.int64, device=DEV), torch.zeros(4096, dtype=torch.int64, device=DEV))
        g = torch.Generator(device=DEV).manual_seed(11)
        outs, kept = [], []
        for step in range(6):
            ids = torch.randint(0, N, (B, 26), device=DEV, generator=g)
            dense = torch.rand(B, 13, device=DEV, generator=g)
            label = (torch.rand(B, 1, device=DEV, generator=g) < 0.3).to(torch.int64)
            loss, pred = m.train_step(ids, dense, label, lr=1e-2 * (1 + step), auc_stats=auc)
            outs.append((loss.cpu().numpy().copy(), pred.cpu().numpy().copy()))
            kept.append((loss, pred))
        # a caller that KEEPS the per-step tensors (host-side metrics after N steps) sees every step's own values: a
        # replayed step writes its loss / predictions into fresh tensors, like the eager step
        for (lk, pk), (lc, pc) in zip(kept, outs):
            assert np.array_equal(lk.cpu().numpy(), lc) and np.array_equal(pk.cpu().numpy(), pc)
        assert len({t[1].data_ptr() for t in kept}) == len(kept)
        runs[mode] = (outs, m.fm.rec.cpu().numpy(), m.sparse_state["mv"].cpu().numpy(), m.dense.data.cpu().numpy(),
                      m.dense.m.cpu().numpy(), m.dense.v.cpu().numpy(), auc[0].cpu().numpy(), auc[1].cpu().numpy(),
                      m.step_count, len(m._plans))
    a, b, c = runs["1"], runs["0"], runs["default"]
    assert a[-1] == 1 and b[-1] == 0 and a[-2] == b[-2] == c[-2] == 6     # the planned run really recorded a plan
    assert c[-1] == (0 if B * 26 <= 15360 else 1)                         # ... and the small batch went through the C step
    for other in (b, c):
        for (la, pa), (lb, pb) in zip(a[0], other[0]):
            assert np.array_equal(la, lb) and np.array_equal(pa, pb)
        for x, y in zip(a[1:8], other[1:8]):
            assert np.array_equal(x, y)


@pytest.mark.gpu
@pytest.mark.parametrize("D,B", [(9, 700), (10, 8192)])
def test_padded_layer0_input_equals_the_dense_layout(engine_lib, monkeypatch, D, B):
    """The reference's own layout (deepfm/config.yaml: 39 fields x D 9, D 10 in config_bigdata) gives layer 0 an input
    width that is no multiple of the GEMM tiles (351 / 390); DeepFMLayer then keeps feat at a padded sample stride (400,
    rec_deepfm_desc.feat_stride) and runs layer 0 on its weight with zero rows behind it.  Same model, same numbers: against
    the dense layout (REC_DEEPFM_PAD0=0) over several steps — predictions and loss at fp32 rounding of the GEMM's K order,
    every parameter inside the Adam bar, the state_dict shapes unchanged."""
    from helpers import assert_adam_weights_close
    from paddlerec_amd.deepfm import DeepFMLayer
    monkeypatch.setenv("REC_STEP_PLAN", "0")
    N, lr, steps = 6000, 1e-2, 3
    runs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("REC_DEEPFM_PAD0", mode)
        torch.manual_seed(4)
        m = DeepFMLayer(N, D, 13, 26, [80, 48], device=DEV)
        assert m.padded == (mode == "1") and m.in0 == 39 * D and (m.ld0 == 400 if m.padded else m.ld0 == m.in0)
        assert tuple(m.state_dict()["dnn.linear_0.weight"].shape) == (39 * D, 80)
        g = torch.Generator(device=DEV).manual_seed(12)
        outs = []
        for step in range(steps):
            ids = torch.randint(0, N, (B, 26), device=DEV, generator=g)
            dense = torch.rand(B, 13, device=DEV, generator=g)
            label = (torch.rand(B, 1, device=DEV, generator=g) < 0.3).to(torch.int64)
            loss, pred = m.train_step(ids, dense, label, lr=lr)
            outs.append((float(loss), pred.cpu().numpy().copy()))
        ev = m(ids, dense).cpu().numpy()                     # the inference path takes the padded layout too
        if m.padded:      # the zero rows behind W_0 live in the flat buffers: parameter, gradient, m and v all stay zero
            o = m.dense.offsets["dnn.linear_0.weight"]
            lo, hi = o + m.in0 * 80, o + m.ld0 * 80
            assert m._w0p.data_ptr() == m.mlp_w[0].data_ptr() and m._dw0p.data_ptr() == m.mlp_dw[0].data_ptr()
            for buf in (m.dense.data, m.dense.grad, m.dense.m, m.dense.v):
                assert float(buf[lo:hi].abs().max()) == 0.0
            assert hi <= m.dense.offsets["dnn.linear_0.bias"]
        runs[mode] = (outs, {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}, ev)
        assert int(m.status.item()) == 0
    (oa, sa, ea), (ob, sb, eb) = runs["1"], runs["0"]
    for (la, pa), (lb, pb) in zip(oa, ob):
        assert abs(la - lb) <= 2e-6 * max(abs(lb), 1e-3)
        np.testing.assert_allclose(pa, pb, rtol=0, atol=5e-6)
    np.testing.assert_allclose(ea, eb, rtol=0, atol=5e-6)
    for k in sa:
        assert_adam_weights_close(sa[k], sb[k], lr, steps, err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("plan", ["0", "1"])
def test_one_launch_dense_fold_equals_the_separate_kernels(engine_lib, monkeypatch, plan):
    """Compact feat (Dn 13 <= D 16): the fold of the dense rows as one launch per direction (rec_dense_fold_*_full, dW_0'
    in scratch) against copy + rec_dense_fold_fwd / copy + rec_dense_fold_bwd on the gradient buffer: the same bits."""
    from paddlerec_amd.deepfm import DeepFMLayer
    monkeypatch.setenv("REC_STEP_PLAN", plan)
    N, D, B = 5000, 16, 600
    runs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("REC_FOLD_FULL", mode)
        torch.manual_seed(5)
        m = DeepFMLayer(N, D, 13, 26, [64, 32], device=DEV)
        assert m.compact and not m.padded
        g = torch.Generator(device=DEV).manual_seed(13)
        outs = []
        for step in range(4):
            ids = torch.randint(0, N, (B, 26), device=DEV, generator=g)
            dense = torch.rand(B, 13, device=DEV, generator=g)
            label = (torch.rand(B, 1, device=DEV, generator=g) < 0.3).to(torch.int64)
            loss, pred = m.train_step(ids, dense, label, lr=1e-2)
            outs.append((loss.cpu().numpy().copy(), pred.cpu().numpy().copy()))
        runs[mode] = (outs, m.fm.rec.cpu().numpy(), m.dense.data.cpu().numpy(), m.dense.grad.cpu().numpy(),
                      m.dense.m.cpu().numpy(), m.dense.v.cpu().numpy())
    a, b = runs["1"], runs["0"]
    for (la, pa), (lb, pb) in zip(a[0], b[0]):
        assert np.array_equal(la, lb) and np.array_equal(pa, pb)
    for x, y in zip(a[1:], b[1:]):
        assert np.array_equal(x, y)


@pytest.mark.gpu
def test_planned_step_on_the_padded_layout_equals_its_eager_step(engine_lib, monkeypatch):
    """The recorded call list on the padded layer-0 layout (D 10: 390 -> 400 columns; persistent zero-padded feat buffer,
    the weight's zero rows inside the flat buffers): bit-identical to the eager step over six steps."""
    from paddlerec_amd.deepfm import DeepFMLayer
    N, D, B = 5000, 10, 512
    runs = {}
    for mode in ("1", "0", "default"):
        # "1": the recorded call list; "0": eager; "default": launch-bound sizes go through rec_deepfm_train_step (the
        # one-launch tail of csrc/tail_roles.h), the others through the call list — all three leave the same bits
        monkeypatch.setenv("REC_STEP_PLAN", "0" if mode == "0" else "1")
        monkeypatch.setenv("REC_SMALL_C_STEP", "1" if mode == "default" else "0")
        torch.manual_seed(3)
        m = DeepFMLayer(N, D, 13, 26, [64, 32], device=DEV)
        assert m.padded and m.ld0 == 400
        g = torch.Generator(device=DEV).manual_seed(11)
        outs = []
        for step in range(6):
            ids = torch.randint(0, N, (B, 26), device=DEV, generator=g)
            dense = torch.rand(B, 13, device=DEV, generator=g)
            label = (torch.rand(B, 1, device=DEV, generator=g) < 0.3).to(torch.int64)
            loss, pred = m.train_step(ids, dense, label, lr=1e-2 * (1 + step))
            outs.append((loss.cpu().numpy().copy(), pred.cpu().numpy().copy()))
        runs[mode] = (outs, m.fm.rec.cpu().numpy(), m.sparse_state["mv"].cpu().numpy(), m.dense.data.cpu().numpy(),
                      m.dense.m.cpu().numpy(), len(m._plans))
    a, b = runs["1"], runs["0"]
    assert a[-1] == 1 and b[-1] == 0
    for (la, pa), (lb, pb) in zip(a[0], b[0]):
        assert np.array_equal(la, lb) and np.array_equal(pa, pb)
    for x, y in zip(a[1:5], b[1:5]):
        assert np.array_equal(x, y)
