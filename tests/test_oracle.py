"""Pins the oracle: NumPy restatement vs the golden fixtures produced by the reference's own
net.py (oracle/make_golden.py), vs torch-CPU autograd, and the C restatement vs NumPy."""
import numpy as np
import pytest
import torch

from helpers import (c_adam_rows, c_fm_bwd, c_fm_fwd, deepfm_params_from_golden, load_golden,
                     make_deepfm_problem)
from oracle import deepfm_ref as R

RTOL = 1e-5  # north star: fp32 logits within 1e-5 relative


def _close(a, b, rtol=RTOL, atol=1e-7):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


@pytest.mark.parametrize("name", ["deepfm_D9", "deepfm_D16"])
def test_numpy_oracle_matches_reference_net(name):
    g = load_golden(name)
    p = deepfm_params_from_golden(g)
    pred, z, (y1, y2, feat, y_dnn) = R.deepfm_forward(g["ids"], g["dense"], p)
    _close(y1, g["y1"])
    _close(y2, g["y2"], atol=1e-6)
    assert np.array_equal(feat, g["feat"])          # pure gather/multiply: bit-exact
    _close(pred, g["pred"])
    _close(R.log_loss_mean(pred, g["label"]), g["loss"])
    # padding rows really are zero rows in the reference output
    assert np.all(g["feat"][:, :26][g["ids"] == 0] == 0)


@pytest.mark.parametrize("name", ["deepfm_D9", "deepfm_D16"])
def test_numpy_oracle_backward_matches_reference_autograd(name):
    g = load_golden(name)
    p = deepfm_params_from_golden(g)
    out = R.deepfm_loss_and_grads(g["ids"], g["dense"], g["label"], p)
    # SelectedRows -> dense gradient, as autograd over the shim produced it
    uniq, merged, counts = R.merge_rows(out["rows"], out["row_valid"], out["row_grad"])
    gW = np.zeros_like(g["gW"])
    gW[uniq] = merged
    _close(gW, g["gW"], atol=1e-8)
    uniq1, merged1, _ = R.merge_rows(out["rows"], out["row_valid"], out["row_grad1"])
    gW1 = np.zeros_like(g["gW1"])
    gW1[uniq1] = merged1
    _close(gW1, g["gW1"], atol=1e-8)
    assert g["gW"][0].sum() == 0 and g["gW1"][0].sum() == 0     # padding row gets no gradient
    _close(out["d_dense_w"], g["g_dense_w"], atol=1e-8)
    _close(out["d_dense_w_one"], g["g_dense_w_one"], atol=1e-8)
    for i in range(int(g["n_mlp"])):
        _close(out["mlp_dw"][i], g["g_mlp_w%d" % i], atol=1e-8)
        _close(out["mlp_db"][i], g["g_mlp_b%d" % i], atol=1e-8)
    assert counts.sum() == (g["ids"] != 0).sum()


def test_numpy_oracle_vs_torch_autograd_fp64():
    pr = make_deepfm_problem(B=33, N=300, D=10, fc=(16, 8), seed=3, pad_frac=0.1, dtype=np.float64)
    p = pr["params"]
    out = R.deepfm_loss_and_grads(pr["ids"], pr["dense"].astype(np.float64), pr["label"], p)
    tp = {k: torch.tensor(v, requires_grad=True) for k, v in p.items() if not isinstance(v, list)}
    tw = [torch.tensor(w, requires_grad=True) for w in p["mlp_w"]]
    tb = [torch.tensor(b, requires_grad=True) for b in p["mlp_b"]]
    ids = torch.tensor(pr["ids"])
    dense = torch.tensor(pr["dense"].astype(np.float64))
    mask = (ids != 0).unsqueeze(-1).double()
    e1 = tp["W1"][ids] * mask
    y1 = e1.sum(1) + (dense * tp["dense_w_one"]).unsqueeze(2).sum(1)
    feat = torch.cat([tp["W"][ids] * mask, dense.unsqueeze(2) * tp["dense_w"]], 1)
    y2 = 0.5 * (feat.sum(1) ** 2 - (feat ** 2).sum(1)).sum(1, keepdim=True)
    x = feat.reshape(feat.shape[0], -1)
    for i, (w, b) in enumerate(zip(tw, tb)):
        x = x @ w + b
        if i < len(tw) - 1:
            x = torch.relu(x)
    pred = torch.sigmoid(y1 + y2 + x)
    t = torch.tensor(pr["label"]).double()
    loss = (-t * torch.log(pred + 1e-4) - (1 - t) * torch.log(1 - pred + 1e-4)).mean()
    loss.backward()
    np.testing.assert_allclose(out["loss"], loss.item(), rtol=1e-12)
    uniq, merged, _ = R.merge_rows(out["rows"], out["row_valid"], out["row_grad"])
    gW = np.zeros_like(p["W"])
    gW[uniq] = merged
    np.testing.assert_allclose(gW, tp["W"].grad.numpy(), rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(out["d_dense_w"], tp["dense_w"].grad.numpy(), rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(out["d_dense_w_one"], tp["dense_w_one"].grad.numpy(), rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(out["mlp_dw"][0], tw[0].grad.numpy(), rtol=1e-9, atol=1e-14)


@pytest.mark.parametrize("D,tables", [(16, False), (9, False), (16, True)])
def test_c_oracle_matches_numpy(oracle_lib, D, tables):
    pr = make_deepfm_problem(B=257, N=2000, D=D, seed=5, tables=tables)
    p = pr["params"]
    y1, y2, feat, sum_emb = c_fm_fwd(oracle_lib, pr["ids"], pr["dense"], p["W"], p["W1"],
                                     p["dense_w"], p["dense_w_one"], 0, pr["slot_offsets"])
    ry1, ry2, rfeat = R.fm_forward(pr["ids"], pr["dense"], p["W1"], p["W"], p["dense_w_one"],
                                   p["dense_w"], 0, pr["slot_offsets"])
    assert np.array_equal(feat, rfeat)
    _close(y1, ry1[:, 0], atol=1e-7)
    _close(y2, ry2[:, 0], rtol=1e-4, atol=1e-7)
    _close(sum_emb, rfeat.sum(1), atol=1e-7)
    rng = np.random.default_rng(1)
    dfeat = rng.standard_normal(feat.shape).astype(np.float32) * 1e-3
    dz = rng.standard_normal((257, 1)).astype(np.float32) * 1e-3
    rg, rg1, ddw, ddw1 = c_fm_bwd(oracle_lib, 26, pr["dense"], feat, sum_emb, dfeat, dz, dz)
    ref = R.fm_backward(pr["ids"], pr["dense"], rfeat, dfeat, dz, dz, 0, pr["slot_offsets"])
    _close(rg, ref["row_grad"], atol=1e-9)
    _close(rg1, ref["row_grad1"][:, 0], atol=0)
    _close(ddw, ref["d_dense_w"][0], rtol=1e-4, atol=1e-8)
    _close(ddw1, ref["d_dense_w_one"], rtol=1e-4, atol=1e-8)
    # merge + lazy Adam
    spos, uniq, offs = R.group_ids(ref["rows"], ref["row_valid"])
    P, M, V = p["W"].copy(), np.zeros_like(p["W"]), np.zeros_like(p["W"])
    P2, M2, V2 = P.copy(), M.copy(), V.copy()
    for step in (1, 2):
        c_adam_rows(oracle_lib, uniq, offs, spos, rg, P, M, V, step)
        u2, merged, _ = R.merge_rows(ref["rows"], ref["row_valid"], ref["row_grad"])
        assert np.array_equal(u2, uniq)
        R.adam_update_rows(P2, M2, V2, uniq, merged, step)
    _close(P, P2, rtol=1e-6, atol=1e-9)
    _close(M, M2, rtol=1e-6, atol=1e-12)
    _close(V, V2, rtol=1e-6, atol=1e-15)


def test_group_ids_properties():
    pr = make_deepfm_problem(B=64, N=50, seed=9, pad_frac=0.2)
    rows, valid = R.effective_rows(pr["ids"])
    spos, uniq, offs = R.group_ids(rows.reshape(-1), valid.reshape(-1))
    assert len(spos) == valid.sum() and offs[-1] == len(spos)
    assert np.all(np.diff(uniq) > 0)
    flat = rows.reshape(-1)
    for u in range(len(uniq)):
        seg = spos[offs[u]:offs[u + 1]]
        assert np.all(flat[seg] == uniq[u]) and np.all(np.diff(seg) > 0)   # stable


def test_adam_lazy_vs_dense_equivalent_semantics():
    """lazy touches only merged rows; the dygraph default moves every row (Appendix B-3)."""
    rng = np.random.default_rng(0)
    P = rng.standard_normal((10, 4)).astype(np.float32)
    M = np.abs(rng.standard_normal((10, 4))).astype(np.float32)
    V = np.abs(rng.standard_normal((10, 4))).astype(np.float32)
    g = rng.standard_normal((2, 4)).astype(np.float32)
    rows = np.array([2, 7])
    Pl, Ml, Vl = P.copy(), M.copy(), V.copy()
    R.adam_update_rows(Pl, Ml, Vl, rows, g, 3)
    Pd, Md, Vd = P.copy(), M.copy(), V.copy()
    R.adam_update_dense_equivalent(Pd, Md, Vd, rows, g, 3)
    untouched = np.setdiff1d(np.arange(10), rows)
    assert np.array_equal(Pl[untouched], P[untouched])
    assert not np.array_equal(Pd[untouched], P[untouched])
    np.testing.assert_allclose(Pl[rows], Pd[rows], rtol=1e-6)


def test_auc_known_answers():
    sk = pytest.importorskip("sklearn.metrics")
    rng = np.random.default_rng(4)
    # predictions on the bucket grid -> bucketed AUC equals the exact AUC
    p = rng.integers(0, 4096, 5000) / 4095.0
    t = (rng.random(5000) < p * 0.5 + 0.1).astype(np.int64)
    pos, neg = R.auc_histogram(p.astype(np.float32), t)
    assert pos.sum() + neg.sum() == 5000 and pos.dtype == np.int64
    np.testing.assert_allclose(R.auc_from_buckets(pos, neg), sk.roc_auc_score(t, p), rtol=1e-9)
    assert R.auc_from_buckets(np.zeros(4096, np.int64), neg) == 0.5          # utils_single.py:199-201
    pos2 = np.zeros(4, np.int64); neg2 = np.zeros(4, np.int64)
    pos2[3] = 1; neg2[0] = 1
    assert R.auc_from_buckets(pos2, neg2) == 1.0


def test_fm_identity_property():
    """0.5[(sum e)^2 - sum e^2] == sum_{i<j} <e_i, e_j>   (SURVEY §8(c) known-answer property)."""
    pr = make_deepfm_problem(B=7, N=100, D=6, seed=2, dtype=np.float64)
    p = pr["params"]
    _, y2, feat = R.fm_forward(pr["ids"], pr["dense"].astype(np.float64), p["W1"], p["W"],
                               p["dense_w_one"], p["dense_w"])
    F = feat.shape[1]
    want = np.zeros(7)
    for i in range(F):
        for j in range(i + 1, F):
            want += (feat[:, i] * feat[:, j]).sum(1)
    np.testing.assert_allclose(y2[:, 0], want, rtol=1e-10, atol=1e-15)


def test_sequence_pool_and_reader():
    rng = np.random.default_rng(1)
    W = rng.standard_normal((20, 3)).astype(np.float32)
    ids = np.array([3, 0, 5, 5, 7, 0], np.int64)
    lod = np.array([0, 2, 2, 5, 6], np.int64)
    out, cnt = R.sequence_pool_sum(W, ids, lod)
    assert cnt.tolist() == [1, 0, 3, 0]
    np.testing.assert_allclose(out[0], W[3]); assert np.all(out[1] == 0) and np.all(out[3] == 0)
    np.testing.assert_allclose(out[2], W[5] + W[5] + W[7], rtol=1e-6)
    line = "click:1 dense_feature:0.5 " + " ".join("dense_feature:0.0" for _ in range(12)) + " 1:11 3:33 26:99"
    label, sid, dense = R.parse_slot_line(line)
    assert label == 1 and sid[0] == 11 and sid[1] == 0 and sid[2] == 33 and sid[25] == 99
    assert dense.shape == (13,) and dense[0] == np.float32(0.5)


# ------------------------------------------------------------------------------ DCN-v2 oracle vs golden
@pytest.mark.parametrize("name", ["dcn_v2_v2", "dcn_v2_mix"])
def test_dcn_v2_oracle_matches_reference_net(name):
    """oracle/dcn_v2_ref.py reproduces the outputs and autograd gradients of the reference's
    unmodified dcn_v2/net.py (fixture = d pred.sum() / d params)."""
    from oracle import dcn_v2_ref as X
    g = load_golden(name)
    p = {k[2:]: v for k, v in g.items() if k.startswith("p.")}
    pred, saved = X.forward(g["ids"], g["dense"], p, return_saved=True)
    np.testing.assert_allclose(saved["feat"], g["feat"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(saved["cross"], g["cross"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(pred, g["pred"], rtol=1e-5, atol=1e-7)
    grads = X.backward(g["ids"], g["dense"], p, saved, np.ones_like(pred))
    n = 0
    for k, v in g.items():
        if k.startswith("g."):
            got = np.asarray(grads[k[2:]]).reshape(v.shape)
            np.testing.assert_allclose(got, v, rtol=2e-4, atol=2e-6, err_msg=k)
            n += 1
    assert n >= 15
    assert np.all(grads["embedding.weight"][0] == 0)       # padding row gets no gradient


def test_din_oracle_matches_reference_net():
    from oracle import din_ref as Dn
    g = load_golden("din")
    p = {k[2:]: v for k, v in g.items() if k.startswith("p.")}
    att = ([g["att_w%d" % i] for i in range(3)], [g["att_b%d" % i] for i in range(3)])
    logit = Dn.forward(p, att, g["hist_item"], g["hist_cat"], g["target_item"], g["target_cat"],
                       g["mask"][:, :, 0])
    np.testing.assert_allclose(logit, g["logit"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(Dn.bce_with_logits_mean(logit, g["label"]), g["loss"], rtol=1e-6)
    # App. B-9: the attention MLP is not among the registered parameters, the top MLP reuses its names
    assert "linear_0.weight" in p and p["linear_0.weight"].shape == (32, 80) and g["att_w0"].shape == (64, 80)
    # known-answer property: equal logits => mean-pool over the valid positions
    B, T, E = 3, 6, 4
    rng = np.random.default_rng(0)
    h = rng.standard_normal((B, T, E)).astype(np.float32)
    q = rng.standard_normal((B, T, E)).astype(np.float32)
    lens = np.array([6, 2, 1])
    mask = np.where(np.arange(T)[None] < lens[:, None], 0.0, -1e9).astype(np.float32)
    zero_w = [np.zeros((4 * E, 80), np.float32), np.zeros((80, 40), np.float32), np.zeros((40, 1), np.float32)]
    zero_b = [np.zeros(80, np.float32), np.zeros(40, np.float32), np.zeros(1, np.float32)]
    out = Dn.attention_pool(h, q, mask, zero_w, zero_b)
    want = np.stack([h[b, :lens[b]].mean(0) for b in range(B)])
    np.testing.assert_allclose(out, want, rtol=1e-6, atol=1e-7)


def test_din_oracle_backward_matches_reference_autograd():
    from oracle import din_ref as Dn
    g = load_golden("din")
    p = {k[2:]: v for k, v in g.items() if k.startswith("p.")}
    att = ([g["att_w%d" % i] for i in range(3)], [g["att_b%d" % i] for i in range(3)])
    grads = Dn.backward(p, att, g["hist_item"], g["hist_cat"], g["target_item"], g["target_cat"],
                        g["mask"][:, :, 0], g["label"])
    n = 0
    for k, v in g.items():
        if k.startswith("g."):
            np.testing.assert_allclose(np.asarray(grads[k[2:]]).reshape(v.shape), v, rtol=3e-4, atol=3e-7, err_msg=k)
            n += 1
    assert n == 15


def test_adam_restatement_vs_torch_optim_adam():
    """Rows O / PS are pinned to a restatement (SURVEY App. B-3 [EXT]): a SECOND, independent implementation must
    agree.  torch.optim.Adam computes p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps), which is algebraically
    Paddle's lr_t * m / (sqrt(v) + eps*sqrt(1-b2^t)) with lr_t = lr*sqrt(1-b2^t)/(1-b1^t)."""
    import torch
    rng = np.random.default_rng(3)
    p0 = rng.standard_normal((37, 16)).astype(np.float32)
    grads = [(rng.standard_normal(p0.shape) * 10.0 ** rng.integers(-4, 1)).astype(np.float32) for _ in range(6)]
    p, m, v = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
    tp = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.Adam([tp], lr=1e-2, betas=(0.9, 0.999), eps=1e-8)
    for t, g in enumerate(grads, 1):
        R.adam_update(p, m, v, g, t, lr=1e-2)
        tp.grad = torch.from_numpy(g.copy())
        opt.step()
        np.testing.assert_allclose(p, tp.detach().numpy(), rtol=2e-6, atol=2e-7)
    st = opt.state[tp]
    # the restatement forms 1 - beta in float32 like Paddle's kernel (1 - 0.999f = 0.00100005): 4.7e-5 relative on
    # the increment of v against torch's double-precision 0.001 — the moments agree to that, the weights to 2e-6
    np.testing.assert_allclose(m, st["exp_avg"].numpy(), rtol=2e-6, atol=2e-7 * float(np.abs(m).max()))
    np.testing.assert_allclose(v, st["exp_avg_sq"].numpy(), rtol=6e-5, atol=1e-12)
    # lazy rows == torch.optim.SparseAdam semantics on the touched rows (untouched rows keep p, m, v)
    P, M, V = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
    rows = np.array([3, 7, 20])
    gr = rng.standard_normal((3, 16)).astype(np.float32)
    R.adam_update_rows(P, M, V, rows, gr, 1, lr=1e-2)
    sp = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    sopt = torch.optim.SparseAdam([sp], lr=1e-2, betas=(0.9, 0.999), eps=1e-8)
    sp.grad = torch.sparse_coo_tensor(torch.from_numpy(rows).unsqueeze(0), torch.from_numpy(gr), p0.shape)
    sopt.step()
    np.testing.assert_allclose(P, sp.detach().numpy(), rtol=2e-6, atol=2e-7)
    untouched = np.setdiff1d(np.arange(37), rows)
    assert np.array_equal(P[untouched], p0[untouched]) and not M[untouched].any()
