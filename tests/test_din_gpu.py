"""DIN on the HIP kernels: fused attention-pool and the full DINLayer forward against the golden fixture of
the reference's unmodified din/net.py and the NumPy oracle (oracle/din_ref.py)."""
import numpy as np
import pytest
import torch

from helpers import assert_close_floor, assert_close_scaled, load_golden
from oracle import din_ref as Dn

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV)


def N_(t):
    return t.detach().cpu().numpy()


def test_din_layer_forward_golden(engine_lib):
    from paddlerec_amd.din import DINLayer
    g = load_golden("din")
    p = {k[2:]: v for k, v in g.items() if k.startswith("p.")}
    m = DINLayer(8, 8, "sigmoid", False, True, 301, 41, device=DEV)
    m.set_dict(p)
    m.set_attention([g["att_w%d" % i] for i in range(3)], [g["att_b%d" % i] for i in range(3)])
    B, Tn = g["hist_item"].shape
    tis = np.repeat(g["target_item"][:, None], Tn, 1)
    tcs = np.repeat(g["target_cat"][:, None], Tn, 1)
    logit = m.forward(T(g["hist_item"]), T(g["hist_cat"]), T(g["target_item"]), T(g["target_cat"]),
                      T(g["label"]), T(g["mask"]), T(tis), T(tcs))
    np.testing.assert_allclose(N_(logit), g["logit"], rtol=1e-5, atol=1e-6)
    assert int(m.status.item()) == 0


@pytest.mark.parametrize("B,Tn,Ei,Ec", [(1, 1, 8, 8), (32, 152, 64, 64), (7, 33, 64, 64), (5, 64, 32, 96),
                                        (300, 40, 64, 64)])
def test_attention_pool_vs_oracle(engine_lib, B, Tn, Ei, Ec):
    from paddlerec_amd import ops
    rng = np.random.default_rng(B + Tn)
    ni, nc, E = 500, 41, Ei + Ec
    tabs = [rng.uniform(-0.3, 0.3, (n, d)).astype(np.float32) for n, d in ((ni, Ei), (nc, Ec), (ni, Ei), (nc, Ec))]
    lens = rng.integers(1, Tn + 1, B)
    lens[0] = Tn
    hi = np.zeros((B, Tn), np.int64); hc = np.zeros((B, Tn), np.int64)
    for b in range(B):
        hi[b, :lens[b]] = rng.integers(1, ni, lens[b]); hc[b, :lens[b]] = rng.integers(1, nc, lens[b])
    mask = np.where(np.arange(Tn)[None] < lens[:, None], 0, -1000000000).astype(np.int64)
    ti = np.repeat(rng.integers(1, ni, B)[:, None], Tn, 1); tc = np.repeat(rng.integers(1, nc, B)[:, None], Tn, 1)
    aw = [rng.uniform(-0.2, 0.2, s).astype(np.float32) for s in ((4 * E, 80), (80, 40), (40, 1))]
    ab = [rng.uniform(-0.1, 0.1, s).astype(np.float32) for s in ((80,), (40,), (1,))]
    out, attw, status = ops.din_attention_pool(T(hi), T(hc), T(ti), T(tc), T(mask), *[T(t) for t in tabs],
                                               [T(w) for w in aw], [T(b) for b in ab])
    assert int(status.item()) == 0
    h = np.concatenate([tabs[0][hi], tabs[1][hc]], 2)
    q = np.concatenate([tabs[2][ti], tabs[3][tc]], 2)
    want, wts = Dn.attention_pool(h, q, mask.astype(np.float32), aw, ab, return_weights=True)
    np.testing.assert_allclose(N_(out), want, rtol=1e-5, atol=2e-6)
    assert_close_scaled(N_(attw), wts, 1e-5)
    # padded positions carry exactly zero weight; weights of a sample sum to 1
    assert np.all(N_(attw)[mask != 0] == 0.0)
    np.testing.assert_allclose(N_(attw).sum(1), 1.0, rtol=1e-5)


def test_attention_pool_skips_only_tiles_that_contribute_nothing(engine_lib):
    """Round 5: the E 128 kernel pair does not compute a 32-position tile that is all padding behind a finite running
    maximum (forward), nor one whose 32 saved weights are all zero (backward).  Masks that are NOT a valid prefix followed
    by padding must still match the oracle: padding in front, a fully padded tile between valid stretches, an all-padding
    history (uniform weights: nothing may be skipped), a full history, a single valid position at the very end."""
    from paddlerec_amd import ops
    rng = np.random.default_rng(515)
    B, Tn, Ei, Ec = 6, 160, 64, 64
    E, ni, nc = Ei + Ec, 300, 41
    valid = np.zeros((B, Tn), bool)
    valid[0, :40] = True
    valid[1, 64:100] = True
    valid[2, :10] = True; valid[2, 96:130] = True
    valid[4, :] = True
    valid[5, 159] = True
    mask = np.where(valid, 0, -1000000000).astype(np.int64)
    tabs = [rng.uniform(-0.3, 0.3, (n, d)).astype(np.float32) for n, d in ((ni, Ei), (nc, Ec), (ni, Ei), (nc, Ec))]
    hi = rng.integers(0, ni, (B, Tn)); hc = rng.integers(0, nc, (B, Tn))
    tis = np.repeat(rng.integers(0, ni, B)[:, None], Tn, 1); tcs = np.repeat(rng.integers(0, nc, B)[:, None], Tn, 1)
    aw = [rng.uniform(-0.3, 0.3, s).astype(np.float32) for s in ((4 * E, 80), (80, 40), (40, 1))]
    ab = [rng.uniform(-0.1, 0.1, s).astype(np.float32) for s in ((80,), (40,), (1,))]
    dout = rng.standard_normal((B, E)).astype(np.float32)
    taw, tab, tt = [T(w) for w in aw], [T(b) for b in ab], [T(t) for t in tabs]
    saved = {}
    out, attw, status = ops.din_attention_pool(T(hi), T(hc), T(tis), T(tcs), T(mask), *tt, taw, tab, saved=saved)
    assert int(status.item()) == 0 and saved["act1"] is not None
    h = np.concatenate([tabs[0][hi], tabs[1][hc]], 2)
    q = np.concatenate([tabs[2][tis], tabs[3][tcs]], 2)
    want, wts = Dn.attention_pool(h, q, mask.astype(np.float32), aw, ab, return_weights=True)
    np.testing.assert_allclose(N_(out), want, rtol=1e-5, atol=2e-6)
    assert_close_scaled(N_(attw), wts, 1e-5)
    got_w = N_(attw)
    assert np.all(got_w[(mask != 0) & valid.any(1, keepdims=True)] == 0.0)          # padded, beside a valid position
    np.testing.assert_allclose(got_w[3], 1.0 / Tn, rtol=1e-5)                       # all padding: uniform, computed
    np.testing.assert_allclose(got_w.sum(1), 1.0, rtol=1e-5)
    dh, dq = ops.din_attention_pool_bwd(T(hi), T(hc), T(tis), T(tcs), *tt, taw, tab, attw, T(dout), saved=saved)
    ref = Dn.attention_pool_backward(h.astype(np.float64), q.astype(np.float64), mask.astype(np.float64),
                                     [w.astype(np.float64) for w in aw], [b.astype(np.float64) for b in ab],
                                     dout.astype(np.float64))
    ref32 = Dn.attention_pool_backward(h, q, mask.astype(np.float32), aw, ab, dout)
    scale = max(np.abs(ref["dh"]).max(), np.abs(ref["dq"]).max())
    assert_close_floor(N_(dh), ref["dh"], ref32["dh"], err_msg="dh", scale=scale)
    assert_close_floor(N_(dq), ref["dq"], ref32["dq"], err_msg="dq", scale=scale)
    pad_beside_valid = (mask != 0) & valid.any(1, keepdims=True)
    assert np.all(N_(dh)[pad_beside_valid] == 0.0) and np.all(N_(dq)[pad_beside_valid] == 0.0)
    # the same through recomputed activations (no saved state): the generic backward, no tile skipping
    dh2, dq2 = ops.din_attention_pool_bwd(T(hi), T(hc), T(tis), T(tcs), *tt, taw, tab, attw, T(dout), saved=None)
    assert_close_floor(N_(dh2), ref["dh"], ref32["dh"], err_msg="dh (recomputed)", scale=scale)


@pytest.mark.parametrize("Ei,Ec", [(8, 8), (64, 64)])      # runtime-shaped kernel / the compile-time-shaped one (E 128)
def test_attention_pool_known_answers(engine_lib, Ei, Ec):
    from paddlerec_amd import ops
    rng = np.random.default_rng(1)
    B, Tn = 4, 70
    E = Ei + Ec
    tabs = [rng.standard_normal((50, d)).astype(np.float32) for d in (Ei, Ec, Ei, Ec)]
    hi = rng.integers(0, 50, (B, Tn)); hc = rng.integers(0, 50, (B, Tn))
    lens = np.array([70, 33, 1, 64])
    mask = np.where(np.arange(Tn)[None] < lens[:, None], 0, -1000000000).astype(np.int64)
    mask[3, :] = -1000000000                      # all-padding history: uniform weights, as the reference
    zw = [np.zeros((4 * E, 80), np.float32), np.zeros((80, 40), np.float32), np.zeros((40, 1), np.float32)]
    zb = [np.zeros(80, np.float32), np.zeros(40, np.float32), np.zeros(1, np.float32)]
    out, attw, _ = ops.din_attention_pool(T(hi), T(hc), T(hi), T(hc), T(mask), *[T(t) for t in tabs],
                                          [T(w) for w in zw], [T(b) for b in zb])
    h = np.concatenate([tabs[0][hi], tabs[1][hc]], 2)
    for b, n in enumerate([70, 33, 1, 70]):       # equal logits => mean-pool over the valid positions
        np.testing.assert_allclose(N_(out)[b], h[b, :n].mean(0), rtol=1e-5, atol=1e-6)
    # out-of-range id: flagged, row read as zero
    hi2 = hi.copy(); hi2[0, 0] = 50
    out2, _, status = ops.din_attention_pool(T(hi2), T(hc), T(hi), T(hc), T(mask), *[T(t) for t in tabs],
                                             [T(w) for w in zw], [T(b) for b in zb])
    assert int(status.item()) & 1
    h2 = h.copy()
    h2[0, 0, :Ei] = 0
    np.testing.assert_allclose(N_(out2)[0], h2[0].mean(0), rtol=1e-5, atol=1e-6)
    # negative id in the last tile of a sample, on the category table
    hc2 = hc.copy(); hc2[1, 32] = -7
    out3, _, status = ops.din_attention_pool(T(hi), T(hc2), T(hi), T(hc), T(mask), *[T(t) for t in tabs],
                                             [T(w) for w in zw], [T(b) for b in zb])
    assert int(status.item()) & 1
    h3 = h.copy()
    h3[1, 32, Ei:] = 0
    np.testing.assert_allclose(N_(out3)[1], h3[1, :33].mean(0), rtol=1e-5, atol=1e-6)


def _din_problem(rng, B, Tn, ni, nc):
    lens = rng.integers(1, Tn + 1, B)
    lens[0] = Tn
    hi = np.zeros((B, Tn), np.int64); hc = np.zeros((B, Tn), np.int64)
    for b in range(B):
        hi[b, :lens[b]] = rng.integers(1, ni, lens[b]); hc[b, :lens[b]] = rng.integers(1, nc, lens[b])
    mask = np.where(np.arange(Tn)[None] < lens[:, None], 0, -1000000000).astype(np.int64)
    ti = rng.integers(1, ni, B).astype(np.int64); tc = rng.integers(1, nc, B).astype(np.int64)
    label = (rng.random((B, 1)) < 0.5).astype(np.float32)
    return hi, hc, ti, tc, mask, label


@pytest.mark.parametrize("use_saved", [False, True])
@pytest.mark.parametrize("B,Tn,Ei,Ec", [(5, 7, 8, 8), (40, 70, 64, 64), (3, 33, 32, 96), (300, 64, 64, 64),
                                        (2, 1, 64, 64)])
def test_attention_pool_bwd_vs_oracle(engine_lib, B, Tn, Ei, Ec, use_saved):
    """use_saved: the backward runs on the forward's saved output and layer-1 activations (the compile-time-shaped
    kernel pair for E 128 / 80-40-1; other shapes save nothing and recompute), else it recomputes everything."""
    from paddlerec_amd import ops
    rng = np.random.default_rng(B * Tn)
    ni, nc, E = 200, 41, Ei + Ec
    tabs = [rng.uniform(-0.3, 0.3, (n, d)).astype(np.float32) for n, d in ((ni, Ei), (nc, Ec), (ni, Ei), (nc, Ec))]
    hi, hc, ti, tc, mask, _ = _din_problem(rng, B, Tn, ni, nc)
    tis, tcs = np.repeat(ti[:, None], Tn, 1), np.repeat(tc[:, None], Tn, 1)
    aw = [rng.uniform(-0.3, 0.3, s).astype(np.float32) for s in ((4 * E, 80), (80, 40), (40, 1))]
    ab = [rng.uniform(-0.1, 0.1, s).astype(np.float32) for s in ((80,), (40,), (1,))]
    dout = rng.standard_normal((B, E)).astype(np.float32)
    taw, tab, tt = [T(w) for w in aw], [T(b) for b in ab], [T(t) for t in tabs]
    saved = {} if use_saved else None
    out, attw, _ = ops.din_attention_pool(T(hi), T(hc), T(tis), T(tcs), T(mask), *tt, taw, tab, saved=saved)
    if use_saved:
        assert (saved["act1"] is not None) == (E == 128)
    dh, dq = ops.din_attention_pool_bwd(T(hi), T(hc), T(tis), T(tcs), *tt, taw, tab, attw, T(dout), saved=saved)
    h = np.concatenate([tabs[0][hi], tabs[1][hc]], 2)
    q = np.concatenate([tabs[2][tis], tabs[3][tcs]], 2)
    ref = Dn.attention_pool_backward(h.astype(np.float64), q.astype(np.float64), mask.astype(np.float64),
                                     [w.astype(np.float64) for w in aw], [b.astype(np.float64) for b in ab],
                                     dout.astype(np.float64))
    # against the float64 oracle, with the fp32 noise floor measured by the same oracle run in float32
    ref32 = Dn.attention_pool_backward(h, q, mask.astype(np.float32), aw, ab, dout)
    scale = max(np.abs(ref["dh"]).max(), np.abs(ref["dq"]).max())   # dq = -sum of the dh-like terms (exactly 0 at T = 1)
    assert_close_floor(N_(dh), ref["dh"], ref32["dh"], err_msg="dh", scale=scale)
    assert_close_floor(N_(dq), ref["dq"], ref32["dq"], err_msg="dq", scale=scale)
    assert np.all(N_(dh)[mask != 0] == 0.0) and np.all(N_(dq)[mask != 0] == 0.0)   # padding: exactly zero


@pytest.mark.parametrize("c_step", [False, True])
def test_din_train_step_golden_grads_and_sgd(engine_lib, monkeypatch, c_step):
    """Gradients of every registered parameter vs the reference's autograd (golden), then the SGD step — through the
    eager mirror (which keeps its gradients for inspection) and through the default of a launch-bound batch, the one-call
    C step (rec_din_train_step: loss and the updated parameters)."""
    from paddlerec_amd.din import DINLayer
    monkeypatch.setenv("REC_SMALL_C_STEP", "1" if c_step else "0")
    g = load_golden("din")
    p = {k[2:]: v for k, v in g.items() if k.startswith("p.")}
    m = DINLayer(8, 8, "sigmoid", False, True, 301, 41, device=DEV)
    m.set_dict(p)
    m.set_attention([g["att_w%d" % i] for i in range(3)], [g["att_b%d" % i] for i in range(3)])
    B, Tn = g["hist_item"].shape
    tis = np.repeat(g["target_item"][:, None], Tn, 1); tcs = np.repeat(g["target_cat"][:, None], Tn, 1)
    lr = 0.85
    loss, pred = m.train_step(T(g["hist_item"]), T(g["hist_cat"]), T(g["target_item"]), T(g["target_cat"]),
                              T(g["label"]), T(g["mask"]), T(tis), T(tcs), base_lr=lr)
    np.testing.assert_allclose(N_(loss)[0], g["loss"], rtol=1e-5)
    for name in ("linear_0", "linear_1", "linear_2", "linearCon"):
        for part in ("weight", "bias"):
            k = "%s.%s" % (name, part)
            if not c_step:
                assert_close_scaled(N_(m._last["dense"][k]), g["g." + k], 1e-5, err_msg=k)
    # after one SGD step every registered parameter equals p - lr * golden gradient
    sd = m.state_dict()
    for k, v in g.items():
        if k.startswith("g."):
            # the UPDATE, not the weight (whose scale would hide it): p_new - p_old against -lr * golden gradient, to
            # 1e-5 of the update's scale + the one fp32 rounding of the stored weight
            upd = -lr * v.astype(np.float64)
            np.testing.assert_allclose(N_(sd[k[2:]]).astype(np.float64) - p[k[2:]], upd, rtol=1e-5, err_msg=k,
                                       atol=1e-5 * np.abs(upd).max() + 1.2e-7 * np.abs(p[k[2:]]).max())
    assert int(m.status.item()) == 0


def test_din_train_steps_vs_oracle(engine_lib):
    from paddlerec_amd.din import DINLayer
    rng = np.random.default_rng(9)
    ni, nc, B, Tn = 120, 30, 48, 40
    m = DINLayer(16, 16, "sigmoid", False, True, ni, nc, device=DEV)
    with torch.no_grad():
        m.params["item_b_attr.weight"].copy_(T((rng.standard_normal((ni, 1)) * 0.1).astype(np.float32)))
    p = {k: N_(v).copy() for k, v in m.state_dict().items()}
    p0 = {k: v.copy() for k, v in p.items()}
    att = ([N_(w).copy() for w in m.attention_w], [N_(b).copy() for b in m.attention_b])
    # the same two steps in float64 (the oracle follows its inputs' dtype): the summation-order-independent truth, and
    # with it THIS test's measured fp32 noise floor = |float32 oracle - float64 oracle| per tensor
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    att64 = ([w.astype(np.float64) for w in att[0]], [b.astype(np.float64) for b in att[1]])
    lr = 0.5
    for step in range(2):
        hi, hc, ti, tc, mask, label = _din_problem(rng, B, Tn, ni, nc)
        tis, tcs = np.repeat(ti[:, None], Tn, 1), np.repeat(tc[:, None], Tn, 1)
        loss, pred = m.train_step(T(hi), T(hc), T(ti), T(tc), T(label), T(mask), T(tis), T(tcs), base_lr=lr)
        grads = Dn.backward(p, att, hi, hc, ti, tc, mask, label)
        np.testing.assert_allclose(N_(loss)[0], Dn.bce_with_logits_mean(grads["_logit"], label), rtol=1e-5)
        for k in p:
            p[k] = (p[k] - lr * np.asarray(grads[k]).reshape(p[k].shape)).astype(np.float32)
        g64 = Dn.backward(p64, att64, hi, hc, ti, tc, mask, label.astype(np.float64))
        for k in p64:
            p64[k] = p64[k] - lr * np.asarray(g64[k]).reshape(p64[k].shape)
    for k, v in m.state_dict().items():      # two SGD steps: the accumulated update against the float64 trajectory's,
        d_got, d_want = N_(v).astype(np.float64) - p0[k], p64[k] - p0[k]       # 1e-5 of its scale + the fp32 rounding of
        floor = float(np.abs((p[k].astype(np.float64) - p0[k]) - d_want).max())  # two stored steps + the float32 oracle's
        np.testing.assert_allclose(d_got, d_want, rtol=1e-5, err_msg=k,           # own distance from the truth
                                   atol=1e-5 * np.abs(d_want).max() + 2.4e-7 * np.abs(p0[k]).max() + 2.0 * floor)


def test_din_train_step_graphed_equals_eager(engine_lib, monkeypatch):
    """train_step_graphed (hipGraph replay per input signature, paddlerec_amd/graph.py) runs the same launches in the
    same order as train_step: after a sequence of batches with TWO padded lengths (two signatures, first sight eager,
    second sight captured, then replays) every parameter and every loss is bit-identical, and no step ran twice."""
    from paddlerec_amd.din import DINLayer
    rng = np.random.default_rng(21)
    ni, nc, B = 150, 30, 32
    a = DINLayer(64, 64, "sigmoid", False, True, ni, nc, device=DEV)
    b = DINLayer(64, 64, "sigmoid", False, True, ni, nc, device=DEV)
    b.set_dict({k: v.clone() for k, v in a.state_dict().items()})
    b.set_attention([w.clone() for w in a.attention_w], [x.clone() for x in a.attention_b])
    batches = []
    for Tn in (40, 24, 40, 40, 24, 40, 24, 24):
        hi, hc, ti, tc, mask, label = _din_problem(rng, B, Tn, ni, nc)
        tis, tcs = np.repeat(ti[:, None], Tn, 1), np.repeat(tc[:, None], Tn, 1)
        batches.append([T(x) for x in (hi, hc, ti, tc, label, mask, tis, tcs)])
    monkeypatch.setenv("REC_STEP_PLAN", "0")        # a: the eager mirror
    for bt in batches:
        la, _ = a.train_step(*bt, base_lr=0.5)
        lb, _ = b.train_step_graphed(*bt, base_lr=0.5)
        assert torch.equal(la, lb)
    g = b._graph
    assert (g.eager, g.captures, g.replays) == (2, 2, 6)          # one eager + one capture per signature, 6 replays
    assert a.step_count == b.step_count == len(batches)
    for k, v in a.state_dict().items():
        assert torch.equal(v, b.state_dict()[k]), k
    assert int(a.status.item()) == 0 and int(b.status.item()) == 0


def test_sparse_sgd_small_multi_equals_single_calls(engine_lib):
    """rec_sparse_sgd_small_multi (up to 8 independent tables in one launch) leaves the same bits as one
    rec_sparse_sgd_small call per table: different sizes, widths 1 / 64 / 200, strided gradient views, a hot row, an
    out-of-range id (flagged, skipped)."""
    from paddlerec_amd import _lib, ops
    rng = np.random.default_rng(77)
    specs = [(4864, 64, 63001, 128), (32, 64, 801, 256), (32, 1, 63001, 1), (15360, 200, 3000, 208), (700, 64, 50, 64)]
    tabs_a, tabs_b, jobs = [], [], []
    for n, D, N, pitch in specs:
        ids = rng.integers(0, N, size=n).astype(np.int64)
        if n > 300:
            ids[rng.integers(0, n, size=n // 4)] = ids[0]
        if n == 700:
            ids[5] = N + 3
        g = rng.standard_normal((n, pitch)).astype(np.float32)
        P0 = rng.standard_normal((N, D)).astype(np.float32)
        tabs_a.append(T(P0.copy()))
        tabs_b.append(T(P0.copy()))
        jobs.append((T(ids), T(g)[:, :D] if pitch > D else T(g), pitch))
    st_a, st_b = ops.new_status(DEV), ops.new_status(DEV)
    ops.sparse_sgd_small_multi([(ids, gv, P, 1, pitch) for (ids, gv, pitch), P in zip(jobs, tabs_a)], 0.37, st_a)
    for (ids, gv, pitch), P in zip(jobs, tabs_b):
        ops.sparse_sgd_small(ids, gv, P, 0.37, None, st_b, grad_group=1, grad_group_stride=pitch)
    for x, y in zip(tabs_a, tabs_b):
        assert torch.equal(x, y)
    assert int(st_a.item()) == int(st_b.item()) and int(st_a.item()) & _lib.REC_FLAG_INDEX_OOB


def test_din_planned_step_equals_eager(engine_lib, monkeypatch):
    """train_step replays a launch-bound step from its recorded C-ABI call list (paddlerec_amd/plan.py) from the third
    sight of an input signature on: with two padded lengths, every loss, prediction and parameter is bit-identical to a
    layer that never plans (REC_STEP_PLAN=0)."""
    from paddlerec_amd.din import DINLayer
    from paddlerec_amd.plan import CallPlan
    rng = np.random.default_rng(22)
    ni, nc, B = 150, 30, 32
    a = DINLayer(64, 64, "sigmoid", False, True, ni, nc, device=DEV)
    b = DINLayer(64, 64, "sigmoid", False, True, ni, nc, device=DEV)
    c = DINLayer(64, 64, "sigmoid", False, True, ni, nc, device=DEV)
    for other in (b, c):
        other.set_dict({k: v.clone() for k, v in a.state_dict().items()})
        other.set_attention([w.clone() for w in a.attention_w], [x.clone() for x in a.attention_b])
    for Tn in (40, 24, 40, 40, 24, 40, 24, 24, 40):
        hi, hc, ti, tc, mask, label = _din_problem(rng, B, Tn, ni, nc)
        tis, tcs = np.repeat(ti[:, None], Tn, 1), np.repeat(tc[:, None], Tn, 1)
        bt = [T(x) for x in (hi, hc, ti, tc, label, mask, tis, tcs)]
        monkeypatch.setenv("REC_STEP_PLAN", "0")
        la, pa = a.train_step(*bt, base_lr=0.5)
        monkeypatch.setenv("REC_STEP_PLAN", "1")
        monkeypatch.setenv("REC_SMALL_C_STEP", "0")          # the recorded call list
        lb, pb = b.train_step(*bt, base_lr=0.5)
        monkeypatch.setenv("REC_SMALL_C_STEP", "1")          # the default: rec_din_train_step (csrc/tail_roles.h)
        lc, pc = c.train_step(*bt, base_lr=0.5)
        assert torch.equal(la, lb) and torch.equal(pa, pb) and torch.equal(la, lc) and torch.equal(pa, pc)
    assert not a._plans and len(b._plans) == 2 and all(isinstance(p, CallPlan) for p in b._plans.values())
    assert not c._plans and c.step_count == a.step_count
    for k, v in a.state_dict().items():
        assert torch.equal(v, b.state_dict()[k]) and torch.equal(v, c.state_dict()[k]), k
    assert int(a.status.item()) == 0 and int(b.status.item()) == 0 and int(c.status.item()) == 0


@pytest.mark.parametrize("n,D,N,hot", [(4864, 128, 63001, True), (32, 64, 801, False), (1, 1, 5, False),
                                        (15360, 200, 3000, True), (777, 1, 100, False), (5000, 256, 50, True)])
def test_sparse_sgd_small_equals_group_then_rows(engine_lib, n, D, N, hot):
    """rec_sparse_sgd_small (merge + SGD in one launch, the bs-32 path) == rec_ids_group + rec_sparse_sgd_rows on the
    same lookups (same ascending-position sums), with a strided gradient view, hot rows, an out-of-range id."""
    from paddlerec_amd import _lib, ops
    rng = np.random.default_rng(n + D)
    ids = rng.integers(0, N, size=n).astype(np.int64)
    if hot and n > 300:
        ids[rng.integers(0, n, size=n // 3)] = ids[0]                   # one row with ~n/3 occurrences
    pitch = D + 8
    gfull = rng.standard_normal((n, pitch)).astype(np.float32)
    P0 = rng.standard_normal((N, D)).astype(np.float32)
    tg = T(gfull)
    Pa, Pb = T(P0), T(P0)
    st = ops.sparse_sgd_small(T(ids), tg[:, 4:4 + D], Pa, 0.37, grad_group=1, grad_group_stride=pitch)
    ws = ops.Workspace(DEV)
    grp, _ = ops.ids_group(T(ids), N, None, ws)
    ops.sparse_sgd_rows(grp, tg[:, 4:4 + D], Pb, 0.37, grad_group=1, grad_group_stride=pitch)
    assert int(st.item()) == 0
    np.testing.assert_allclose(N_(Pa), N_(Pb), rtol=2e-6, atol=1e-6)
    # oracle: float64 merge
    want = P0.astype(np.float64)
    np.subtract.at(want, ids, 0.37 * gfull[:, 4:4 + D].astype(np.float64))
    want32 = P0.copy()
    np.subtract.at(want32, ids, np.float32(0.37) * gfull[:, 4:4 + D])      # the same merge in float32: the noise floor
    assert_close_floor(N_(Pa), want, want32)
    if n >= 32:        # padding id skipped, out-of-range id skipped + flagged
        ids2 = ids.copy(); ids2[5] = N + 3; ids2[7] = 0
        Pc = T(P0)
        st2 = ops.sparse_sgd_small(T(ids2), tg[:, 4:4 + D], Pc, 0.37, padding_idx=0, grad_group=1, grad_group_stride=pitch)
        assert int(st2.item()) & _lib.REC_FLAG_INDEX_OOB
        keep = (ids2 != 0) & (ids2 < N)
        want2 = P0.astype(np.float64)
        np.subtract.at(want2, ids2[keep], 0.37 * gfull[keep, 4:4 + D].astype(np.float64))
        want2_32 = P0.copy()
        np.subtract.at(want2_32, ids2[keep], np.float32(0.37) * gfull[keep, 4:4 + D])
        assert_close_floor(N_(Pc), want2, want2_32)


@pytest.mark.gpu
@pytest.mark.parametrize("B,T", [(32, 152), (3, 100), (7, 33), (40, 64)])
def test_tile_split_forward_and_backward_equal_the_per_sample_walk(engine_lib, monkeypatch, B, T):
    """Few samples (din/config.yaml batch size 32: fewer than the chip has block slots): the forward runs one block per
    32-position history tile and a combine launch rescales the per-tile softmax pieces (rec_din_attention_pool_fwd_ws),
    the backward deals (sample, tile) pairs to the blocks.  Against the same kernels walking a sample's tiles in one
    block (REC_DIN_TILE_SPLIT=0): pooled output, attention weights and both gradients at fp32 rounding."""
    from paddlerec_amd import ops
    g = torch.Generator(device=DEV).manual_seed(B * 1000 + T)
    Ei = Ec = 64
    n_item, n_cat = 500, 60
    tabs = [torch.randn(n, d, device=DEV, generator=g) * 0.3 for n, d in ((n_item, Ei), (n_cat, Ec), (n_item, Ei), (n_cat, Ec))]
    hi = torch.randint(0, n_item, (B, T), device=DEV, generator=g)
    hc = torch.randint(0, n_cat, (B, T), device=DEV, generator=g)
    ti = torch.randint(0, n_item, (B, 1), device=DEV, generator=g).expand(B, T).contiguous()
    tc = torch.randint(0, n_cat, (B, 1), device=DEV, generator=g).expand(B, T).contiguous()
    lens = torch.randint(1, T + 1, (B, 1), device=DEV, generator=g)
    mask = torch.where(torch.arange(T, device=DEV)[None, :] < lens, 0, -(2 ** 32) + 1).to(torch.int64)
    E = Ei + Ec
    aw = [torch.randn(4 * E, 80, device=DEV, generator=g) * 0.05, torch.randn(80, 40, device=DEV, generator=g) * 0.1,
          torch.randn(40, 1, device=DEV, generator=g) * 0.1]
    ab = [torch.randn(80, device=DEV, generator=g) * 0.1, torch.randn(40, device=DEV, generator=g) * 0.1,
          torch.randn(1, device=DEV, generator=g) * 0.1]
    dout = torch.randn(B, E, device=DEV, generator=g)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("REC_DIN_TILE_SPLIT", mode)
        ws, saved = ops.Workspace(DEV), {}
        out, attw, st = ops.din_attention_pool(hi, hc, ti, tc, mask, tabs[0], tabs[1], tabs[2], tabs[3], aw, ab,
                                               saved=saved, ws=ws)
        dh, dq = ops.din_attention_pool_bwd(hi, hc, ti, tc, tabs[0], tabs[1], tabs[2], tabs[3], aw, ab, attw, dout,
                                            saved=saved)
        res[mode] = [x.cpu().numpy() for x in (out, attw, dh, dq)]
        assert int(st.item()) == 0
    for a, b, name in zip(res["1"], res["0"], ("out", "att_weight", "d_hist", "d_tgt")):
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-6 * max(float(np.abs(b).max()), 1e-6), err_msg=name)
    np.testing.assert_allclose(res["1"][1].sum(1), 1.0, atol=1e-5)
