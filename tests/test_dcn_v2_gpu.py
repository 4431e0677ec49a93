"""DCN-v2 on the HIP kernels (paddlerec_amd/dcn_v2.py) against the golden fixtures of the reference's
unmodified dcn_v2/net.py and the NumPy oracle (oracle/dcn_v2_ref.py)."""
import numpy as np
import pytest
import torch

from helpers import assert_close_scaled, assert_moments_close, OracleDCNTrainer, load_golden
from oracle import dcn_v2_ref as X

pytestmark = pytest.mark.gpu
DEV = "cuda"
RTOL = 1e-5


def T(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV)


def N_(t):
    return t.detach().cpu().numpy()


def _model_from(p, stacked=True, **kw):
    from paddlerec_amd.dcn_v2 import DCN_V2Layer
    c = X.config_of(p)
    N, D = p["embedding.weight"].shape
    sizes = [p["DNN_.linear_%d.weight" % i].shape[1] for i in range(c["n_dnn"])]
    m = DCN_V2Layer(N, D, 13, 26, sizes, c["n_cross"], is_Stacked=c["stacked"], use_low_rank_mixture=c["mix"],
                    low_rank=p[X.P + "U_list.0"].shape[2] if c["mix"] else 8, num_experts=c["n_exp"] or 4,
                    device=DEV)
    m.set_dict(p)
    return m


@pytest.mark.parametrize("name", ["dcn_v2_v2", "dcn_v2_mix"])
def test_forward_golden(engine_lib, name):
    g = load_golden(name)
    p = {k[2:]: v for k, v in g.items() if k.startswith("p.")}
    m = _model_from(p)
    sparse_inputs = [T(g["ids"][:, s:s + 1]) for s in range(26)]
    pred = m.forward(sparse_inputs, T(g["dense"]))
    np.testing.assert_allclose(N_(pred), g["pred"], rtol=RTOL, atol=2e-7)
    feat = m._feat(T(g["ids"]), T(g["dense"]))
    np.testing.assert_allclose(N_(feat), g["feat"], rtol=RTOL, atol=1e-7)
    assert np.array_equal(N_(feat)[:, :26 * 4], g["feat"][:, :26 * 4])          # the gather part: bit-exact
    assert int(m.status.item()) == 0


def test_v2_gradients_golden(engine_lib):
    """Backward chain (MLP, CrossNetV2, dense_emb, sparse rows) vs the reference's autograd gradients."""
    g = load_golden("dcn_v2_v2")
    p = {k[2:]: v for k, v in g.items() if k.startswith("p.")}
    m = _model_from(p)
    pred = g["pred"]
    dlogit = (pred * (1 - pred)).astype(np.float32)                  # fixture = d pred.sum() / d params
    label = torch.zeros(len(pred), 1, dtype=torch.int64, device=DEV)
    m.train_step(T(g["ids"]), T(g["dense"]), label, lr=0.0, clip_norm=None, dlogit=T(dlogit))
    got = m.grad_dict()
    n = 0
    for k, v in g.items():
        if k.startswith("g.") and k[2:] in got:
            assert_close_scaled(N_(got[k[2:]]), v, 1e-5, err_msg=k)
            n += 1
    assert n >= 14
    # sparse part: per-position row gradients, merged on the host for comparison with autograd's dense grad
    dfeat = N_(m._last_dfeat)[:, :26 * 4].reshape(-1, 4)
    gW = np.zeros_like(g["g.embedding.weight"])
    rows = g["ids"].reshape(-1)
    np.add.at(gW, rows[rows != 0], dfeat[rows != 0])
    assert_close_scaled(gW, g["g.embedding.weight"], 1e-5)


@pytest.mark.parametrize("stacked,D,B", [(True, 8, 200), (False, 8, 130), (True, 40, 64)])
def test_v2_train_steps_vs_oracle(engine_lib, stacked, D, B):
    """3 steps: log-loss head, ClipGradByGlobalNorm over dense + merged sparse grads, lazy Adam rows."""
    from paddlerec_amd.dcn_v2 import DCN_V2Layer
    rng = np.random.default_rng(B + D)
    N, fc = 300, [32, 16]
    m = DCN_V2Layer(N, D, 13, 26, fc, 3, is_Stacked=stacked, device=DEV)
    with torch.no_grad():                                   # biases away from zero so a bias bug cannot hide
        for k, v in m.dense.p.items():
            if k.endswith("bias"):
                v.copy_(T((rng.standard_normal(tuple(v.shape)) * 0.05).astype(np.float32)))
    p = {k: N_(v).copy() for k, v in m.state_dict().items()}
    tr = OracleDCNTrainer(p, lr=1e-2, clip_norm=0.05)       # small clip so the coefficient is < 1
    for step in range(3):
        ids = rng.integers(0, N, (B, 26), dtype=np.int64)
        dense = np.log(rng.random((B, 13), dtype=np.float32) * 50 + 1).astype(np.float32)
        label = (rng.random((B, 1)) < 0.3).astype(np.int64)
        loss, pred = m.train_step(T(ids), T(dense), T(label), lr=1e-2, clip_norm=0.05)
        oloss, opred, og = tr.train_step(ids, dense, label)
        np.testing.assert_allclose(N_(loss)[0], oloss, rtol=1e-5)
        np.testing.assert_allclose(N_(pred), opred, rtol=1e-5, atol=1e-6)
    assert int(m.status.item()) == 0
    # the optimizer state after three clipped steps: Adam's moments at 1e-5 of their scale (weights would amplify
    # eps-sized gradient noise to lr-sized differences: helpers.assert_moments_close)
    assert assert_moments_close(m, tr.m, tr.v) >= 10
    assert_close_scaled(N_(m.sparse_state["m"]), tr.m["embedding.weight"])
    assert_close_scaled(N_(m.sparse_state["v"]), tr.v["embedding.weight"])


def test_mix_gradients_golden(engine_lib):
    """CrossNetMix backward (experts, tanh chain, softmax gate shared across layers) vs the reference autograd."""
    g = load_golden("dcn_v2_mix")
    p = {k[2:]: v for k, v in g.items() if k.startswith("p.")}
    m = _model_from(p)
    pred = g["pred"]
    dlogit = (pred * (1 - pred)).astype(np.float32)
    label = torch.zeros(len(pred), 1, dtype=torch.int64, device=DEV)
    m.train_step(T(g["ids"]), T(g["dense"]), label, lr=0.0, clip_norm=None, dlogit=T(dlogit))
    got = m.grad_dict()
    n = 0
    for k, v in g.items():
        if k.startswith("g.") and k[2:] in got:
            assert_close_scaled(N_(got[k[2:]]), v, 1e-5, err_msg=k)
            n += 1
    assert n >= 24
    dfeat = N_(m._last_dfeat)[:, :26 * 4].reshape(-1, 4)
    gW = np.zeros_like(g["g.embedding.weight"])
    rows = g["ids"].reshape(-1)
    np.add.at(gW, rows[rows != 0], dfeat[rows != 0])
    assert_close_scaled(gW, g["g.embedding.weight"], 1e-5)


@pytest.mark.parametrize("stacked", [True, False])
def test_mix_train_steps_vs_oracle(engine_lib, stacked):
    from paddlerec_amd.dcn_v2 import DCN_V2Layer
    rng = np.random.default_rng(17 + stacked)
    N, D, B, fc = 200, 8, 96, [32, 16]
    m = DCN_V2Layer(N, D, 13, 26, fc, 2, is_Stacked=stacked, use_low_rank_mixture=True, low_rank=16,
                    num_experts=4, device=DEV)
    with torch.no_grad():
        for k, v in m.dense.p.items():
            if "bias" in k:
                v.copy_(T((rng.standard_normal(tuple(v.shape)) * 0.05).astype(np.float32)))
    p = {k: N_(v).copy() for k, v in m.state_dict().items()}
    tr = OracleDCNTrainer(p, lr=1e-2, clip_norm=0.05)
    for step in range(2):
        ids = rng.integers(0, N, (B, 26), dtype=np.int64)
        dense = np.log(rng.random((B, 13), dtype=np.float32) * 50 + 1).astype(np.float32)
        label = (rng.random((B, 1)) < 0.3).astype(np.int64)
        loss, pred = m.train_step(T(ids), T(dense), T(label), lr=1e-2, clip_norm=0.05)
        oloss, opred, _ = tr.train_step(ids, dense, label)
        np.testing.assert_allclose(N_(loss)[0], oloss, rtol=1e-5)
        np.testing.assert_allclose(N_(pred), opred, rtol=1e-5, atol=1e-6)
    # optimizer state: moments at 1e-5 of scale (the four gating Linear layers live in one [d, E] parameter here)
    want_m, want_v = dict(tr.m), dict(tr.v)
    for mv in (want_m, want_v):
        mv[X.P + "gating.weight"] = np.concatenate([mv[X.P + "gating.%d.weight" % e] for e in range(4)], axis=1)
        mv[X.P + "gating.bias"] = np.concatenate([mv[X.P + "gating.%d.bias" % e] for e in range(4)])
    assert assert_moments_close(m, want_m, want_v) >= 12
    assert_close_scaled(N_(m.sparse_state["m"]), tr.m["embedding.weight"])
    assert_close_scaled(N_(m.sparse_state["v"]), tr.v["embedding.weight"])


def test_emb_gather_grouped_output(engine_lib):
    from paddlerec_amd import ops
    rng = np.random.default_rng(0)
    W = rng.standard_normal((100, 8)).astype(np.float32)
    ids = rng.integers(0, 100, (7, 5), dtype=np.int64)
    out = torch.full((7, 64), -1.0, device=DEV)
    ops.emb_gather(T(ids).reshape(-1), T(W), 0, out=out, out_group=5, out_group_stride=64)
    want = W[ids] * (ids != 0)[..., None]
    assert np.array_equal(N_(out)[:, :40], want.reshape(7, 40))
    assert np.all(N_(out)[:, 40:] == -1.0)


@pytest.mark.parametrize("stacked,mix", [(True, False), (False, False), (True, True)])
def test_train_mode_dropout_and_l2_vs_oracle(engine_lib, stacked, mix):
    """The reference's TRAIN-mode graph (net.py:181-183: Dropout(0.5) after every Linear and every ReLU of the DNN
    tower; net.py:164-170: L2Decay on its weights; dygraph_model.py:81-88: clip then Adam) — three steps against the
    oracle with the same counter-based masks: loss / pred per step, then every Adam moment at 1e-5 of its scale."""
    _check_train_mode(DEV, None, stacked, mix)


@pytest.mark.parametrize("stacked,drop", [(True, 0.5), (False, 0.5), (True, 0.0)])
def test_planned_step_equals_eager_step(engine_lib, monkeypatch, stacked, drop):
    """The DCN-v2 train step replayed from its recorded call list (paddlerec_amd/plan.py: Adam step count AND the
    per-step dropout mask streams re-derived on every replay) leaves the same bits as the eager step: six steps with
    fresh inputs, train-mode dropout + L2Decay + clip."""
    from paddlerec_amd.dcn_v2 import DCN_V2Layer
    from paddlerec_amd.plan import CallPlan
    N, D, B, fc = 300, 8, 96, [32, 16]
    runs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("REC_STEP_PLAN", mode)
        torch.manual_seed(5)
        m = DCN_V2Layer(N, D, 13, 26, fc, 2, is_Stacked=stacked, device=DEV, dropout_rate=drop, dropout_seed=77,
                        l2_dnn=1e-3)
        g = torch.Generator(device=DEV).manual_seed(3)
        outs = []
        for step in range(6):
            ids = torch.randint(0, N, (B, 26), device=DEV, generator=g)
            dense = torch.rand(B, 13, device=DEV, generator=g)
            label = (torch.rand(B, 1, device=DEV, generator=g) < 0.3).to(torch.int64)
            loss, pred = m.train_step(ids, dense, label, lr=1e-2, clip_norm=0.05)
            outs.append((N_(loss).copy(), N_(pred).copy()))
        runs[mode] = (outs, N_(m.rec), N_(m.sparse_state["mv"]), N_(m.dense.data), N_(m.dense.m), N_(m.dense.v),
                      [type(p_) for p_ in m._plans.values()])
    a, b = runs["1"], runs["0"]
    assert a[-1] == [CallPlan] and b[-1] == []
    for (la, pa), (lb, pb) in zip(a[0], b[0]):
        assert np.array_equal(la, lb) and np.array_equal(pa, pb)
    for x, y in zip(a[1:6], b[1:6]):
        assert np.array_equal(x, y)


def test_dropout_kernel_matches_oracle_masks(engine_lib):
    from paddlerec_amd import ops
    rng = np.random.default_rng(1)
    x = rng.standard_normal((37, 50)).astype(np.float32)
    wide = torch.zeros(37, 64, device=DEV)
    wide[:, 3:53] = T(x)
    for p, sa, sb in ((0.5, 7, None), (0.5, 7, 8), (0.3, 100000, 3), (0.0, 1, 2), (0.5, (1 << 24) + 5, (1 << 61) + 9)):
        y = ops.dropout(wide[:, 3:53].clone(), p, 99, sa, sb)
        keep = X.dropout_keep(x.shape, p, 99, sa)
        s = np.float32(1) / (np.float32(1) - np.float32(p))
        if sb is not None:
            keep, s = keep & X.dropout_keep(x.shape, p, 99, sb), s * s
        np.testing.assert_allclose(N_(y), np.where(keep, x * s, 0), rtol=1e-6)
        assert abs(keep.mean() - (1 - p) ** (2 if sb is not None else 1)) < 0.06
    out = torch.full((37, 64), -1.0, device=DEV)
    ops.dropout(wide[:, 3:53], 0.5, 99, 7, None, out=out[:, 10:60])       # strided in / out, not in place
    assert np.all(N_(out)[:, :10] == -1) and np.all(N_(out)[:, 60:] == -1)
    g = T(rng.standard_normal(1000).astype(np.float32))
    w = T(rng.standard_normal(1000).astype(np.float32))
    sc = T(np.float32([0.25]))
    want = N_(g) + (np.float32(1e-3) / np.float32(0.25)) * N_(w)
    np.testing.assert_allclose(N_(ops.l2_decay_grad(g, w, 1e-3, sc)), want, rtol=1e-6)


def _check_train_mode(device, kernels, stacked, mix):
    from paddlerec_amd.dcn_v2 import DCN_V2Layer
    rng = np.random.default_rng(31 + stacked + 2 * mix)
    N, D, B, fc = 200, 8, 64, [32, 16]
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(device)
    m = DCN_V2Layer(N, D, 13, 26, fc, 2, is_Stacked=stacked, use_low_rank_mixture=mix, low_rank=16, num_experts=4,
                    device=device, kernels=kernels, dropout_rate=0.5, dropout_seed=4321, l2_dnn=1e-3)
    with torch.no_grad():
        for k, v in m.dense.p.items():
            if "bias" in k:
                v.copy_(t((rng.standard_normal(tuple(v.shape)) * 0.05).astype(np.float32)))
    p = {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}
    tr = OracleDCNTrainer(p, lr=1e-2, clip_norm=0.05, dropout=(0.5, 4321), l2_dnn=1e-3)

    def moments_vs_oracle():
        want_m, want_v = dict(tr.m), dict(tr.v)
        if mix:
            for mv in (want_m, want_v):
                mv[X.P + "gating.weight"] = np.concatenate([mv[X.P + "gating.%d.weight" % e] for e in range(4)], axis=1)
                mv[X.P + "gating.bias"] = np.concatenate([mv[X.P + "gating.%d.bias" % e] for e in range(4)])
        assert assert_moments_close(m, want_m, want_v) >= 8
        assert_close_scaled(m.sparse_state["m"].cpu().numpy(), tr.m["embedding.weight"])

    def oracle_takes_the_mirrors_state():
        """Every step starts from ONE state on both sides (the mirror's parameters and moments): Adam turns the fp32 noise
        of an ~eps-sized gradient into an lr-sized step, which would otherwise reach the next step's predictions (1e-4)
        — each step's loss, predictions and moments are then held to the stated bar on their own."""
        from helpers import layer_moments
        sd = {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}
        mom = layer_moments(m)
        for k in tr.p:
            tr.p[k] = sd[k].reshape(tr.p[k].shape)
            if k == "embedding.weight":
                tr.m[k] = m.sparse_state["m"].cpu().numpy().copy()
                tr.v[k] = m.sparse_state["v"].cpu().numpy().copy()
                continue
            if mix and ".gating." in k:
                e, leaf = int(k.split(".gating.")[1].split(".")[0]), k.rsplit(".", 1)[1]
                gm, gv = mom[X.P + "gating." + leaf]
                tr.m[k] = (gm[:, e:e + 1] if leaf == "weight" else gm[e:e + 1]).copy().reshape(tr.p[k].shape)
                tr.v[k] = (gv[:, e:e + 1] if leaf == "weight" else gv[e:e + 1]).copy().reshape(tr.p[k].shape)
            else:
                tr.m[k], tr.v[k] = mom[k][0].copy().reshape(tr.p[k].shape), mom[k][1].copy().reshape(tr.p[k].shape)

    for step in range(3):
        ids = rng.integers(0, N, (B, 26), dtype=np.int64)
        dense = np.log(rng.random((B, 13), dtype=np.float32) * 50 + 1).astype(np.float32)
        label = (rng.random((B, 1)) < 0.3).astype(np.int64)
        loss, pred = m.train_step(t(ids), t(dense), t(label), lr=1e-2, clip_norm=0.05)
        oloss, opred, _ = tr.train_step(ids, dense, label)
        np.testing.assert_allclose(loss.cpu().numpy()[0], oloss, rtol=1e-5)
        np.testing.assert_allclose(pred.cpu().numpy(), opred, rtol=1e-5, atol=1e-6)
        moments_vs_oracle()
        oracle_takes_the_mirrors_state()
    # eval forward is untouched by the dropout settings
    ev = m.forward(t(ids), t(dense)).cpu().numpy()
    np.testing.assert_allclose(ev, X.forward(ids, dense, {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}),
                               rtol=1e-5, atol=1e-6)
