"""rank/dlrm sibling net (paddlerec_amd/dlrm.py + csrc/dlrm_ops.hip; reference: models/rank/dlrm/net.py, dygraph_model.py).

Oracle (oracle/dlrm_ref.py) pinned to tests/golden/dlrm_D16.npz = the reference's unmodified net.py executed over the
paddle shim in train mode (raw scores, loss, every autograd gradient, the BatchNorm running statistics after the step,
and an eval-mode forward).  The host mirror is checked against golden + oracle with the oracle-backed operator backend
on the CPU (orchestration only) and with the HIP kernels (`-m gpu`); the BatchNorm / interaction / accuracy kernels
themselves against numpy."""
import numpy as np
import pytest
import torch

from helpers import assert_adam_weights_close, load_golden
from oracle import deepfm_ref as R
from oracle import dlrm_ref as Dr


def _layers(g, name):
    out = []
    for i in range(int(g["n_" + name])):
        k = "%s%d_" % (name, i)
        out.append(dict(w=g[k + "w"].copy(), b=g[k + "b"].copy(), gamma=g[k + "gamma"].copy(), beta=g[k + "beta"].copy(),
                        mean=g[k + "mean0"].copy(), var=g[k + "var0"].copy()))
    return out


def _params(g):
    return dict(W=g["W"].copy(), bot=_layers(g, "bot"), top=_layers(g, "top"))


def _state_dict(p):
    sd = {"embedding.weight": p["W"]}
    for name, key in (("bot", "bot_mlp"), ("top", "top_mlp")):
        for i, L in enumerate(p[name]):
            sd["%s.dense_%d.weight" % (key, i)], sd["%s.dense_%d.bias" % (key, i)] = L["w"], L["b"]
            sd["%s.norm_%d.weight" % (key, i)], sd["%s.norm_%d.bias" % (key, i)] = L["gamma"], L["beta"]
            sd["%s.norm_%d._mean" % (key, i)], sd["%s.norm_%d._variance" % (key, i)] = L["mean"], L["var"]
    return sd


def test_oracle_matches_reference_golden():
    g = load_golden("dlrm_D16")
    p = _params(g)
    o = Dr.loss_and_grads(g["ids"], g["dense"], g["label"], p)
    np.testing.assert_allclose(o["raw"], g["raw"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(o["loss"], g["loss"], rtol=1e-6)
    for name in ("bot", "top"):
        for i in range(int(g["n_" + name])):
            k, G = "%s%d_" % (name, i), o[name][i]
            for mine, ref in (("dw", "gw"), ("db", "gb"), ("dgamma", "ggamma"), ("dbeta", "gbeta")):
                want = g[k + ref]
                np.testing.assert_allclose(G[mine], want, rtol=1e-5, atol=1e-5 * max(np.abs(want).max(), 1e-3),
                                           err_msg=k + mine)
            np.testing.assert_allclose(p[name][i]["mean"], g[k + "mean1"], rtol=1e-6, atol=1e-7)   # running stats moved
            np.testing.assert_allclose(p[name][i]["var"], g[k + "var1"], rtol=1e-6, atol=1e-7)
    gW = np.zeros_like(g["gW"])
    np.add.at(gW, o["rows"], o["row_grad"])
    np.testing.assert_allclose(gW, g["gW"], rtol=1e-5, atol=1e-5 * np.abs(g["gW"]).max())
    assert np.abs(g["gW"][0]).max() > 0          # net.py:70-77: no padding_idx — row 0 is looked up and trained
    raw_eval, _ = Dr.forward(g["ids"], g["dense"], p, training=False)          # eval: the (moved) running statistics
    np.testing.assert_allclose(raw_eval, g["raw_eval"], rtol=1e-5, atol=2e-6)


class _OracleTrainer:
    """paddle.optimizer.Adam(parameters=...) (dygraph_model.py:60-64): non-lazy — every table row's moments move."""

    def __init__(self, p, lr):
        self.p, self.lr, self.step, self.st = p, lr, 0, {}

    def _adam(self, key, arr, grad):
        m, v = self.st.setdefault(key, (np.zeros_like(arr), np.zeros_like(arr)))
        R.adam_update(arr, m, v, grad.reshape(arr.shape).astype(arr.dtype), self.step, lr=self.lr)

    def train_step(self, ids, dense, label):
        self.step += 1
        o = Dr.loss_and_grads(ids, dense, label, self.p)
        gW = np.zeros_like(self.p["W"])
        uniq, merged, _ = R.merge_rows(o["rows"], np.ones(len(o["rows"]), bool), o["row_grad"])
        gW[uniq] = merged
        self._adam("W", self.p["W"], gW)
        for name in ("bot", "top"):
            for i, L in enumerate(self.p[name]):
                G = o[name][i]
                for key, gk in (("w", "dw"), ("b", "db"), ("gamma", "dgamma"), ("beta", "dbeta")):
                    self._adam((name, i, key), L[key], G[gk])
        return o["loss"], o["pred"]


def _check_layer(device, kernels, tol):
    from paddlerec_amd.dlrm import DLRMLayer, DygraphModel
    rtol, p_atol = tol
    g = load_golden("dlrm_D16")
    N, D = g["W"].shape
    T = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(device)
    m = DLRMLayer(13, [32, D], N, D, [48, 2], 26, device=device, kernels=kernels)
    assert set(m.state_dict()) == set(str(k) for k in g["state_keys"])              # the reference's own key set
    m.set_dict(_state_dict(_params(g)))
    m.eval()
    # eval forward with the golden's POST-step running statistics
    p1 = _params(g)
    for name in ("bot", "top"):
        for i, L in enumerate(p1[name]):
            L["mean"], L["var"] = g["%s%d_mean1" % (name, i)].copy(), g["%s%d_var1" % (name, i)].copy()
    m.set_dict(_state_dict(p1))
    raw = m.forward([T(g["ids"][:, s:s + 1]) for s in range(26)], T(g["dense"]))     # the reference's list of [B,1]
    np.testing.assert_allclose(raw.cpu().numpy(), g["raw_eval"], rtol=rtol, atol=10 * rtol)
    # one training step on the golden batch with lr 0: loss, gradient buffers, moved running statistics
    m.set_dict(_state_dict(_params(g)))
    m.train()
    loss, _ = m.train_step(T(g["ids"]), T(g["dense"]), T(g["label"]), lr=0.0)
    np.testing.assert_allclose(loss.cpu().numpy()[0], g["loss"], rtol=rtol)
    for name, key in (("bot", "bot_mlp"), ("top", "top_mlp")):
        for i in range(int(g["n_" + name])):
            k = "%s%d_" % (name, i)
            for mine, ref in (("dense_%d.weight", "gw"), ("dense_%d.bias", "gb"), ("norm_%d.weight", "ggamma"),
                              ("norm_%d.bias", "gbeta")):
                want = g[k + ref]
                got = m.dense.g[key + "." + mine % i].cpu().numpy()
                # 2e-5 of the gradient tensor's scale (BatchNorm cancels a bias shift exactly: those entries are pure
                # round-off around zero, which an element-wise rtol cannot describe)
                np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5 * max(np.abs(want).max(), 1e-3),
                                           err_msg=k + ref)
            np.testing.assert_allclose(m.buffers["%s.norm_%d._mean" % (key, i)].cpu().numpy(), g[k + "mean1"],
                                       rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(m.buffers["%s.norm_%d._variance" % (key, i)].cpu().numpy(), g[k + "var1"],
                                       rtol=1e-5, atol=1e-6)
    # Adam steps against the oracle trainer.  A bias in front of ReLU + BatchNorm whose unit is active on the whole batch
    # has a mathematically ZERO gradient (BatchNorm removes the shift): numerically ~1e-9 of either sign, which Adam's
    # first steps turn into +-lr moves — chaotic in any implementation, Paddle's included.  So every step starts from
    # parameters and moments shared by both sides, parameters are compared where the gradient is not noise, and the
    # noise-driven ones only have to stay within one Adam step.
    lr = 1e-2
    rng = np.random.default_rng(4)
    m = DLRMLayer(13, [32, D], N, D, [48, 2], 26, device=device, kernels=kernels)
    m.set_dict(_state_dict(_params(g)))
    tr = _OracleTrainer(_params(g), lr=lr)
    for step in range(3):
        ids = rng.integers(0, N, (48, 26), dtype=np.int64)
        ids[:, 3] = 0                                                             # heavy duplicates on row 0
        dense = rng.random((48, 13), dtype=np.float32)
        label = (rng.random((48, 1)) < 0.4).astype(np.int64)
        before = {k: v.copy() for k, v in _state_dict(tr.p).items()}
        o = Dr.loss_and_grads(ids, dense, label, {"W": tr.p["W"].copy(), "bot": [dict(L, mean=L["mean"].copy(),
                              var=L["var"].copy()) for L in tr.p["bot"]], "top": [dict(L, mean=L["mean"].copy(),
                              var=L["var"].copy()) for L in tr.p["top"]]})
        loss, pred = m.train_step(T(ids), T(dense), T(label), lr=lr)
        ol, op = tr.train_step(ids, dense, label)
        np.testing.assert_allclose(loss.cpu().numpy()[0], ol, rtol=10 * rtol, err_msg="step %d" % step)
        np.testing.assert_allclose(pred.cpu().numpy(), op, rtol=10 * rtol, atol=1e-5)
        sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
        want = _state_dict(tr.p)
        grads = {"embedding.weight": None}
        for name, key in (("bot", "bot_mlp"), ("top", "top_mlp")):
            for i, G in enumerate(o[name]):
                grads["%s.dense_%d.weight" % (key, i)], grads["%s.dense_%d.bias" % (key, i)] = G["dw"], G["db"]
                grads["%s.norm_%d.weight" % (key, i)], grads["%s.norm_%d.bias" % (key, i)] = G["dgamma"], G["dbeta"]
        for k in want:
            w = want[k].reshape(sd[k].shape)
            gk = grads.get(k)
            if gk is None:                      # table (dense gradient rows, real magnitudes) and running statistics
                assert_adam_weights_close(sd[k], w, lr=lr, steps=1, err_msg=k)
                continue
            live = np.abs(gk.reshape(w.shape)) > 1e-6 * max(np.abs(gk).max(), 1e-12)
            assert_adam_weights_close(sd[k][live], w[live], lr=lr, steps=1, err_msg=k)
            assert np.all(np.abs(sd[k] - before[k].reshape(w.shape)) <= lr * 1.01), k      # at most one Adam step
        # next step: both sides continue from the SAME point (the mirror's), moments included
        tr.p = None
        cur = {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}
        p2 = _params(g)
        p2["W"] = cur["embedding.weight"]
        for name, key in (("bot", "bot_mlp"), ("top", "top_mlp")):
            for i, L in enumerate(p2[name]):
                L["w"], L["b"] = cur["%s.dense_%d.weight" % (key, i)], cur["%s.dense_%d.bias" % (key, i)]
                L["gamma"], L["beta"] = cur["%s.norm_%d.weight" % (key, i)], cur["%s.norm_%d.bias" % (key, i)]
                L["mean"], L["var"] = cur["%s.norm_%d._mean" % (key, i)], cur["%s.norm_%d._variance" % (key, i)]
        tr.p = p2
        Dd = m.sparse_feature_dim
        tr.st = {"W": (m.sparse_state["m"].cpu().numpy().copy(), m.sparse_state["v"].cpu().numpy().copy())}
        for nm in m.dense.names:
            key, rest = nm.split(".", 1)
            kind, attr = rest.split(".")
            i = int(kind.split("_")[1])
            lk = {("dense", "weight"): "w", ("dense", "bias"): "b", ("norm", "weight"): "gamma",
                  ("norm", "bias"): "beta"}[(kind.split("_")[0], attr)]
            tr.st[("bot" if key == "bot_mlp" else "top", i, lk)] = (m.dense.pm[nm].cpu().numpy().copy(),
                                                                    m.dense.pv[nm].cpu().numpy().copy())
    assert int(m.status.item()) == 0
    dm = DygraphModel()
    cfg = {"hyper_parameters.sparse_feature_number": N, "hyper_parameters.sparse_feature_dim": D,
           "hyper_parameters.dense_input_dim": 13, "hyper_parameters.sparse_inputs_slots": 27,
           "hyper_parameters.num_field": 26, "hyper_parameters.bot_layer_sizes": [32, D],
           "hyper_parameters.top_layer_sizes": [48, 2], "hyper_parameters.optimizer.learning_rate": 0.001}
    net = dm.create_model(cfg, device, kernels=kernels)
    metrics, names = dm.create_metrics(device)
    batch = [g["label"]] + [g["ids"][:, s:s + 1] for s in range(26)] + [g["dense"]]
    loss, metrics, _ = dm.train_forward(net, metrics, batch, cfg)
    dm.infer_forward(net, metrics, batch, cfg)
    assert np.isfinite(float(loss.reshape(-1)[0])) and names == ["auc", "accuracy"]
    assert int(metrics[0][0].sum() + metrics[0][1].sum()) == 2 * len(g["label"])
    assert int(metrics[1][1]) == 2 * len(g["label"]) and 0 <= int(metrics[1][0]) <= int(metrics[1][1])
    assert 0.0 <= dm.metric_value("accuracy", metrics[1]) <= 1.0 and 0.0 <= dm.metric_value("auc", metrics[0]) <= 1.0


def test_dlrm_layer_host_logic_cpu_backend():
    import cpu_kernels
    _check_layer("cpu", cpu_kernels, (2e-6, 5e-6))


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,relu", [(24, 32, True), (1000, 513, False), (4099, 16, True), (1, 5, False),
                                      (70000, 64, True)])
def test_batchnorm_kernels_vs_numpy(engine_lib, M, N, relu):
    from paddlerec_amd import ops
    rng = np.random.default_rng(M + N)
    x = (rng.standard_normal((M, N)) * 2 + 3).astype(np.float32)
    if relu:
        x = np.maximum(x - 3, 0)
    gamma, beta = (1 + 0.3 * rng.standard_normal(N)).astype(np.float32), rng.standard_normal(N).astype(np.float32)
    rm, rv = rng.standard_normal(N).astype(np.float32), (1 + rng.random(N)).astype(np.float32)
    T = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
    ws = ops.Workspace("cuda")
    xw = torch.zeros(M, N + 3, device="cuda")                 # strided input and output
    xw[:, 1:1 + N] = T(x)
    yw = torch.zeros(M, N + 5, device="cuda")
    trm, trv = T(rm), T(rv)
    y, sm, si = ops.batchnorm_fwd(xw[:, 1:1 + N], T(gamma), T(beta), trm, trv, ws, True, out=yw[:, 2:2 + N])
    x64 = x.astype(np.float64)
    mean, var = x64.mean(0), x64.var(0)
    want = (x64 - mean) / np.sqrt(var + 1e-5) * gamma + beta
    np.testing.assert_allclose(y.cpu().numpy(), want, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(sm.cpu().numpy(), mean, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(si.cpu().numpy(), 1 / np.sqrt(var + 1e-5), rtol=2e-5)
    np.testing.assert_allclose(trm.cpu().numpy(), 0.9 * rm + 0.1 * mean, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(trv.cpu().numpy(), 0.9 * rv + 0.1 * var, rtol=1e-5, atol=1e-6)
    assert float(yw[:, :2].abs().max()) == 0 and float(yw[:, 2 + N:].abs().max()) == 0
    # eval: the running statistics
    ye, _, _ = ops.batchnorm_fwd(xw[:, 1:1 + N], T(gamma), T(beta), trm, trv, ws, False)
    r_m, r_v = trm.cpu().numpy().astype(np.float64), trv.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(ye.cpu().numpy(), (x64 - r_m) / np.sqrt(r_v + 1e-5) * gamma + beta, rtol=2e-5, atol=2e-5)
    # backward
    dy = rng.standard_normal((M, N)).astype(np.float32)
    dx, dg, db = ops.batchnorm_bwd(xw[:, 1:1 + N], T(dy), T(gamma), sm, si, ws, relu_mask=relu)
    xhat = (x64 - mean) / np.sqrt(var + 1e-5)
    wdb, wdg = dy.astype(np.float64).sum(0), (dy * xhat).sum(0)
    wdx = gamma / np.sqrt(var + 1e-5) * (dy - wdb / M - xhat * wdg / M)
    if relu:
        wdx = wdx * (x > 0)
    scale = max(np.abs(wdx).max(), 1e-3)
    np.testing.assert_allclose(dx.cpu().numpy(), wdx, rtol=2e-5, atol=2e-5 * scale)      # float64 reference
    np.testing.assert_allclose(dg.cpu().numpy(), wdg, rtol=2e-5, atol=1e-5 * max(np.abs(wdg).max(), 1))
    np.testing.assert_allclose(db.cpu().numpy(), wdb, rtol=2e-5, atol=1e-5 * max(np.abs(wdb).max(), 1))


@pytest.mark.gpu
@pytest.mark.parametrize("B,F,D", [(5, 27, 16), (1, 2, 1), (300, 27, 16), (33, 9, 7), (4, 40, 32)])
def test_dot_interact_vs_oracle(engine_lib, B, F, D):
    from paddlerec_amd import ops
    rng = np.random.default_rng(B * F)
    t = rng.standard_normal((B, F, D)).astype(np.float32)
    T = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
    R_ = ops.dot_interact_fwd(T(t))
    want = Dr.dot_interact(t.astype(np.float64))
    np.testing.assert_allclose(R_.cpu().numpy(), want, rtol=1e-5, atol=1e-5)
    assert np.array_equal(R_.cpu().numpy()[:, :D], t[:, F - 1, :])                 # x is copied, bit-exact
    dR = rng.standard_normal(want.shape).astype(np.float32)
    dT = ops.dot_interact_bwd(T(t), T(dR))
    np.testing.assert_allclose(dT.cpu().numpy(), Dr.dot_interact_backward(t.astype(np.float64), dR.astype(np.float64)),
                               rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_accuracy_count(engine_lib):
    from paddlerec_amd import ops
    rng = np.random.default_rng(3)
    pred = rng.random(10007).astype(np.float32)
    pred[:5] = 0.5                                              # a tie is class 0 (argmax picks the first)
    label = (rng.random(10007) < 0.5).astype(np.int64)
    c = torch.tensor([7, 11], dtype=torch.int64, device="cuda")
    ops.accuracy_count(torch.as_tensor(pred).cuda(), torch.as_tensor(label).cuda(), c)
    assert c.tolist() == [7 + int(((pred > 0.5) == (label != 0)).sum()), 11 + 10007]


@pytest.mark.gpu
def test_dlrm_layer_gpu(engine_lib):
    _check_layer("cuda", None, (2e-5, 2e-4))
