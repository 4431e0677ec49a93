"""rank/fm sibling net (paddlerec_amd/fm.py; reference: models/rank/fm/net.py, fm/dygraph_model.py).

Oracle (oracle/fm_ref.py) pinned to tests/golden/fm_D9.npz = the reference's unmodified net.py executed over the
paddle shim.  The host mirror is checked against golden + oracle twice: with the oracle-backed operator backend on
the CPU (orchestration only) and with the HIP kernels (`-m gpu`)."""
import os

import numpy as np
import pytest
import torch

from helpers import load_golden
from oracle import deepfm_ref as R
from oracle import fm_ref

KEYS = ("W", "W1", "dense_w", "dense_w_one", "bias")


def _params(g):
    return {k: g[k].copy() for k in KEYS}


def _state_dict(p):
    return {"fm.embedding.weight": p["W"], "fm.embedding_one.weight": p["W1"], "fm.dense_w": p["dense_w"],
            "fm.dense_w_one": p["dense_w_one"], "bias": p["bias"]}


def test_oracle_matches_reference_golden():
    g = load_golden("fm_D9")
    o = fm_ref.fm_loss_and_grads(g["ids"], g["dense"], g["label"], _params(g))
    np.testing.assert_allclose(o["pred"], g["pred"], rtol=1e-6)
    np.testing.assert_allclose(o["y1"], g["y1"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(o["y2"], g["y2"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(o["loss"], g["loss"], rtol=1e-6)
    np.testing.assert_allclose(o["d_bias"], g["g_bias"], rtol=1e-5)
    np.testing.assert_allclose(o["d_dense_w"].reshape(g["g_dense_w"].shape), g["g_dense_w"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(o["d_dense_w_one"], g["g_dense_w_one"], rtol=1e-5, atol=1e-7)
    for key, gk in (("row_grad", "gW"), ("row_grad1", "gW1")):
        uniq, merged, _ = R.merge_rows(o["rows"], o["row_valid"], o[key])
        dense_g = np.zeros_like(g[gk])
        dense_g[uniq] = merged
        np.testing.assert_allclose(dense_g, g[gk], rtol=1e-5, atol=1e-7)
    assert np.abs(g["gW"][0]).max() > 0          # fm/net.py:55-73: no padding_idx — row 0 is looked up and trained


class _OracleFMTrainer:
    def __init__(self, p, lr):
        self.p = {k: v.copy() for k, v in p.items()}
        self.lr, self.step = lr, 0
        self.st = {k: (np.zeros_like(v), np.zeros_like(v)) for k, v in self.p.items()}

    def train_step(self, ids, dense, label):
        self.step += 1
        o = fm_ref.fm_loss_and_grads(ids, dense, label, self.p)
        for key, gk in (("W", "row_grad"), ("W1", "row_grad1")):
            uniq, merged, _ = R.merge_rows(o["rows"], o["row_valid"], o[gk])
            R.adam_update_rows(self.p[key], self.st[key][0], self.st[key][1], uniq, merged, self.step, lr=self.lr)
        for key, gr in (("dense_w", o["d_dense_w"]), ("dense_w_one", o["d_dense_w_one"]), ("bias", o["d_bias"])):
            R.adam_update(self.p[key], self.st[key][0], self.st[key][1],
                          gr.reshape(self.p[key].shape).astype(np.float32), self.step, lr=self.lr)
        return o["loss"], o["pred"]


def _check_layer(device, kernels, tol):
    from paddlerec_amd.fm import DygraphModel, FMLayer
    rtol, p_atol = tol
    g = load_golden("fm_D9")
    N, D = g["W"].shape
    T = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(device)
    m = FMLayer(N, D, 13, 26, device=device, kernels=kernels)
    assert float(m.dense.p["fm.dense_w"].min()) == 1.0 and float(m.dense.p["bias"]) == 0.0   # net.py:25-29,78-88
    m.set_dict(_state_dict(_params(g)))
    sparse_inputs = [T(g["ids"][:, s:s + 1]) for s in range(26)]                 # the reference's list of [B,1]
    pred = m.forward(sparse_inputs, T(g["dense"]))
    np.testing.assert_allclose(pred.cpu().numpy(), g["pred"], rtol=rtol)
    # three optimizer steps against the oracle (fresh batches; ids include 0 = an ordinary row here)
    tr = _OracleFMTrainer(_params(g), lr=1e-2)
    rng = np.random.default_rng(3)
    for step in range(3):
        ids = rng.integers(0, N, (64, 26), dtype=np.int64)
        ids[:, 0] = 0                                                             # heavy duplicates on row 0
        dense = rng.random((64, 13), dtype=np.float32)
        label = (rng.random((64, 1)) < 0.3).astype(np.int64)
        loss, pred = m.train_step(T(ids), T(dense), T(label), lr=1e-2)
        ol, op = tr.train_step(ids, dense, label)
        np.testing.assert_allclose(loss.cpu().numpy()[0], ol, rtol=rtol)
        np.testing.assert_allclose(pred.cpu().numpy(), op, rtol=rtol, atol=1e-6)
    assert int(m.status.item()) == 0
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    for k, ok in (("fm.embedding.weight", "W"), ("fm.embedding_one.weight", "W1"), ("fm.dense_w", "dense_w"),
                  ("fm.dense_w_one", "dense_w_one"), ("bias", "bias")):
        from helpers import assert_adam_weights_close
        assert_adam_weights_close(sd[k], tr.p[ok], lr=1e-2, steps=3, err_msg=k)
    # ... and the optimizer state at the stated bar: Adam's moments within 1e-5 of their scale (the weights above carry
    # lr-sized differences wherever a gradient is ~eps-sized: helpers.assert_moments_close)
    from helpers import assert_sibling_moments
    assert assert_sibling_moments(m, tr.st) >= 3
    # plugin surface with the reference's 28-array batch
    dm = DygraphModel()
    cfg = {"hyper_parameters.sparse_feature_number": N, "hyper_parameters.sparse_feature_dim": D,
           "hyper_parameters.dense_input_dim": 13, "hyper_parameters.sparse_inputs_slots": 27,
           "hyper_parameters.optimizer.learning_rate": 0.001}
    net = dm.create_model(cfg, device, kernels=kernels)
    metrics, names = dm.create_metrics(device)
    batch = [g["label"]] + [g["ids"][:, s:s + 1] for s in range(26)] + [g["dense"]]
    loss, metrics, _ = dm.train_forward(net, metrics, batch, cfg)
    dm.infer_forward(net, metrics, batch, cfg)
    assert np.isfinite(float(loss.reshape(-1)[0])) and names == ["auc"]
    assert int(metrics[0][0].sum() + metrics[0][1].sum()) == 2 * len(g["label"])


def test_fm_layer_host_logic_cpu_backend():
    import cpu_kernels
    _check_layer("cpu", cpu_kernels, (1e-6, 1e-6))


@pytest.mark.gpu
def test_fm_layer_gpu(engine_lib):
    _check_layer("cuda", None, (1e-5, 2e-4))
