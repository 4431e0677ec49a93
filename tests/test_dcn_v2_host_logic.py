"""DCN-v2 host mirror (paddlerec_amd/dcn_v2.py) with the oracle-backed operator stand-in on the CPU: the
orchestration of both cross networks (forward, the explicit backward chain, gate softmax shared across layers, the
clipping norm over dense + merged sparse gradients, optimizer calls) against the golden fixtures of the reference's
dcn_v2/net.py and the NumPy oracle — no kernel is involved (tests/test_dcn_v2_gpu.py runs the same checks on the HIP
kernels)."""
import numpy as np
import pytest
import torch

import cpu_kernels
from helpers import OracleDCNTrainer, load_golden
from oracle import dcn_v2_ref as X

T = lambda a: torch.as_tensor(np.ascontiguousarray(a))
N_ = lambda t: t.detach().numpy()


def _model_from(p):
    from paddlerec_amd.dcn_v2 import DCN_V2Layer
    c = X.config_of(p)
    N, D = p["embedding.weight"].shape
    sizes = [p["DNN_.linear_%d.weight" % i].shape[1] for i in range(c["n_dnn"])]
    m = DCN_V2Layer(N, D, 13, 26, sizes, c["n_cross"], is_Stacked=c["stacked"], use_low_rank_mixture=c["mix"],
                    low_rank=p[X.P + "U_list.0"].shape[2] if c["mix"] else 8, num_experts=c["n_exp"] or 4,
                    device="cpu", kernels=cpu_kernels)
    m.set_dict(p)
    return m


@pytest.mark.parametrize("name,min_grads", [("dcn_v2_v2", 14), ("dcn_v2_mix", 24)])
def test_forward_and_gradients_golden(name, min_grads):
    g = load_golden(name)
    p = {k[2:]: v for k, v in g.items() if k.startswith("p.")}
    m = _model_from(p)
    pred = m.forward([T(g["ids"][:, s:s + 1]) for s in range(26)], T(g["dense"]))
    np.testing.assert_allclose(N_(pred), g["pred"], rtol=1e-5, atol=2e-7)
    dlogit = (g["pred"] * (1 - g["pred"])).astype(np.float32)          # fixture = d pred.sum() / d params
    label = torch.zeros(len(g["pred"]), 1, dtype=torch.int64)
    m.train_step(T(g["ids"]), T(g["dense"]), label, lr=0.0, clip_norm=None, dlogit=T(dlogit))
    got = m.grad_dict()
    n = 0
    for k, v in g.items():
        if k.startswith("g.") and k[2:] in got:
            np.testing.assert_allclose(N_(got[k[2:]]).reshape(v.shape), v, rtol=3e-4, atol=3e-6, err_msg=k)
            n += 1
    assert n >= min_grads
    D = p["embedding.weight"].shape[1]
    dfeat = N_(m._last_dfeat)[:, :26 * D].reshape(-1, D)
    gW = np.zeros_like(g["g.embedding.weight"])
    rows = g["ids"].reshape(-1)
    np.add.at(gW, rows[rows != 0], dfeat[rows != 0])
    np.testing.assert_allclose(gW, g["g.embedding.weight"], rtol=3e-4, atol=3e-6)


@pytest.mark.parametrize("mix,stacked", [(False, True), (False, False), (True, True), (True, False)])
def test_train_steps_vs_oracle(mix, stacked):
    from paddlerec_amd.dcn_v2 import DCN_V2Layer
    rng = np.random.default_rng(7 + 2 * mix + stacked)
    N, D, B, fc = 120, 4, 48, [16, 8]
    m = DCN_V2Layer(N, D, 13, 26, fc, 2, is_Stacked=stacked, use_low_rank_mixture=mix, low_rank=8, num_experts=3,
                    device="cpu", kernels=cpu_kernels)
    with torch.no_grad():
        for k, v in m.dense.p.items():
            if "bias" in k:
                v.copy_(T((rng.standard_normal(tuple(v.shape)) * 0.05).astype(np.float32)))
    p = {k: N_(v).copy() for k, v in m.state_dict().items()}
    tr = OracleDCNTrainer(p, lr=1e-2, clip_norm=0.05)
    for _ in range(3):
        ids = rng.integers(0, N, (B, 26), dtype=np.int64)
        dense = np.log(rng.random((B, 13), dtype=np.float32) * 50 + 1).astype(np.float32)
        label = (rng.random((B, 1)) < 0.3).astype(np.int64)
        loss, pred = m.train_step(T(ids), T(dense), T(label), lr=1e-2, clip_norm=0.05)
        oloss, opred, _ = tr.train_step(ids, dense, label)
        np.testing.assert_allclose(N_(loss)[0], oloss, rtol=2e-5)
        np.testing.assert_allclose(N_(pred), opred, rtol=2e-5, atol=1e-6)
    sd = m.state_dict()
    for k in tr.p:
        if k in sd:
            np.testing.assert_allclose(N_(sd[k]), tr.p[k].reshape(tuple(sd[k].shape)), rtol=1e-3, atol=1e-3, err_msg=k)


@pytest.mark.parametrize("stacked,mix", [(True, False), (False, False), (False, True)])
def test_train_mode_dropout_and_l2_host_logic(stacked, mix):
    """tests/test_dcn_v2_gpu.py::_check_train_mode on the CPU operator stand-in: the train-mode orchestration (which
    mask stream goes where, the gradient's way back through both dropouts, clip -> L2Decay -> Adam)."""
    import cpu_kernels
    import test_dcn_v2_gpu as G
    G._check_train_mode("cpu", cpu_kernels, stacked, mix)
