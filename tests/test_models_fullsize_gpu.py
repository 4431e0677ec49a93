"""BASELINE-size runs of the DCN-v2 and DIN rows checked through size-independent properties (the oracle
cannot run these sizes in seconds): known answers from SURVEY.md §8(c), linearity, conservation, determinism."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_dcn_v2_crossnet_full_size_properties(engine_lib):
    """configs[2] shapes: N 1 100 001, D 40, d 1560, CrossNetV2 depth 3, B 65536."""
    from paddlerec_amd.dcn_v2 import DCN_V2Layer, P
    B = 65536
    g = torch.Generator(device=DEV).manual_seed(5)
    m = DCN_V2Layer(1100001, 40, 13, 26, [768, 768], 3, is_Stacked=True, device=DEV)
    ids = torch.randint(1, 1100001, (B, 26), device=DEV, generator=g)
    ids[torch.rand(B, 26, device=DEV, generator=g) < 0.03] = 0
    dense = torch.log(torch.rand(B, 13, device=DEV, generator=g) * 50 + 1)
    feat = m._feat(ids, dense)
    # (1) the lookup part of the feature row is exactly the gathered rows (padding -> zeros)
    want = m.embedding[ids] * (ids != 0).unsqueeze(-1)
    assert torch.equal(feat[:, :26 * 40], want.reshape(B, -1))
    # (2) known answer: W = 0  =>  X_{l+1} = X_l + X_0 * b   
    for i in range(3):
        m.dense.p[P + "cross_layers.%d.weight" % i].zero_()
        m.dense.p[P + "cross_layers.%d.bias" % i].uniform_(-0.1, 0.1, generator=g)
    x, xs, us = m._cross_v2(feat)
    ref = feat.clone()
    for i in range(3):
        ref = ref + feat * m.dense.p[P + "cross_layers.%d.bias" % i]
    torch.testing.assert_close(x, ref, rtol=1e-6, atol=1e-7)
    # (3) a real step at this size: finite, deterministic forward, loss decreases over a few steps
    m2 = DCN_V2Layer(1100001, 40, 13, 26, [768, 768], 3, is_Stacked=True, device=DEV)
    label = (torch.rand(B, 1, device=DEV, generator=g) < 0.25).long()
    p1, p2 = m2.forward(ids, dense), m2.forward(ids, dense)
    assert torch.equal(p1, p2) and bool(torch.isfinite(p1).all())
    losses = [float(m2.train_step(ids, dense, label, lr=1e-3)[0].item()) for _ in range(3)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert int(m2.status.item()) == 0
    # (4) global-norm clipping: the coefficient equals clip / max(||g||, clip) recomputed from the engine's own grads
    ss = m2._scalar("sumsq")
    sc = m2._scalar("scale")
    assert 0.0 < float(sc.item()) <= 1.0
    np.testing.assert_allclose(float(sc.item()), 10.0 / max(float(ss.item()) ** 0.5, 10.0), rtol=1e-6)


def test_din_attention_full_size_properties(engine_lib):
    """configs[3] shapes: item 63001 x 64, cat 801 x 64, B 4096, T 152 (longest history of the sample data)."""
    from paddlerec_amd import ops
    B, T = 4096, 152
    g = torch.Generator(device=DEV).manual_seed(6)
    tabs = [torch.randn(n, 64, device=DEV, generator=g) * 0.1 for n in (63001, 801, 63001, 801)]
    hi = torch.randint(0, 63001, (B, T), device=DEV, generator=g)
    hc = torch.randint(0, 801, (B, T), device=DEV, generator=g)
    lens = torch.randint(1, T + 1, (B, 1), device=DEV, generator=g)
    valid = torch.arange(T, device=DEV)[None] < lens
    mask = torch.where(valid, 0, -1000000000).long()
    aw = [torch.randn(s, device=DEV, generator=g) * 0.05 for s in ((512, 80), (80, 40), (40, 1))]
    ab = [torch.randn(s, device=DEV, generator=g) * 0.05 for s in (80, 40, 1)]
    out, attw, status = ops.din_attention_pool(hi, hc, hi, hc, mask, *tabs, aw, ab)
    assert int(status.item()) == 0
    # (1) softmax weights: non-negative, exactly zero on padding, sum to one per sample
    assert bool((attw >= 0).all()) and bool((attw[~valid] == 0).all())
    torch.testing.assert_close(attw.sum(1), torch.ones(B, device=DEV), rtol=1e-5, atol=1e-6)
    # (2) the output is the weighted sum of the gathered history rows (float64 check from the kernel's own weights)
    h = torch.cat([tabs[0][hi], tabs[1][hc]], 2).double()
    want = (attw.double().unsqueeze(2) * h).sum(1)
    torch.testing.assert_close(out.double(), want, rtol=1e-5, atol=1e-6)
    # (3) known answer: zero attention MLP  =>  mean-pool over the valid positions
    z = [torch.zeros_like(w) for w in aw]
    zb = [torch.zeros_like(b) for b in ab]
    out0, _, _ = ops.din_attention_pool(hi, hc, hi, hc, mask, *tabs, z, zb)
    want0 = (h * valid.unsqueeze(2)).sum(1) / lens.double()
    torch.testing.assert_close(out0.double(), want0, rtol=1e-5, atol=1e-6)
    # (4) backward: linear in d_out, zero on padding, and sum_t dh_t carries d_out through the pooling weights
    dout = torch.randn(B, 128, device=DEV, generator=g)
    dh1, dq1 = ops.din_attention_pool_bwd(hi, hc, hi, hc, *tabs, aw, ab, attw, dout)
    dh2, dq2 = ops.din_attention_pool_bwd(hi, hc, hi, hc, *tabs, aw, ab, attw, 2 * dout)
    torch.testing.assert_close(dh2, 2 * dh1, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(dq2, 2 * dq1, rtol=1e-5, atol=1e-7)
    assert bool((dh1[~valid] == 0).all()) and bool((dq1[~valid] == 0).all())
    # with a zero attention MLP the only path is the pooling itself: dh_t = p_t * d_out
    dh0, dq0 = ops.din_attention_pool_bwd(hi, hc, hi, hc, *tabs, z, zb, (valid / lens).float().contiguous(), dout)
    torch.testing.assert_close(dh0, (valid / lens).float().unsqueeze(2) * dout.unsqueeze(1), rtol=1e-6, atol=1e-8)
    assert float(dq0.abs().max()) == 0.0


def test_dygraph_model_mirrors_run(engine_lib):
    """The PaddleRec plugin surface (create_model / create_feeds / train_forward / infer_forward) of the three nets."""
    from paddlerec_amd import dcn_v2, deepfm, din
    from paddlerec_amd.deepfm import auc_from_buckets
    rng = np.random.default_rng(0)
    cfg = {"hyper_parameters.sparse_feature_number": 1000, "hyper_parameters.sparse_feature_dim": 8,
           "hyper_parameters.dense_input_dim": 13, "hyper_parameters.sparse_inputs_slots": 27,
           "hyper_parameters.fc_sizes": [32, 16], "hyper_parameters.cross_num": 2,
           "hyper_parameters.is_Stacked": True, "hyper_parameters.use_low_rank_mixture": False,
           "hyper_parameters.optimizer.learning_rate": 0.001, "hyper_parameters.item_emb_size": 16,
           "hyper_parameters.cat_emb_size": 16, "hyper_parameters.item_count": 300, "hyper_parameters.cat_count": 40,
           "hyper_parameters.optimizer.learning_rate_base_lr": 0.85}
    Bn = 64
    batch = [rng.integers(0, 2, (Bn, 1))] + [rng.integers(0, 1000, (Bn, 1)) for _ in range(26)] + \
            [rng.random((Bn, 13)).astype(np.float32)]
    for mod in (deepfm, dcn_v2):
        dm = mod.DygraphModel()
        net = dm.create_model(cfg, device=DEV)
        metrics, names = dm.create_metrics(device=DEV)
        loss, metrics, _ = dm.train_forward(net, metrics, batch, cfg)
        assert np.isfinite(float(loss.item())) and names == ["auc"]
        dm.infer_forward(net, metrics, batch, cfg)
        assert int(metrics[0][0].sum() + metrics[0][1].sum()) == 2 * Bn
        assert 0.0 <= auc_from_buckets(*metrics[0]) <= 1.0
    T = 9
    lens = rng.integers(1, T + 1, Bn)
    hi = np.where(np.arange(T)[None] < lens[:, None], rng.integers(1, 300, (Bn, T)), 0)
    hc = np.where(np.arange(T)[None] < lens[:, None], rng.integers(1, 40, (Bn, T)), 0)
    ti, tc = rng.integers(1, 300, Bn), rng.integers(1, 40, Bn)
    mask = np.where(np.arange(T)[None] < lens[:, None], 0, -1000000000).reshape(Bn, T, 1)
    dbatch = [hi, hc, ti, tc, rng.integers(0, 2, Bn).astype(np.float32), mask, np.repeat(ti[:, None], T, 1),
              np.repeat(tc[:, None], T, 1)]
    dm = din.DygraphModel()
    net = dm.create_model(cfg, device=DEV)
    metrics, _ = dm.create_metrics(device=DEV)
    loss, metrics, _ = dm.train_forward(net, metrics, dbatch, cfg)
    assert np.isfinite(float(loss.item()))
    dm.infer_forward(net, metrics, dbatch, cfg)
    assert int(metrics[0][0].sum() + metrics[0][1].sum()) == 2 * Bn


def test_dcn_v2_full_width_step_vs_oracle(engine_lib):
    """configs[2] at its real WIDTH against the oracle (the full-size tests checked properties only):
    D 40 => d = 39 * 40 = 1560, CrossNetV2 depth 3 (three 1560 x 1560 cross weights), DNN 768-768, B 4096 — the 256x128 /
    128x128 GEMM tiles, the 1560-wide cross epilogues and their backward at the shapes the benchmark runs.  The table is
    cut to 30 001 rows so that the NumPy oracle's merge finishes in seconds.  One clipped step: loss, predictions,
    Adam moments of every dense parameter and of the touched table rows."""
    from helpers import OracleDCNTrainer, assert_close_scaled, assert_moments_close
    from paddlerec_amd.dcn_v2 import DCN_V2Layer
    rng = np.random.default_rng(40)
    N, D, B, fc = 30001, 40, 4096, [768, 768]
    m = DCN_V2Layer(N, D, 13, 26, fc, 3, is_Stacked=True, device=DEV)
    with torch.no_grad():
        for k, v in m.dense.p.items():
            if k.endswith("bias"):
                v.copy_(torch.as_tensor((rng.standard_normal(tuple(v.shape)) * 0.05).astype(np.float32)).to(DEV))
    p = {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}
    assert p["DeepCrossLayer_.crossNet.cross_layers.0.weight"].shape == (1560, 1560)
    tr = OracleDCNTrainer(p, lr=1e-3, clip_norm=10.0)
    ids = rng.integers(0, N, (B, 26), dtype=np.int64)
    ids[rng.random((B, 26)) < 0.03] = 0
    dense = np.log(rng.random((B, 13), dtype=np.float32) * 50 + 1).astype(np.float32)
    label = (rng.random((B, 1)) < 0.3).astype(np.int64)
    T = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(DEV)
    loss, pred = m.train_step(T(ids), T(dense), T(label), lr=1e-3, clip_norm=10.0)
    oloss, opred, _ = tr.train_step(ids, dense, label)
    assert int(m.status.item()) == 0
    np.testing.assert_allclose(float(loss.item()), oloss, rtol=1e-5)
    np.testing.assert_allclose(pred.cpu().numpy(), opred, rtol=1e-5, atol=1e-6)
    assert assert_moments_close(m, tr.m, tr.v) >= 10
    assert_close_scaled(m.sparse_state["m"].cpu().numpy(), tr.m["embedding.weight"])
    assert_close_scaled(m.sparse_state["v"].cpu().numpy(), tr.v["embedding.weight"])


def test_din_attention_long_history_vs_oracle(engine_lib):
    """configs[3] shapes with the longest histories the benchmark runs (E 128, MLP 80-40-1, T 512, B 256): the compile-time-
    shaped attention kernels (16 tiles of 32 positions per sample, online softmax across them) against oracle/din_ref.py —
    output, softmax weights, and the backward's dh / dq."""
    from oracle import din_ref as Dn
    from paddlerec_amd import ops
    rng = np.random.default_rng(512)
    B, Tn, Ei, Ec, ni, nc = 256, 512, 64, 64, 5000, 301
    E = Ei + Ec
    tabs = [rng.uniform(-0.3, 0.3, (n, d)).astype(np.float32) for n, d in ((ni, Ei), (nc, Ec), (ni, Ei), (nc, Ec))]
    lens = rng.integers(1, Tn + 1, B)
    lens[:4] = (Tn, 1, 33, 480)
    hi = np.zeros((B, Tn), np.int64)
    hc = np.zeros((B, Tn), np.int64)
    for b in range(B):
        hi[b, :lens[b]] = rng.integers(1, ni, lens[b])
        hc[b, :lens[b]] = rng.integers(1, nc, lens[b])
    mask = np.where(np.arange(Tn)[None] < lens[:, None], 0, -1000000000).astype(np.int64)
    ti = np.repeat(rng.integers(1, ni, B)[:, None], Tn, 1)
    tc = np.repeat(rng.integers(1, nc, B)[:, None], Tn, 1)
    aw = [rng.uniform(-0.2, 0.2, s).astype(np.float32) for s in ((4 * E, 80), (80, 40), (40, 1))]
    ab = [rng.uniform(-0.1, 0.1, s).astype(np.float32) for s in ((80,), (40,), (1,))]
    T = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(DEV)
    tt = [T(t) for t in tabs]
    out, attw, status = ops.din_attention_pool(T(hi), T(hc), T(ti), T(tc), T(mask), *tt, [T(w) for w in aw],
                                               [T(b) for b in ab])
    assert int(status.item()) == 0
    h = np.concatenate([tabs[0][hi], tabs[1][hc]], 2)
    q = np.concatenate([tabs[2][ti], tabs[3][tc]], 2)
    want, wts = Dn.attention_pool(h, q, mask.astype(np.float32), aw, ab, return_weights=True)
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(attw.cpu().numpy(), wts.reshape(B, Tn), rtol=1e-5, atol=1e-7)
    dout = (rng.standard_normal((B, E)) * 0.1).astype(np.float32)
    dh, dq = ops.din_attention_pool_bwd(T(hi), T(hc), T(ti), T(tc), *tt, [T(w) for w in aw], [T(b) for b in ab], attw,
                                        T(dout))
    ref = Dn.attention_pool_backward(h.astype(np.float64), q.astype(np.float64), mask.astype(np.float64),
                                     [w.astype(np.float64) for w in aw], [b.astype(np.float64) for b in ab],
                                     dout.astype(np.float64))
    ref32 = Dn.attention_pool_backward(h, q, mask.astype(np.float32), aw, ab, dout)
    from helpers import assert_close_floor
    scale = max(np.abs(ref["dh"]).max(), np.abs(ref["dq"]).max())
    assert_close_floor(dh.cpu().numpy(), ref["dh"], ref32["dh"], err_msg="dh", scale=scale)
    assert_close_floor(dq.cpu().numpy(), ref["dq"], ref32["dq"], err_msg="dq", scale=scale)
