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
