#!/usr/bin/env python3
"""Generate tests/golden/*.npz by executing the reference's UNMODIFIED net.py files.

Runs only in the build container (needs /root/reference); the GPU box uses the committed
fixtures.  `paddle` resolves to oracle/paddle_shim (torch-CPU stand-in), so
what is pinned is the reference's own graph code — op order, shapes, padding handling, the DIN
duplicate-sublayer quirk — not Paddle's kernels (those are [EXT], not installable here).

    python oracle/make_golden.py [name ...]  # rewrites tests/golden/*.npz (all, or the named ones) deterministically
"""
import importlib.util
import os
import sys
sys.dont_write_bytecode = True      # importing the reference's net.py files must not write __pycache__ next to them

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get("PADDLEREC_REF", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, os.path.join(HERE, "paddle_shim"))


def load_ref_module(rel, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def npy(t):
    return t.detach().cpu().numpy().copy()


def make_ids(rng, B, S, N, pad_frac=0.1, dup=True):
    ids = rng.integers(1, N, size=(B, S), dtype=np.int64)
    if dup:  # force duplicate rows inside the batch (SelectedRows merge path)
        ids[1::2, : S // 2] = ids[0::2, : S // 2][: ids[1::2].shape[0]]
    ids[rng.random((B, S)) < pad_frac] = 0
    return ids


def golden_deepfm(D, seed):
    """models/rank/deepfm/net.py:21-174 + dygraph_model.py:53-58 (loss)."""
    import paddle  # the shim
    net = load_ref_module("models/rank/deepfm/net.py", "ref_deepfm_net")
    rng = np.random.default_rng(seed)
    N, S, Dn, B, fc = 1001, 26, 13, 12, [32, 16]
    torch.manual_seed(seed)
    model = net.DeepFMLayer(N, D, Dn, S, fc)
    ids = make_ids(rng, B, S, N)
    dense = rng.random((B, Dn), dtype=np.float32)
    label = (rng.random((B, 1)) < 0.3).astype(np.int64)
    sparse_inputs = [paddle.to_tensor(ids[:, s:s + 1]) for s in range(S)]   # list of [B,1] int64
    pred = model.forward(sparse_inputs, paddle.to_tensor(dense))
    # deepfm/dygraph_model.py:53-58
    cost = paddle.nn.functional.log_loss(input=pred, label=paddle.cast(paddle.to_tensor(label),
                                                                        dtype="float32"))
    loss = paddle.mean(x=cost)
    y1, y2, feat = model.fm.forward(sparse_inputs, paddle.to_tensor(dense))
    loss.backward()
    g = dict(
        ids=ids, dense=dense, label=label, D=np.int64(D),
        W=npy(model.fm.embedding.weight), W1=npy(model.fm.embedding_one.weight),
        dense_w=npy(model.fm.dense_w), dense_w_one=npy(model.fm.dense_w_one),
        pred=npy(pred), loss=npy(loss), y1=npy(y1), y2=npy(y2), feat=npy(feat),
        gW=npy(model.fm.embedding.weight.grad), gW1=npy(model.fm.embedding_one.weight.grad),
        g_dense_w=npy(model.fm.dense_w.grad), g_dense_w_one=npy(model.fm.dense_w_one.grad),
    )
    lin = [m for m in model.dnn._mlp_layers if hasattr(m, "weight")]
    for i, l in enumerate(lin):
        g[f"mlp_w{i}"], g[f"mlp_b{i}"] = npy(l.weight), npy(l.bias)
        g[f"g_mlp_w{i}"], g[f"g_mlp_b{i}"] = npy(l.weight.grad), npy(l.bias.grad)
    g["n_mlp"] = np.int64(len(lin))
    np.savez_compressed(os.path.join(OUT, f"deepfm_D{D}.npz"), **g)
    print("deepfm D=%d loss=%.6f" % (D, float(loss)))


def golden_fm(D, seed):
    """models/rank/fm/net.py:20-124 + fm/dygraph_model.py:53-58 (loss)."""
    import paddle  # the shim
    net = load_ref_module("models/rank/fm/net.py", "ref_fm_net")
    rng = np.random.default_rng(seed)
    N, S, Dn, B = 1001, 26, 13, 12
    torch.manual_seed(seed)
    model = net.FMLayer(N, D, Dn, S)
    with torch.no_grad():
        # the Constant(1.0) dense weights saturate the sigmoid on [0,1) dense inputs (y2 ~ 200: every gradient is
        # exactly 0) and, being all equal, would hide a swapped or dropped weight; Constant(0.0) bias likewise
        model.fm.dense_w_one.copy_(torch.as_tensor(0.2 * (1.0 + 0.5 * rng.standard_normal(Dn)).astype(np.float32)))
        model.fm.dense_w.copy_(torch.as_tensor(0.05 * (1.0 + 0.5 * rng.standard_normal((1, Dn, D))).astype(np.float32)))
        model.bias.copy_(torch.as_tensor(np.asarray([0.37], np.float32)))
    ids = make_ids(rng, B, S, N)
    dense = rng.random((B, Dn), dtype=np.float32)
    label = (rng.random((B, 1)) < 0.3).astype(np.int64)
    sparse_inputs = [paddle.to_tensor(ids[:, s:s + 1]) for s in range(S)]
    pred = model.forward(sparse_inputs, paddle.to_tensor(dense))
    cost = paddle.nn.functional.log_loss(input=pred, label=paddle.cast(paddle.to_tensor(label), dtype="float32"))
    loss = paddle.mean(x=cost)
    y1, y2 = model.fm.forward(sparse_inputs, paddle.to_tensor(dense))
    loss.backward()
    g = dict(ids=ids, dense=dense, label=label, D=np.int64(D),
             W=npy(model.fm.embedding.weight), W1=npy(model.fm.embedding_one.weight),
             dense_w=npy(model.fm.dense_w), dense_w_one=npy(model.fm.dense_w_one), bias=npy(model.bias),
             pred=npy(pred), loss=npy(loss), y1=npy(y1), y2=npy(y2),
             gW=npy(model.fm.embedding.weight.grad), gW1=npy(model.fm.embedding_one.weight.grad),
             g_dense_w=npy(model.fm.dense_w.grad), g_dense_w_one=npy(model.fm.dense_w_one.grad),
             g_bias=npy(model.bias.grad))
    np.savez_compressed(os.path.join(OUT, f"fm_D{D}.npz"), **g)
    print("fm D=%d loss=%.6f" % (D, float(loss.detach())))


def golden_wide_deep(D, seed):
    """models/rank/wide_deep/net.py:20-104 + wide_deep/dygraph_model.py:54-59 (loss)."""
    import paddle  # the shim
    net = load_ref_module("models/rank/wide_deep/net.py", "ref_wide_deep_net")
    rng = np.random.default_rng(seed)
    N, S, Dn, B, fc = 1001, 26, 13, 12, [32, 16]
    torch.manual_seed(seed)
    model = net.WideDeepLayer(N, D, Dn, S, fc)
    with torch.no_grad():       # Uniform(-1,1) rows x 26 slots would saturate a 2-layer toy MLP; biases away from 0
        model.embedding.weight.mul_(0.2)
        for m in [model.wide_part] + [l for l in model._mlp_layers if hasattr(l, "weight")]:
            m.bias.copy_(torch.as_tensor((0.05 * rng.standard_normal(tuple(m.bias.shape))).astype(np.float32)))
    ids = make_ids(rng, B, S, N)
    dense = rng.random((B, Dn), dtype=np.float32)
    label = (rng.random((B, 1)) < 0.3).astype(np.int64)
    sparse_inputs = [paddle.to_tensor(ids[:, s:s + 1]) for s in range(S)]
    pred = model.forward(sparse_inputs, paddle.to_tensor(dense))
    cost = paddle.nn.functional.log_loss(input=pred, label=paddle.cast(paddle.to_tensor(label), dtype="float32"))
    loss = paddle.mean(x=cost)
    loss.backward()
    g = dict(ids=ids, dense=dense, label=label, D=np.int64(D), W=npy(model.embedding.weight),
             wide_w=npy(model.wide_part.weight), wide_b=npy(model.wide_part.bias), pred=npy(pred), loss=npy(loss),
             gW=npy(model.embedding.weight.grad), g_wide_w=npy(model.wide_part.weight.grad),
             g_wide_b=npy(model.wide_part.bias.grad))
    lin = [m for m in model._mlp_layers if hasattr(m, "weight")]
    for i, l in enumerate(lin):
        g[f"mlp_w{i}"], g[f"mlp_b{i}"] = npy(l.weight), npy(l.bias)
        g[f"g_mlp_w{i}"], g[f"g_mlp_b{i}"] = npy(l.weight.grad), npy(l.bias.grad)
    g["n_mlp"] = np.int64(len(lin))
    np.savez_compressed(os.path.join(OUT, f"wide_deep_D{D}.npz"), **g)
    print("wide_deep D=%d loss=%.6f" % (D, float(loss.detach())))


def golden_dnn(D, seed):
    """models/rank/dnn/net.py:20-95 (DNNLayer) + dnn/dygraph_model.py:53-58 (softmax cross-entropy, mean)."""
    import paddle  # the shim
    net = load_ref_module("models/rank/dnn/net.py", "ref_dnn_net")
    rng = np.random.default_rng(seed)
    N, S, Dn, B, fc = 1001, 26, 13, 12, [32, 16]
    torch.manual_seed(seed)
    model = net.DNNLayer(N, D, Dn, S, fc)
    with torch.no_grad():       # as for wide_deep: Uniform(-1,1) rows saturate a toy MLP; biases away from 0
        model.embedding.weight.mul_(0.2)
        for m in [l for l in model._mlp_layers if hasattr(l, "weight")]:
            m.bias.copy_(torch.as_tensor((0.05 * rng.standard_normal(tuple(m.bias.shape))).astype(np.float32)))
    ids = make_ids(rng, B, S, N)
    dense = rng.random((B, Dn), dtype=np.float32)
    label = (rng.random((B, 1)) < 0.4).astype(np.int64)
    raw = model.forward([paddle.to_tensor(ids[:, s:s + 1]) for s in range(S)], paddle.to_tensor(dense))
    # paddle.nn.functional.cross_entropy(input, label [B,1] int64): softmax inside, hard labels, reduction 'mean' [EXT]
    loss = torch.nn.functional.cross_entropy(raw, torch.as_tensor(label).reshape(-1))
    loss.backward()
    g = dict(ids=ids, dense=dense, label=label, D=np.int64(D), W=npy(model.embedding.weight), raw=npy(raw),
             loss=npy(loss), gW=npy(model.embedding.weight.grad))
    lin = [m for m in model._mlp_layers if hasattr(m, "weight")]
    for i, l in enumerate(lin):
        g[f"mlp_w{i}"], g[f"mlp_b{i}"] = npy(l.weight), npy(l.bias)
        g[f"g_mlp_w{i}"], g[f"g_mlp_b{i}"] = npy(l.weight.grad), npy(l.bias.grad)
    g["n_mlp"] = np.int64(len(lin))
    np.savez_compressed(os.path.join(OUT, f"dnn_D{D}.npz"), **g)
    print("dnn D=%d loss=%.6f" % (D, float(loss.detach())))


def golden_dcn_v2(mix, seed):
    """models/rank/dcn_v2/net.py:20-320 in eval() mode (Dropout off — Appendix B-10)."""
    import paddle
    net = load_ref_module("models/rank/dcn_v2/net.py", "ref_dcn_net")
    rng = np.random.default_rng(seed)
    N, S, Dn, B, D, fc = 501, 26, 13, 6, 4, [24, 16]
    torch.manual_seed(seed)
    model = net.DCN_V2Layer(N, D, Dn, S, fc, cross_num=3 if not mix else 2, is_Stacked=True,
                            use_low_rank_mixture=mix, low_rank=8, num_experts=4)
    model.eval()
    # give the zero-initialised biases some signal so a bias bug cannot hide
    with torch.no_grad():
        for p_name, p in model.named_parameters():
            if "bias" in p_name:
                p.copy_(torch.from_numpy(rng.standard_normal(tuple(p.shape)).astype(np.float32) * 0.1))
    ids = make_ids(rng, B, S, N)
    dense = np.log(rng.random((B, Dn), dtype=np.float32) * 50 + 1).astype(np.float32)  # reader.py:63-64
    sparse_inputs = [paddle.to_tensor(ids[:, s:s + 1]) for s in range(S)]
    pred = model.forward(sparse_inputs, paddle.to_tensor(dense))
    emb = model.embedding(paddle.concat(sparse_inputs, axis=1))
    feat = paddle.concat([paddle.reshape(emb, [-1, S * D]), model.dense_emb(paddle.to_tensor(dense))], 1)
    cross = model.DeepCrossLayer_(feat)
    pred.sum().backward()
    g = dict(ids=ids, dense=dense, pred=npy(pred), feat=npy(feat), cross=npy(cross),
             mix=np.int64(mix))
    for k, v in model.state_dict().items():
        g["p." + k] = npy(v)
    for k, p in model.named_parameters():
        if p.grad is not None:
            g["g." + k] = npy(p.grad)
    np.savez_compressed(os.path.join(OUT, f"dcn_v2_{'mix' if mix else 'v2'}.npz"), **g)
    print("dcn_v2 mix=%d pred[0]=%.6f" % (mix, float(pred[0])))


def golden_din(seed):
    """models/rank/din/net.py:20-184 + mask construction of din/dinReader.py:81-99."""
    import paddle
    net = load_ref_module("models/rank/din/net.py", "ref_din_net")
    rng = np.random.default_rng(seed)
    B, T, item_count, cat_count, E = 5, 7, 301, 41, 8
    torch.manual_seed(seed)
    model = net.DINLayer(E, E, "sigmoid", False, True, item_count, cat_count)
    with torch.no_grad():
        model.item_b_attr.weight.copy_(torch.from_numpy(
            rng.standard_normal((item_count, 1)).astype(np.float32) * 0.1))
    lens = np.array([7, 3, 1, 5, 7])
    hist_item = np.zeros((B, T), np.int64)
    hist_cat = np.zeros((B, T), np.int64)
    mask = np.zeros((B, T, 1), np.float32)
    for b in range(B):
        hist_item[b, : lens[b]] = rng.integers(1, item_count, lens[b])
        hist_cat[b, : lens[b]] = rng.integers(1, cat_count, lens[b])
        mask[b, lens[b]:, 0] = -1e9                       # dinReader.py:81-84
    mask_i64 = mask.astype(np.int64)                      # dinReader.py:99  (cast to int64)
    target_item = rng.integers(1, item_count, B).astype(np.int64)
    target_cat = rng.integers(1, cat_count, B).astype(np.int64)
    target_item_seq = np.repeat(target_item[:, None], T, 1)
    target_cat_seq = np.repeat(target_cat[:, None], T, 1)
    label = (rng.random((B, 1)) < 0.5).astype(np.float32)
    tt = paddle.to_tensor
    logit = model.forward(tt(hist_item), tt(hist_cat), tt(target_item), tt(target_cat), tt(label),
                          tt(mask_i64), tt(target_item_seq), tt(target_cat_seq))
    loss = paddle.nn.functional.binary_cross_entropy_with_logits(logit, tt(label))  # dygraph_model.py:58-61
    loss.backward()
    g = dict(hist_item=hist_item, hist_cat=hist_cat, target_item=target_item, target_cat=target_cat,
             mask=mask_i64, lens=lens, label=label, logit=npy(logit), loss=npy(loss))
    for k, v in model.state_dict().items():
        g["p." + k] = npy(v)
    for i, l in enumerate([m for m in model.attention_layer if hasattr(m, "weight")]):
        g[f"att_w{i}"], g[f"att_b{i}"] = npy(l.weight), npy(l.bias)   # NOT in state_dict (App. B-9)
    for k, p in model.named_parameters():
        if p.grad is not None:
            g["g." + k] = npy(p.grad)
    g["registered"] = np.array(sorted(k for k, _ in model.named_parameters()))
    np.savez_compressed(os.path.join(OUT, "din.npz"), **g)
    print("din loss=%.6f" % float(loss))


def golden_xdeepfm(D, seed):
    """models/rank/xdeepfm/net.py:23-242 + xdeepfm/dygraph_model.py:53-58 (loss).  Two CIN layers so that the second
    one consumes a d-major X_k, a DNN of two hidden layers."""
    import paddle  # the shim
    net = load_ref_module("models/rank/xdeepfm/net.py", "ref_xdeepfm_net")
    rng = np.random.default_rng(seed)
    N, S, Dn, B, cin, dnn = 1001, 26, 13, 12, [8, 4], [32, 16]
    torch.manual_seed(seed)
    model = net.xDeepFMLayer(N, D, Dn, S, cin, dnn)
    with torch.no_grad():
        # Constant(1.0) dense weights are all equal (would hide a swapped weight) and the tiny initial CIN / DNN
        # weights make those branches vanish beside y_linear: rescale so every branch moves the logit
        model.fm.dense_w_one.copy_(torch.as_tensor(0.2 * (1.0 + 0.5 * rng.standard_normal(Dn)).astype(np.float32)))
        model.fm.dense_w.copy_(torch.as_tensor(0.5 * (1.0 + 0.5 * rng.standard_normal((1, Dn, D))).astype(np.float32)))
        model.fm.embedding.weight.mul_(12.0)
        model.bias.copy_(torch.as_tensor(np.asarray([0.21], np.float32)))
        model.cin.cin_linear.weight.mul_(16.0)
        model.cin.cin_linear.bias.copy_(torch.as_tensor(np.asarray([-0.13], np.float32)))
        for lin in model.dnn._mlp_layers:
            if hasattr(lin, "weight"):
                lin.weight.mul_(10.0)
                lin.bias.copy_(torch.as_tensor(0.1 * rng.standard_normal(lin.bias.shape[0]).astype(np.float32)))
    ids = make_ids(rng, B, S, N)
    dense = rng.random((B, Dn), dtype=np.float32)
    label = (rng.random((B, 1)) < 0.3).astype(np.int64)
    sparse_inputs = [paddle.to_tensor(ids[:, s:s + 1]) for s in range(S)]
    pred = model.forward(sparse_inputs, paddle.to_tensor(dense))
    cost = paddle.nn.functional.log_loss(input=pred, label=paddle.cast(paddle.to_tensor(label), dtype="float32"))
    loss = paddle.mean(x=cost)
    y_linear, feat = model.fm.forward(sparse_inputs, paddle.to_tensor(dense))
    y_cin, y_dnn = model.cin.forward(feat), model.dnn.forward(feat)
    loss.backward()
    lins = [l for l in model.dnn._mlp_layers if hasattr(l, "weight")]
    g = dict(ids=ids, dense=dense, label=label, D=np.int64(D),
             W=npy(model.fm.embedding.weight), W1=npy(model.fm.embedding_one.weight),
             dense_w=npy(model.fm.dense_w), dense_w_one=npy(model.fm.dense_w_one), bias=npy(model.bias),
             fc_w=npy(model.cin.cin_linear.weight), fc_b=npy(model.cin.cin_linear.bias),
             pred=npy(pred), loss=npy(loss), y1=npy(y_linear), feat=npy(feat), y_cin=npy(y_cin), y_dnn=npy(y_dnn),
             gW=npy(model.fm.embedding.weight.grad), gW1=npy(model.fm.embedding_one.weight.grad),
             g_dense_w=npy(model.fm.dense_w.grad), g_dense_w_one=npy(model.fm.dense_w_one.grad),
             g_bias=npy(model.bias.grad), g_fc_w=npy(model.cin.cin_linear.weight.grad),
             g_fc_b=npy(model.cin.cin_linear.bias.grad), n_cin=np.int64(len(cin)), n_mlp=np.int64(len(lins)),
             state_keys=np.array(sorted(model.state_dict().keys())))
    for i, conv in enumerate(model.cin.cnn_layers):
        g["cin_w%d" % i], g["g_cin_w%d" % i] = npy(conv.weight), npy(conv.weight.grad)
    for i, lin in enumerate(lins):
        g["mlp_w%d" % i], g["mlp_b%d" % i] = npy(lin.weight), npy(lin.bias)
        g["g_mlp_w%d" % i], g["g_mlp_b%d" % i] = npy(lin.weight.grad), npy(lin.bias.grad)
    np.savez_compressed(os.path.join(OUT, f"xdeepfm_D{D}.npz"), **g)
    print("xdeepfm D=%d loss=%.6f pred range %.3f..%.3f  y1 %.2f y_cin %.2f y_dnn %.2f (abs means)" % (
        D, float(loss.detach()), float(pred.detach().min()), float(pred.detach().max()), float(y_linear.detach().abs().mean()),
        float(y_cin.detach().abs().mean()), float(y_dnn.detach().abs().mean())))


def golden_dlrm(D, seed):
    """models/rank/dlrm/net.py:23-178 + dlrm/dygraph_model.py:53-57 (loss), train mode (BatchNorm on batch statistics)."""
    import paddle  # the shim
    net = load_ref_module("models/rank/dlrm/net.py", "ref_dlrm_net")
    rng = np.random.default_rng(seed)
    N, S, Dn, B, bot, top = 1001, 26, 13, 24, [32, D], [48, 2]
    torch.manual_seed(seed)
    model = net.DLRMLayer(dense_feature_dim=Dn, bot_layer_sizes=bot, sparse_feature_number=N, sparse_feature_dim=D,
                          top_layer_sizes=top, num_field=S, self_interaction=False)
    model.train()
    norms = [m for m in model.modules() if isinstance(m, paddle.nn.BatchNorm1D)]
    with torch.no_grad():
        model.embedding.weight.mul_(0.3)                       # TruncatedNormal() std 1: dots of 16-vectors would swamp x
        for bn in norms:                                       # non-trivial affine parameters and running statistics
            bn.weight.copy_(torch.as_tensor((1.0 + 0.3 * rng.standard_normal(bn.weight.shape[0])).astype(np.float32)))
            bn.bias.copy_(torch.as_tensor((0.2 * rng.standard_normal(bn.bias.shape[0])).astype(np.float32)))
            bn._mean.copy_(torch.as_tensor((0.1 * rng.standard_normal(bn.bias.shape[0])).astype(np.float32)))
            bn._variance.copy_(torch.as_tensor((1.0 + 0.2 * rng.random(bn.bias.shape[0])).astype(np.float32)))
    run0 = [(npy(bn._mean), npy(bn._variance)) for bn in norms]
    ids = make_ids(rng, B, S, N, pad_frac=0.0)
    ids[:3, :4] = 0                                            # id 0 is an ordinary row here (no padding_idx)
    dense = rng.random((B, Dn), dtype=np.float32)
    label = (rng.random((B, 1)) < 0.4).astype(np.int64)
    sparse_inputs = [paddle.to_tensor(ids[:, s:s + 1]) for s in range(S)]
    raw = model.forward(sparse_inputs, paddle.to_tensor(dense))
    cost = paddle.nn.functional.cross_entropy(input=raw, label=paddle.to_tensor(label))
    loss = paddle.mean(x=cost)
    loss.backward()
    g = dict(ids=ids, dense=dense, label=label, D=np.int64(D), W=npy(model.embedding.weight),
             gW=npy(model.embedding.weight.grad), raw=npy(raw), loss=npy(loss),
             state_keys=np.array(sorted(model.state_dict().keys())))
    for name, mlp in (("bot", model.bot_mlp), ("top", model.top_mlp)):
        lins = [l for l in mlp.mlp if isinstance(l, paddle.nn.Linear)]
        bns = [l for l in mlp.mlp if isinstance(l, paddle.nn.BatchNorm1D)]
        assert len(lins) == len(bns)                           # net.py:145: every layer carries ReLU + BatchNorm
        g["n_" + name] = np.int64(len(lins))
        for i, (lin, bn) in enumerate(zip(lins, bns)):
            k = "%s%d_" % (name, i)
            j = norms.index(bn)
            g[k + "w"], g[k + "b"], g[k + "gamma"], g[k + "beta"] = npy(lin.weight), npy(lin.bias), npy(bn.weight), npy(bn.bias)
            g[k + "mean0"], g[k + "var0"] = run0[j]
            g[k + "mean1"], g[k + "var1"] = npy(bn._mean), npy(bn._variance)      # after this training forward
            g[k + "gw"], g[k + "gb"] = npy(lin.weight.grad), npy(lin.bias.grad)
            g[k + "ggamma"], g[k + "gbeta"] = npy(bn.weight.grad), npy(bn.bias.grad)
    model.eval()
    with torch.no_grad():
        g["raw_eval"] = npy(model.forward(sparse_inputs, paddle.to_tensor(dense)))
    np.savez_compressed(os.path.join(OUT, f"dlrm_D{D}.npz"), **g)
    print("dlrm D=%d loss=%.6f raw range %.3f..%.3f" % (D, float(loss.detach()), float(raw.detach().min()),
                                                       float(raw.detach().max())))


def golden_slot_dnn(D, seed):
    """models/rank/slot_dnn/net.py:21-85 (BenchmarkDNNLayer) + static_model.py:104-108 (loss): multi-value slots over
    ONE shared table, sum-pooled per slot, concat, MLP, sigmoid(clip(+-15))."""
    import paddle  # the shim
    from paddle.static import nn as snn
    net = load_ref_module("models/rank/slot_dnn/net.py", "ref_slot_dnn_net")
    rng = np.random.default_rng(seed)
    N, S, B, fc = 503, 12, 9, [32, 16]
    torch.manual_seed(seed)
    snn.PARAMS.clear()
    model = net.BenchmarkDNNLayer(N, D, S, fc)
    # per (sample, slot): 0..4 ids, an absent slot holds the single padding id 0 (queuedataset_reader.py:75-80);
    # duplicates inside a segment and across samples are forced (SelectedRows merge path)
    samples = []
    for b in range(B):
        row = []
        for s in range(S):
            k = int(rng.integers(0, 5))
            v = [int(x) for x in rng.integers(1, N, size=k)] if k else [0]
            if k >= 2 and rng.random() < 0.5:
                v[1] = v[0]
            if b % 2 == 1 and s < S // 2:
                v = list(samples[b - 1][s])
            if len(v) >= 3 and rng.random() < 0.3:
                v[2] = 0                                  # an explicit padding id inside a longer segment
            row.append(v)
        samples.append(row)
    label = (rng.random((B, 1)) < 0.4).astype(np.int64)
    slot_inputs = []
    for s in range(S):
        vals, lod = [], [0]
        for b in range(B):
            vals.extend(samples[b][s])
            lod.append(len(vals))
        slot_inputs.append(paddle.LoDTensor(paddle.to_tensor(np.asarray(vals, np.int64).reshape(-1, 1)), lod, str(s + 2)))
    show = paddle.LoDTensor(paddle.to_tensor(np.ones((B, 1), np.int64)), list(range(B + 1)), "show")
    click = paddle.LoDTensor(paddle.to_tensor(label), list(range(B + 1)), "click")
    # the table parameter is created by the first sparse_embedding call: run once, then set seeded weights
    model.forward(show, click, slot_inputs)
    W = snn.PARAMS["embedding"]
    with torch.no_grad():
        W.copy_(torch.as_tensor((rng.standard_normal((N, D)) * 0.3).astype(np.float32)))
        lin = [m for m in model._mlp_layers if hasattr(m, "weight")]
        lin[-1].bias.fill_(0.1)
        for l in lin:                     # spread the logits so that some samples sit outside the +-15 clip
            l.weight.mul_(24.0)
    pred = model.forward(show, click, slot_inputs)
    cost = paddle.nn.functional.log_loss(input=pred, label=paddle.cast(paddle.to_tensor(label), "float32"))
    loss = paddle.mean(x=cost)
    loss.backward()
    pooled = model.all_vars[S]           # concat of the S bows (net.py:77-78)
    logits = model.all_vars[-2]
    g = dict(D=np.int64(D), N=np.int64(N), S=np.int64(S), label=label,
             samples=np.array([[",".join(str(x) for x in v) for v in r] for r in samples]),
             W=npy(W), gW=npy(W.grad), pred=npy(pred), loss=npy(loss), pooled=npy(pooled), logits=npy(logits),
             n_clipped=np.int64(int((npy(logits).__abs__() >= 15).sum())))
    for i, l in enumerate(lin):
        g[f"mlp_w{i}"], g[f"mlp_b{i}"] = npy(l.weight), npy(l.bias)
        g[f"g_mlp_w{i}"], g[f"g_mlp_b{i}"] = npy(l.weight.grad), npy(l.bias.grad)
    g["n_mlp"] = np.int64(len(lin))
    np.savez_compressed(os.path.join(OUT, f"slot_dnn_D{D}.npz"), **g)
    print("slot_dnn D=%d loss=%.6f clipped=%d" % (D, float(loss), int(g["n_clipped"])))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    jobs = {"deepfm_D9": lambda: golden_deepfm(9, 20250404), "deepfm_D16": lambda: golden_deepfm(16, 20250405),
            "dcn_v2_v2": lambda: golden_dcn_v2(False, 20250406), "dcn_v2_mix": lambda: golden_dcn_v2(True, 20250407),
            "din": lambda: golden_din(20250408), "fm_D9": lambda: golden_fm(9, 20250409),
            "wide_deep_D9": lambda: golden_wide_deep(9, 20250410), "dnn_D9": lambda: golden_dnn(9, 20250411),
            "slot_dnn_D9": lambda: golden_slot_dnn(9, 20250412), "xdeepfm_D9": lambda: golden_xdeepfm(9, 20250413),
            "dlrm_D16": lambda: golden_dlrm(16, 20250414)}
    for name in (sys.argv[1:] or list(jobs)):      # optional: only the named fixtures
        jobs[name]()
