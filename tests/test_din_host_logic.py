"""DIN host mirror (paddlerec_amd/din.py) with the oracle-backed operator stand-in on the CPU: forward, the explicit
backward chain through the four Linear layers and the attention-pool, the seven embedding tables' merged SGD rows and
the dense SGD — against the golden fixture of the reference's din/net.py and the NumPy oracle (no kernel involved;
tests/test_din_gpu.py runs the same checks on the HIP kernels)."""
import numpy as np
import torch

import cpu_kernels
from helpers import load_golden
from oracle import din_ref as Dn

T = lambda a: torch.as_tensor(np.ascontiguousarray(a))
N_ = lambda t: t.detach().numpy()


def _golden_model():
    from paddlerec_amd.din import DINLayer
    g = load_golden("din")
    p = {k[2:]: v for k, v in g.items() if k.startswith("p.")}
    m = DINLayer(8, 8, "sigmoid", False, True, 301, 41, device="cpu", kernels=cpu_kernels)
    m.set_dict(p)
    m.set_attention([g["att_w%d" % i] for i in range(3)], [g["att_b%d" % i] for i in range(3)])
    B, Tn = g["hist_item"].shape
    tis = np.repeat(g["target_item"][:, None], Tn, 1)
    tcs = np.repeat(g["target_cat"][:, None], Tn, 1)
    feeds = [T(g[k]) for k in ("hist_item", "hist_cat", "target_item", "target_cat", "label", "mask")] + [T(tis), T(tcs)]
    return g, p, m, feeds


def test_forward_golden():
    g, p, m, feeds = _golden_model()
    np.testing.assert_allclose(N_(m.forward(*feeds)), g["logit"], rtol=1e-5, atol=1e-6)


def test_train_step_golden_grads_and_sgd():
    g, p, m, feeds = _golden_model()
    lr = 0.85
    loss, pred = m.train_step(*feeds, base_lr=lr)
    np.testing.assert_allclose(N_(loss)[0], g["loss"], rtol=1e-5)
    for name in ("linear_0", "linear_1", "linear_2", "linearCon"):
        for part in ("weight", "bias"):
            k = "%s.%s" % (name, part)
            np.testing.assert_allclose(N_(m._last["dense"][k]).reshape(g["g." + k].shape), g["g." + k], rtol=3e-4,
                                       atol=3e-7, err_msg=k)
    sd = m.state_dict()
    n = 0
    for k, v in g.items():
        if k.startswith("g."):                   # every registered parameter moved by -lr * the reference's gradient
            np.testing.assert_allclose(N_(sd[k[2:]]), p[k[2:]] - lr * v, rtol=2e-4, atol=2e-6, err_msg=k)
            n += 1
    assert n >= 15


def test_train_steps_vs_oracle():
    from paddlerec_amd.din import DINLayer
    rng = np.random.default_rng(9)
    ni, nc, B, Tn = 120, 30, 24, 20
    m = DINLayer(8, 8, "sigmoid", False, True, ni, nc, device="cpu", kernels=cpu_kernels)
    with torch.no_grad():
        m.params["item_b_attr.weight"].copy_(T((rng.standard_normal((ni, 1)) * 0.1).astype(np.float32)))
    p = {k: N_(v).copy() for k, v in m.state_dict().items()}
    att = ([N_(w).copy() for w in m.attention_w], [N_(b).copy() for b in m.attention_b])
    lr = 0.5
    for _ in range(2):
        lens = rng.integers(1, Tn + 1, B)
        hi = np.where(np.arange(Tn)[None] < lens[:, None], rng.integers(1, ni, (B, Tn)), 0)
        hc = np.where(np.arange(Tn)[None] < lens[:, None], rng.integers(1, nc, (B, Tn)), 0)
        ti, tc = rng.integers(1, ni, B), rng.integers(1, nc, B)
        mask = np.where(np.arange(Tn)[None] < lens[:, None], 0, -1000000000).astype(np.int64).reshape(B, Tn, 1)
        label = (rng.random((B, 1)) < 0.4).astype(np.float32)
        tis, tcs = np.repeat(ti[:, None], Tn, 1), np.repeat(tc[:, None], Tn, 1)
        loss, pred = m.train_step(T(hi), T(hc), T(ti), T(tc), T(label), T(mask), T(tis), T(tcs), base_lr=lr)
        grads = Dn.backward(p, att, hi, hc, ti, tc, mask, label)
        np.testing.assert_allclose(N_(loss)[0], Dn.bce_with_logits_mean(grads["_logit"], label), rtol=2e-5)
        for k in p:
            p[k] = (p[k] - lr * np.asarray(grads[k]).reshape(p[k].shape)).astype(np.float32)
    for k, v in m.state_dict().items():
        np.testing.assert_allclose(N_(v), p[k], rtol=2e-4, atol=3e-6, err_msg=k)


def test_train_step_graphed_on_the_cpu_backend_is_the_eager_step():
    """paddlerec_amd/graph.py host logic: with host tensors (the operator stand-in) StepGraph runs the step eagerly on
    one shared buffer set; the step counter and the piecewise learning rate advance per call either way."""
    g, p, m, feeds = _golden_model()
    _, _, m2, _ = _golden_model()
    for _ in range(3):
        la, _ = m.train_step(*feeds, base_lr=0.85)
        lb, _ = m2.train_step_graphed(*feeds, base_lr=0.85)
        assert torch.equal(la, lb)
    assert m.step_count == m2.step_count == 3
    assert (m2._graph.eager, m2._graph.captures, m2._graph.replays) == (3, 0, 0)
    for k, v in m.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k]), k
    assert m.learning_rate(409999, 0.85) == 0.85 and m.learning_rate(410000, 0.85) == 0.2


def test_step_graph_signature_and_eviction_logic():
    """StepGraph bookkeeping without a GPU: signatures are (shape, dtype, stride) per tensor + the keyword constants."""
    from paddlerec_amd.graph import StepGraph
    a, b = torch.zeros(3, 4), torch.zeros(3, 5)
    s1 = StepGraph._signature((a, b), {"lr": 0.1})
    assert s1 == StepGraph._signature((torch.ones(3, 4), torch.ones(3, 5)), {"lr": 0.1})
    assert s1 != StepGraph._signature((a, b), {"lr": 0.2})
    assert s1 != StepGraph._signature((a.t().contiguous().t(), b), {"lr": 0.1})        # same shape, other strides
    calls = []
    sg = StepGraph(lambda st, x, lr: calls.append((st, lr)) or x + lr, state_factory=lambda: "bufs")
    out = sg(a, lr=1.0)
    assert torch.equal(out, a + 1.0) and calls == [("bufs", 1.0)] and sg.eager == 1
