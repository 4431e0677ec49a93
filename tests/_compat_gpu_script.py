"""Runs in a subprocess (the compat namespace patches torch.Tensor methods): a small CTR net written against the
`paddle` compat API — Embedding(sparse=True, padding_idx=0) + Linear stack + log_loss, Adam — on cuda:0 with the HIP
kernels, against the same net in plain torch fp32 (autograd, dense Adam with Paddle's epsilon placement)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "paddlerec_amd", "compat"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import paddle  # noqa: E402  (the compat namespace)
import paddle.nn as nn  # noqa: E402

paddle.seed(7)
paddle.set_device("gpu")
N, D, S, B = 500, 16, 6, 64


class Net(nn.Layer):
    def __init__(self):
        super().__init__()
        self.emb = nn.Embedding(N, D, padding_idx=0, sparse=True,
                                weight_attr=paddle.ParamAttr(initializer=nn.initializer.TruncatedNormal(0.0, 0.1)))
        self.l0 = nn.Linear(S * D, 32)
        self.l1 = nn.Linear(32, 1)

    def forward(self, ids):
        e = self.emb(ids)                                   # [B, S, D]
        x = paddle.reshape(e, [-1, S * D])
        return nn.functional.sigmoid(self.l1(nn.functional.relu(self.l0(x))))


net = Net()
opt = paddle.optimizer.Adam(learning_rate=0.01, parameters=net.parameters())
ref = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
mom = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in ref.items()}
rng = np.random.default_rng(3)
worst = 0.0
for step in range(1, 4):
    ids = rng.integers(0, 40, (B, S)).astype(np.int64)       # small id range: duplicates inside the batch; 0 = padding
    lab = (rng.random((B, 1)) < 0.3).astype(np.float32)
    tid, tl = paddle.to_tensor(ids), paddle.to_tensor(lab)
    opt.clear_grad()
    pred = net.forward(tid)
    loss = paddle.mean(nn.functional.log_loss(pred, tl))
    loss.backward()
    opt.step()
    # torch reference of the same step
    for v in ref.values():
        v.grad = None
    e = torch.nn.functional.embedding(tid, ref["emb.weight"]) * (tid != 0).unsqueeze(-1)
    h = torch.relu(e.reshape(B, -1) @ ref["l0.weight"] + ref["l0.bias"])
    p = torch.sigmoid(h @ ref["l1.weight"] + ref["l1.bias"])
    rl = (-tl * torch.log(p + 1e-4) - (1 - tl) * torch.log(1 - p + 1e-4)).mean()
    rl.backward()
    np.testing.assert_allclose(float(loss), float(rl), rtol=1e-5)
    with torch.no_grad():
        b1, b2, eps, lr = 0.9, 0.999, 1e-8, 0.01
        lr_t = lr * (1 - b2 ** step) ** 0.5 / (1 - b1 ** step)
        for k, v in ref.items():
            g = v.grad if v.grad is not None else torch.zeros_like(v)
            m, vv = mom[k]
            m.mul_(b1).add_(g, alpha=1 - b1)
            vv.mul_(b2).addcmul_(g, g, value=1 - b2)
            v.sub_(lr_t * m / (vv.sqrt() + eps * (1 - b2 ** step) ** 0.5))     # Paddle's epsilon placement [EXT]
    for k, v in net.state_dict().items():
        d = float((v - ref[k]).abs().max())
        worst = max(worst, d)
        assert d < 2e-6, (step, k, d)
print("compat gpu ok: 3 steps, worst |param diff| %.2e, final loss %.6f" % (worst, float(loss)))
