"""paddle.optimizer of the compat namespace: weight_decay / ParamAttr(regularizer=L2Decay) and
grad_clip=ClipGradByGlobalNorm are APPLIED (they used to be accepted and dropped, so dcn_v2 / xdeepfm
configs trained with silently different semantics).  Runs in a subprocess (the namespace patches torch.Tensor); operator
backend = the oracle-backed stand-in; expected values = plain torch autograd + a hand-written Adam with Paddle's
epsilon placement, gradients clipped and regularised in Paddle's order (clip over the raw dense
gradients and the MERGED SelectedRows rows first, then the regulariser appended)."""
import os
import subprocess
import sys

from conftest import REPO

SCRIPT = r"""
import os, sys
import numpy as np, torch
import paddle, paddle.nn as nn

paddle.seed(3); paddle.set_device("cpu")
N, D, S, B = 50, 8, 3, 16
class Net(nn.Layer):
    def __init__(self):
        super().__init__()
        self.emb = nn.Embedding(N, D, padding_idx=0, sparse=True)
        self.l0 = nn.Linear(S * D, 12, weight_attr=paddle.ParamAttr(regularizer=paddle.regularizer.L2Decay(0.3)))
        self.l1 = nn.Linear(12, 1)
    def forward(self, ids):
        x = paddle.reshape(self.emb(ids), [-1, S * D])
        return nn.functional.sigmoid(self.l1(nn.functional.relu(self.l0(x))))

CLIP, WD, LR = 0.05, 0.1, 0.01
net = Net()
opt = paddle.optimizer.Adam(learning_rate=LR, parameters=net.parameters(), weight_decay=WD,
                            grad_clip=nn.ClipGradByGlobalNorm(CLIP))
ref = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
mom = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in ref.items()}
reg = {"l0.weight": 0.3, "l0.bias": WD, "l1.weight": WD, "l1.bias": WD, "emb.weight": 0.0}   # own regulariser wins
rng = np.random.default_rng(0)
clipped = 0
for step in range(1, 4):
    ids = torch.as_tensor(rng.integers(0, N, (B, S)))
    ids[0, 0] = ids[1, 1] = 7                       # duplicates: the clip norm is over MERGED rows
    y = torch.as_tensor((rng.random((B, 1)) < 0.4).astype(np.float32))
    loss = nn.functional.log_loss(net(ids), y).mean() * 30.0       # large gradients: the clip is active
    loss.backward(); opt.step(); opt.clear_grad()
    e = torch.nn.functional.embedding(ids, ref["emb.weight"], padding_idx=0).reshape(B, S * D)
    p = torch.sigmoid(torch.relu(e @ ref["l0.weight"] + ref["l0.bias"]) @ ref["l1.weight"] + ref["l1.bias"])
    rl = (-(y * torch.log(p + 1e-4) + (1 - y) * torch.log(1 - p + 1e-4))).mean() * 30.0
    gs = dict(zip(ref, torch.autograd.grad(rl, list(ref.values()))))
    norm = torch.sqrt(sum((g.double() ** 2).sum() for g in gs.values())).float()
    scale = CLIP / max(float(norm), CLIP)
    clipped += scale < 1
    with torch.no_grad():
        for k, w in ref.items():
            g = gs[k] * scale + reg[k] * w
            m, v = mom[k]
            m.mul_(0.9).add_(g, alpha=0.1); v.mul_(0.999).addcmul_(g, g, value=0.001)
            lr_t = LR * (1 - 0.999 ** step) ** 0.5 / (1 - 0.9 ** step)
            w -= lr_t * m / (v.sqrt() + 1e-8 * (1 - 0.999 ** step) ** 0.5)
assert clipped == 3, clipped
for k, w in net.state_dict().items():
    np.testing.assert_allclose(w.detach().numpy(), ref[k].detach().numpy(), rtol=2e-5, atol=2e-7, err_msg=k)
try:
    paddle.optimizer.Adam(parameters=net.parameters(), grad_clip=object())
    raise SystemExit("an unknown grad_clip must be refused")
except NotImplementedError:
    pass
print("COMPAT_OPT_OK")
"""


def test_weight_decay_regularizer_and_global_norm_clip_are_applied():
    env = dict(os.environ, REC_COMPAT_KERNELS="cpu_kernels", OMP_NUM_THREADS="2")
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(REPO, "paddlerec_amd", "compat"), os.path.join(REPO, "tests"), REPO])
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "COMPAT_OPT_OK" in r.stdout, (r.stdout + r.stderr)[-3000:]
