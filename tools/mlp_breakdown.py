#!/usr/bin/env python3
"""Per-call timing of the DeepFM top-MLP forward/backward on rec_gemm_f32 (B=65536, 624-400-400-400-1)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from paddlerec_amd import ops  # noqa: E402

DEV = "cuda"
B = 65536
sizes = [624, 400, 400, 400, 1]
ws = ops.Workspace(DEV)
W = [torch.randn(sizes[i], sizes[i + 1], device=DEV) * 0.05 for i in range(4)]
b = [torch.zeros(sizes[i + 1], device=DEV) for i in range(4)]
dW = [torch.empty_like(w) for w in W]
db = [torch.empty_like(x) for x in b]
x = torch.randn(B, 624, device=DEV)
dz = torch.randn(B, 1, device=DEV) * 1e-4


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, c in ev:
        a.record(); fn(); c.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(c) for a, c in ev)
    return ts[len(ts) // 2]


y, acts = ops.mlp_forward(x, W, b, ws)
print("mlp_forward total %.3f ms" % timeit(lambda: ops.mlp_forward(x, W, b, ws)))
print("mlp_backward total %.3f ms" % timeit(lambda: ops.mlp_backward(dz, acts, W, dW, db, ws)))
for i in range(4):
    print("fwd L%d %.3f ms" % (i, timeit(lambda: ops.gemm(acts[i], W[i], ws, epilogue="bias_relu" if i < 3 else "bias", bias=b[i]))))
g = dz
for i in reversed(range(4)):
    t1 = timeit(lambda: ops.gemm(acts[i], g, ws, trans_a=True, out=dW[i]))
    t2 = timeit(lambda: ops.colsum(g, ws, out=db[i]))
    if i > 0:
        t3 = timeit(lambda: ops.gemm(g, W[i], ws, trans_b=True, epilogue="relu_mask", aux0=acts[i]))
        g = ops.gemm(g, W[i], ws, trans_b=True, epilogue="relu_mask", aux0=acts[i])
    else:
        t3 = timeit(lambda: ops.gemm(g, W[i], ws, trans_b=True))
    print("bwd L%d: dW %.3f  colsum %.3f  dX %.3f ms" % (i, t1, t2, t3))
