#!/usr/bin/env python3
"""Every GEMM of the slot_dnn MLP (3672-512-256-128-128-128-1, batch 65536) on its own: forward (bias + ReLU), dX (ReLU
mask), dW (+ bias gradient) — time and TFLOP/s, to see which layers of the chain are far from the MFMA rate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paddlerec_amd import ops
DEV = "cuda"
B = 65536
sizes = [3672, 512, 256, 128, 128, 128, 1]
g = torch.Generator(device=DEV).manual_seed(1)
ws = ops.Workspace(DEV)


def timeit(fn, R=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(R):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / R


tot = {"fwd": 0.0, "dX": 0.0, "dW": 0.0}
for i in range(len(sizes) - 1):
    K, N = sizes[i], sizes[i + 1]
    x = torch.relu(torch.randn(B, K, device=DEV, generator=g))
    w = torch.randn(K, N, device=DEV, generator=g) * 0.05
    bias = torch.zeros(N, device=DEV)
    y = torch.empty(B, N, device=DEV)
    dy = torch.randn(B, N, device=DEV, generator=g)
    dx = torch.empty(B, K, device=DEV)
    dw = torch.empty(K, N, device=DEV)
    db = torch.empty(N, device=DEV)
    fl = 2.0 * B * K * N
    t_f = timeit(lambda: ops.gemm(x, w, ws, epilogue="bias_relu" if i < len(sizes) - 2 else "bias", bias=bias, out=y))
    t_x = timeit(lambda: ops.gemm(dy, w, ws, trans_b=True, epilogue="relu_mask", aux0=x, out=dx))
    t_w = timeit(lambda: ops.gemm(x, dy, ws, trans_a=True, out=dw, b_colsum=db))
    tot["fwd"] += t_f; tot["dX"] += t_x; tot["dW"] += t_w
    print("layer %d  %5d -> %4d   fwd %7.1f us %6.1f TF   dX %7.1f us %6.1f TF   dW %7.1f us %6.1f TF   (HBM floor fwd %5.1f us)"
          % (i, K, N, t_f * 1e3, fl / t_f / 1e9, t_x * 1e3, fl / t_x / 1e9, t_w * 1e3, fl / t_w / 1e9,
             (B * K + B * N) * 4 / 5.0e12 * 1e6))
print("sum: fwd %.2f ms  dX %.2f ms  dW %.2f ms" % (tot["fwd"], tot["dX"], tot["dW"]))
