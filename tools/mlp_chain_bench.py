#!/usr/bin/env python3
"""The DeepFM MLP chain of bench.py (feat' [B,432] -> 400 -> 400 -> 400 -> 1, forward + backward) on its own: the same
ops.mlp_forward / ops.mlp_backward calls as a train step, nothing else on the GPU.  Per-GEMM HIP-event times, to tell
what a GEMM costs inside the step (profiles/*_bench_kernel_stats.csv) from what it costs alone."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from paddlerec_amd import ops  # noqa: E402

DEV = "cuda"
B = 65536
sizes = [432, 400, 400, 400, 1]
g = torch.Generator(device=DEV).manual_seed(1)
W = [torch.randn(sizes[i], sizes[i + 1], device=DEV, generator=g) * 0.05 for i in range(4)]
Bb = [torch.zeros(sizes[i + 1], device=DEV) for i in range(4)]
dW = [torch.empty_like(w) for w in W]
dB = [torch.empty_like(b) for b in Bb]
x = torch.randn(B, 432, device=DEV, generator=g)
dz = torch.randn(B, 1, device=DEV, generator=g) * 1e-3
ws = ops.Workspace(DEV)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


y, acts = ops.mlp_forward(x, W, Bb, ws)
t_f = timed(lambda: ops.mlp_forward(x, W, Bb, ws))
t_b = timed(lambda: ops.mlp_backward(dz, acts, W, dW, dB, ws))
fl_f = 2.0 * B * (432 * 400 + 400 * 400 * 2 + 400)
print("MLP chain alone, B %d: forward %.3f ms (%.1f TF)   backward %.3f ms (%.1f TF)   fwd+bwd %.3f ms (%.1f TF)"
      % (B, t_f, fl_f / t_f / 1e9, t_b, 2 * fl_f / t_b / 1e9, t_f + t_b, 3 * fl_f / (t_f + t_b) / 1e9))
for i in range(3):
    a_in = acts[i]
    t = timed(lambda: ops.gemm(a_in, W[i], ws, epilogue="bias_relu", bias=Bb[i]))
    fl = 2.0 * B * sizes[i] * sizes[i + 1]
    print("  fwd layer %d (K %d): %.1f us  %.1f TF" % (i, sizes[i], 1e3 * t, fl / t / 1e9))
gg = torch.randn(B, 400, device=DEV, generator=g)
for i in (2, 1):
    t = timed(lambda: ops.gemm(gg, W[i], ws, trans_b=True, epilogue="relu_mask", aux0=acts[i]))
    print("  dX layer %d with ReLU mask: %.1f us  %.1f TF" % (i, 1e3 * t, 2.0 * B * 400 * 400 / t / 1e9))
    t = timed(lambda: ops.gemm(gg, W[i], ws, trans_b=True))
    print("  dX layer %d plain:          %.1f us  %.1f TF" % (i, 1e3 * t, 2.0 * B * 400 * 400 / t / 1e9))
    t = timed(lambda: ops.gemm(acts[i], gg, ws, trans_a=True, out=dW[i], b_colsum=dB[i]))
    print("  dW layer %d (+ bias grad):  %.1f us  %.1f TF" % (i, 1e3 * t, 2.0 * B * 400 * 400 / t / 1e9))
t = timed(lambda: ops.gemm(gg, W[0], ws, trans_b=True))
print("  dX layer 0: %.1f us  %.1f TF" % (1e3 * t, 2.0 * B * 432 * 400 / t / 1e9))
t = timed(lambda: ops.gemm(acts[0], gg, ws, trans_a=True, out=dW[0], b_colsum=dB[0]))
print("  dW layer 0: %.1f us  %.1f TF" % (1e3 * t, 2.0 * B * 432 * 400 / t / 1e9))
