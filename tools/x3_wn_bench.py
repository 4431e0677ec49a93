#!/usr/bin/env python3
"""Round 6: the bf16 x 3 forward / dX GEMM in the shape the environment selects (REC_X3_WN=1: four-wave workgroups, two
per CU; 2: the round-5 eight-wave workgroup): time per launch (20 launches between one HIP-event pair, 3 rounds, median)
over rotating operands, and the error against float64 on a sample of rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paddlerec_amd import ops
g = torch.Generator(device="cuda").manual_seed(1)
rnd = lambda *s: torch.rand(*s, device="cuda", generator=g) - 0.5
ws = ops.Workspace("cuda")
cases = [(65536, 400, 400, False, "bias_relu"), (65536, 400, 400, True, "relu_mask"), (65536, 400, 432, False, "bias_relu"),
         (65536, 432, 400, True, "none"), (65536, 512, 3680, False, "bias_relu"), (65536, 1560, 1560, False, "bias_relu")]
for M, N, K, tb, epi in cases:
    As = [rnd(M, K) for _ in range(3)]
    B, bias, X0, C = (rnd(N, K) if tb else rnd(K, N)), rnd(N), rnd(M, N), torch.empty(M, N, device="cuda")
    img = ops.GemmImages([(B, tb)], "cuda"); img.refresh()
    run = lambda i, im=None: ops.gemm(As[i % 3], B, ws, trans_b=tb, epilogue=epi, bias=bias if epi.startswith("bias") else None,
                                      aux0=X0 if epi == "relu_mask" else None, out=C, b_image=im)
    def timed(im):
        ts = []
        for _ in range(3):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(20): run(i, im)
            b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3 / 20)
        return sorted(ts)[1]
    t_img, t_self = timed(img.get(0)), timed(None)
    run(0, img.get(0)); torch.cuda.synchronize(); rows = torch.arange(0, M, 997, device="cuda")
    ref = As[0][rows].double() @ (B.double().t() if tb else B.double())
    if epi.startswith("bias"): ref = torch.relu(ref + bias.double())
    if epi == "relu_mask": ref = torch.where(X0[rows] > 0, ref, torch.zeros_like(ref))
    scale = (As[0][rows].abs().double() @ (B.abs().double().t() if tb else B.abs().double())).max()
    err = float((C[rows].double() - ref).abs().max() / scale)
    print("M %6d N %5d K %5d tb %d %-10s  %7.1f us with image (%5.1f TF-eq)  %7.1f us self-split   err/sum|a||b| %.2e"
          % (M, N, K, tb, epi, t_img, 2.0 * M * N * K / t_img / 1e6, t_self, err), flush=True)
