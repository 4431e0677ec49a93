#!/usr/bin/env python3
"""Host time of one engine call through the ctypes adapter (launch-bound steps pay this per kernel): ops.gemm on a
32 x 80 x 40 problem, 5000 calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paddlerec_amd import ops
x = torch.randn(32, 80, device="cuda"); w = torch.randn(80, 40, device="cuda"); b = torch.zeros(40, device="cuda")
out = torch.empty(32, 40, device="cuda"); ws = ops.Workspace("cuda")
for _ in range(100):
    ops.gemm(x, w, ws, epilogue="bias_sigmoid", bias=b, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5000):
    ops.gemm(x, w, ws, epilogue="bias_sigmoid", bias=b, out=out)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("ops.gemm host time per call: %.2f us" % ((t1 - t0) / 5000 * 1e6))
