#!/usr/bin/env python3
"""Runs ONE GEMM shape a few times (for rocprofv3 --pmc / --kernel-trace passes): gemm_one.py M N K [tb] [epi] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from paddlerec_amd import ops  # noqa: E402

M, N, K = (int(x) for x in sys.argv[1:4])
tb = len(sys.argv) > 4 and sys.argv[4] == "1"
epi = sys.argv[5] if len(sys.argv) > 5 else "bias_relu"
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 6
g = torch.Generator(device="cuda").manual_seed(1)
rnd = lambda *s: torch.rand(*s, device="cuda", generator=g) - 0.5
A, B, bias, X0, C = rnd(M, K), (rnd(N, K) if tb else rnd(K, N)), rnd(N), rnd(M, N), torch.empty(M, N, device="cuda")
ws = ops.Workspace("cuda")
for _ in range(reps):
    ops.gemm(A, B, ws, trans_b=tb, epilogue=epi, bias=bias if epi.startswith("bias") else None,
             aux0=X0 if epi == "relu_mask" else None, out=C)
torch.cuda.synchronize()
