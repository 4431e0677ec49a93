#!/usr/bin/env python3
"""One rec_gemm_f32 shape in a loop (for rocprofv3): python tools/gemm_loop.py fwd0|dx0|dw0|cross [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from paddlerec_amd import ops  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "fwd0"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
DEV, B = "cuda", 65536
cfg = {"fwd0": (B, 400, 624, False, False), "dx0": (B, 624, 400, False, True),
       "dw0": (624, 400, B, True, False), "cross": (B, 1560, 1560, False, False)}[which]
M, N, K, ta, tb = cfg
A = torch.randn((K, M) if ta else (M, K), device=DEV)
Bm = torch.randn((N, K) if tb else (K, N), device=DEV)
out = torch.empty(M, N, device=DEV)
ws = ops.Workspace(DEV)
for _ in range(iters):
    ops.gemm(A, Bm, ws, trans_a=ta, trans_b=tb, out=out)
torch.cuda.synchronize()
print("done")
