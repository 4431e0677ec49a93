#!/usr/bin/env python3
"""DeepFM's layer-0 weight gradient (432 x 400 x 65536, with its bias-gradient column sums) alone: the 144x80 tile
against the 80x80 tile (REC_GEMM_144=0), K split 16 / 32 / planner's."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from paddlerec_amd import ops  # noqa: E402
DEV = "cuda"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


ws = ops.Workspace(DEV)
B = 65536
for M in (432, 400):
    X, G = torch.randn(B, M, device=DEV), torch.randn(B, 400, device=DEV)
    dW, db = torch.empty(M, 400, device=DEV), torch.empty(400, device=DEV)
    for split in (0, 16, 32):
        t = timeit(lambda: ops.gemm(X, G, ws, trans_a=True, out=dW, b_colsum=db, split_k=split))
        print("dW %d x 400 x %d split %2d: %.1f us  %.1f TF  (REC_GEMM_144=%s)" %
              (M, B, split, t * 1e3, 2.0 * M * 400 * B / t / 1e9, os.environ.get("REC_GEMM_144", "1")))
