#!/usr/bin/env python3
"""How much of the forward GEMM's loss is the half-empty last round of blocks?  65536 x 400 x 400 is 1280 blocks of
256 x 80 on 512 resident slots = 2.5 rounds; M is varied so that the block count lands just below / above whole rounds."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from paddlerec_amd import ops  # noqa: E402

DEV = "cuda"
g = torch.Generator(device=DEV).manual_seed(1)
w = torch.randn(400, 400, device=DEV, generator=g) * 0.05
b = torch.zeros(400, device=DEV)
ws = ops.Workspace(DEV)
for tiles in (102, 103, 153, 154, 204, 205, 256, 306, 307, 408, 409, 512):
    M = tiles * 256
    x = torch.randn(M, 400, device=DEV, generator=g)
    out = torch.empty(M, 400, device=DEV)
    fn = lambda: ops.gemm(x, w, ws, epilogue="bias_relu", bias=b, out=out)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        fn()
    e.record()
    torch.cuda.synchronize()
    t = a.elapsed_time(e) / 20
    print("M %6d  blocks %4d = %.2f rounds of 512   %.1f us   %.1f TF" % (M, tiles * 5, tiles * 5 / 512, 1e3 * t,
                                                                          2.0 * M * 400 * 400 / t / 1e9), flush=True)
