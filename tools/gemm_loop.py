#!/usr/bin/env python3
"""Three launches each of the bench's forward GEMM (65536 x 400 x 400, bias + ReLU), its dX (ReLU mask) and its dW, for
a rocprofv3 --pmc pass (tools/pmc.sh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from paddlerec_amd import ops  # noqa: E402

DEV = "cuda"
g = torch.Generator(device=DEV).manual_seed(3)
B = 65536
x = torch.randn(B, 400, device=DEV, generator=g)
w = torch.randn(400, 400, device=DEV, generator=g) * 0.05
b = torch.zeros(400, device=DEV)
gg = torch.randn(B, 400, device=DEV, generator=g)
dw, db = torch.empty_like(w), torch.empty_like(b)
ws = ops.Workspace(DEV)
for _ in range(3):
    ops.gemm(x, w, ws, epilogue="bias_relu", bias=b)
    ops.gemm(gg, w, ws, trans_b=True, epilogue="relu_mask", aux0=x)
    ops.gemm(x, gg, ws, trans_a=True, out=dw, b_colsum=db)
torch.cuda.synchronize()
print("done")
