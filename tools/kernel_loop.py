#!/usr/bin/env python3
"""Runs ONE engine kernel in a loop on BASELINE config-2 shapes (for rocprofv3 --pmc / --kernel-trace).
    python tools/kernel_loop.py fm_fwd|fm_bwd|sparse_adam|ids_group [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from paddlerec_amd import ops  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "fm_fwd"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
DEV = "cuda"
B, S, Dn, D, NT = 65536, 26, 13, 16, 1_000_000
g = torch.Generator(device=DEV).manual_seed(1)
N = NT * S
W = torch.randn(N, D, device=DEV, generator=g) * 0.02
W1 = torch.randn(N, 1, device=DEV, generator=g) * 0.02
dw = torch.randn(1, Dn, D, device=DEV, generator=g) * 0.02
dw1 = torch.randn(Dn, device=DEV, generator=g) * 0.02
batches = []
for _ in range(4):
    ids = torch.randint(1, NT, (B, S), device=DEV, generator=g)
    ids[torch.rand(B, S, device=DEV, generator=g) < 0.03] = 0
    batches.append(ids)
dense = torch.rand(B, Dn, device=DEV, generator=g)
so = torch.arange(S, device=DEV, dtype=torch.int64) * NT
y1, y2, feat, sum_emb, status = ops.deepfm_fm_fwd(batches[0], dense, W, W1, dw, dw1, 0, so)
dfeat = torch.randn(B, S + Dn, D, device=DEV, generator=g) * 1e-3
dz = torch.randn(B, 1, device=DEV, generator=g) * 1e-3
ws = ops.Workspace(DEV)
o = ops.deepfm_fm_bwd(dense, feat, sum_emb, dfeat, dz, dz, S, ws, dense_w=dw)
groups, _ = ops.ids_group(batches[0], N, 0, ws, so)
M, V = torch.zeros_like(W), torch.zeros_like(W)
torch.cuda.synchronize()
for i in range(iters):
    ids = batches[i % 4]
    if which == "fm_fwd":
        ops.deepfm_fm_fwd(ids, dense, W, W1, dw, dw1, 0, so, status, (y1, y2, feat, sum_emb))
    elif which == "fm_bwd":
        ops.deepfm_fm_bwd(dense, feat, sum_emb, dfeat, dz, dz, S, ws, o, dense_w=dw)
    elif which == "sparse_adam":
        ops.sparse_adam_rows(groups, o[0], 1, W, M, V, i + 1)
    elif which == "ids_group":
        ops.ids_group(ids, N, 0, ws, so, status, groups)
torch.cuda.synchronize()
print("done", which, iters)
