#!/usr/bin/env python3
"""rec_sparse_adam_record_small alone (the one-launch merge + lazy Adam of a DeepFM record) at the reference's batch size:
512 samples x 26 slots x 1 M rows per slot, D 16 — and at other sizes, to see what the launch's time is made of."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paddlerec_amd import ops
DEV = "cuda"
g = torch.Generator(device=DEV).manual_seed(1)
S, D, R = 26, 16, 1000000
rec = torch.zeros(S * R, 32, device=DEV)
mv = torch.zeros(S * R, 32, device=DEV)
so = torch.arange(S, device=DEV, dtype=torch.int64) * R
st = ops.new_status(DEV)


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for B in (64, 128, 256, 512):
    ids = torch.randint(0, R, (B, S), device=DEV, generator=g)
    grad = torch.randn(B * S, D, device=DEV, generator=g)
    dz = torch.randn(B, 1, device=DEV, generator=g)
    t = timeit(lambda: ops.sparse_adam_record_small(ids, so, 0, grad, dz, S, rec, mv, D, 1, 1e-3, v_offset=16, status=st))
    print("B %4d  (%5d lookups, %4d blocks of 16 waves)  %6.1f us" % (B, B * S, (B * S + 15) // 16, t))
