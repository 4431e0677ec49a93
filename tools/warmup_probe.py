#!/usr/bin/env python3
"""Does a fresh process / fresh box need more than a few warm-up steps?  Times consecutive blocks of 10 DeepFM train
steps (bench.py's workload) from the very first step."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from paddlerec_amd.deepfm import DeepFMLayer  # noqa: E402

dev = torch.device("cuda", 0)
B, S, Dn, D = 65536, 26, 13, 16
so = torch.arange(S, dtype=torch.int64, device=dev) * 1_000_000
torch.manual_seed(20250404)
model = DeepFMLayer(26_000_000, D, Dn, S, [400, 400, 400], device=dev, slot_offset=so)
batches = bench.make_batches(4, B, S, Dn, 1_000_000, dev, 20250404, "uniform")
torch.cuda.synchronize()
t_start = time.perf_counter()
for blk in range(12):
    t0 = time.perf_counter()
    for i in range(10):
        ids, dense, label = batches[i % 4]
        model.train_step(ids, dense, label, lr=1e-3)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print("steps %3d-%3d: %.3f ms/step   (%.2f s since the first step)" % (blk * 10, blk * 10 + 9, 1e2 * (t1 - t0),
                                                                           t1 - t_start), flush=True)
