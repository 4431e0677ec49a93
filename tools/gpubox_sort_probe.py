#!/usr/bin/env python3
"""The gpubox model's merge-key sort alone: 40.35 M (row, segment) pairs, 44 % of them the padding row, rows in a
10 000 019-row table (tools/slot_dnn_bench.py's batch) — rec_ids_group_payload, and the accessor push behind it."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from paddlerec_amd import ops  # noqa: E402

DEV = "cuda"
n, N = 40354718, 10000019
g = torch.Generator(device=DEV).manual_seed(1)
rows = torch.randint(1, N, (n,), device=DEV, generator=g, dtype=torch.int64)
rows[torch.rand(n, device=DEV, generator=g) < (1 - 22706897 / n)] = 0
seg = torch.randint(0, 65536 * 408, (n,), device=DEV, generator=g, dtype=torch.int32)
ws, status, groups = ops.Workspace(DEV), ops.new_status(DEV), ops.IdGroups(int(n * 1.25) + 1, DEV)


def run():
    ops.ids_group(rows, N, 0, ws, None, status, groups, payload=seg)


for _ in range(3):
    run()
torch.cuda.synchronize()
for rep in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        run()
    b.record()
    torch.cuda.synchronize()
    print("rec_ids_group_payload alone: %.3f ms   (n_uniq %d, live %d)" % (a.elapsed_time(b) / 5, int(groups.n_uniq[0]), int(groups.n_uniq[1])))
