#!/usr/bin/env python3
"""rec_ids_group_slots alone, N calls in a row (run under rocprofv3 --kernel-trace --stats for per-kernel times)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from paddlerec_amd import ops  # noqa: E402
DEV = "cuda"
g = torch.Generator(device=DEV).manual_seed(1)
B, S, NT = 65536, 26, 1_000_000
ids = torch.randint(1, NT, (B, S), device=DEV, generator=g)
ids[torch.rand(B, S, device=DEV, generator=g) < 0.03] = 0
ws, st, grp = ops.Workspace(DEV), ops.new_status(DEV), ops.IdGroups(B * S, DEV)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for _ in range(3):
    ops.ids_group_slots(ids, NT, 0, ws, st, grp, want_rank=True)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(n):
    ops.ids_group_slots(ids, NT, 0, ws, st, grp, want_rank=True)
b.record()
torch.cuda.synchronize()
print("ids_group_slots: %.1f us per call (dbg %s)" % (a.elapsed_time(b) * 1e3 / n, os.environ.get("REC_SG_DBG", "0")))
