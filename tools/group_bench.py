#!/usr/bin/env python3
"""rec_ids_group (hand-written radix grouping, csrc/radix_sort.h) timed standalone: the 1.7 M lookups of a Criteo batch
(uniform and Zipf ids), the 40 M ids of a slot_dnn batch, and rec_shard_route for 8 owners."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from paddlerec_amd import ops  # noqa: E402

DEV = "cuda"


def timeit(fn, R=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(R):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / R


g = torch.Generator(device=DEV).manual_seed(1)
B, S, NT = 65536, 26, 1_000_000
so = torch.arange(S, device=DEV, dtype=torch.int64) * NT
ids = torch.randint(1, NT, (B, S), device=DEV, generator=g)
ids[torch.rand(B, S, device=DEV, generator=g) < 0.03] = 0
ws = ops.Workspace(DEV)
st = ops.new_status(DEV)
grp = ops.IdGroups(B * S, DEV)
print("ids_group uniform 26 x 1M rows, n = %d: %.1f us" % (B * S, timeit(lambda: ops.ids_group(ids, NT * S, 0, ws, so, st, grp))))
print("ids_group_slots (slot-local, + rank):        %.1f us" % timeit(lambda: ops.ids_group_slots(ids, NT, 0, ws, st, grp, want_rank=True)))
rng = np.random.default_rng(1)
z = torch.as_tensor(np.minimum(rng.zipf(1.05, size=(B, S)), NT - 1)).to(DEV)
print("ids_group zipf 1.05:                         %.1f us  (long segment flag %d)" %
      (timeit(lambda: ops.ids_group(z, NT * S, 0, ws, so, st, grp)), int(grp.n_uniq[2].item())))
print("ids_group_slots zipf 1.05:                   %.1f us" % timeit(lambda: ops.ids_group_slots(z, NT, 0, ws, st, grp, want_rank=True)))
big = torch.randint(1, 10_000_019, (40_000_000,), device=DEV, generator=g)
grp2 = ops.IdGroups(big.numel(), DEV)
print("ids_group slot_dnn 40 M ids, 10 M rows:      %.1f us" % timeit(lambda: ops.ids_group(big, 10_000_019, 0, ws, None, st, grp2), R=5))
wide = torch.randint(1, 10 ** 10, (B, S), device=DEV, generator=g)
print("ids_group 64-bit keys (10^10 rows):          %.1f us" % timeit(lambda: ops.ids_group(wide, 10 ** 10, 0, ws, None, st, grp)))
route = ops.ShardRoute(B * S, 8, DEV)
print("shard_route 8 owners:                        %.1f us" % timeit(lambda: ops.shard_route(ids, NT * S, 0, 8, ws, so, st, route)))
