#!/usr/bin/env python3
"""rec_sparse_sgd_small on the three id patterns of a DIN step at batch 32 (T 152): unique-ish ids (history items),
~6 duplicates per row (history categories), 152 occurrences per row (the target item of a sample per position), and the
32-lookup tables; against rec_ids_group + rec_segment_partials + rec_sparse_sgd_rows on the same lookups."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paddlerec_amd import ops
DEV = "cuda"
g = torch.Generator(device=DEV).manual_seed(1)
B, T, E = 32, 152, 128


def timeit(fn, R=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(R):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / R * 1e3


cases = {
    "history items  (4864 ids over 63001 rows)": (torch.randint(0, 63001, (B * T,), device=DEV, generator=g), 63001, 64),
    "history cats   (4864 ids over 801 rows)": (torch.randint(0, 801, (B * T,), device=DEV, generator=g), 801, 64),
    "target per pos (32 rows x 152 occurrences)": (torch.randint(0, 63001, (B, 1), device=DEV, generator=g).expand(B, T).reshape(-1).contiguous(), 63001, 64),
    "target item    (32 ids)": (torch.randint(0, 63001, (B,), device=DEV, generator=g), 63001, 64),
}
ws = ops.Workspace(DEV)
st = ops.new_status(DEV)
for name, (ids, N, D) in cases.items():
    n = ids.numel()
    grad = torch.randn(n, E, device=DEV, generator=g)
    P = torch.randn(N, D, device=DEV, generator=g)
    t_small = timeit(lambda: ops.sparse_sgd_small(ids, grad[:, :D], P, 0.1, None, st, grad_group=1, grad_group_stride=E))
    grp = ops.IdGroups(n, DEV)

    def sorted_path():
        ops.ids_group(ids, N, None, ws, None, st, grp)
        pp = ops.segment_partials(grp, grad[:, :D], D, grad_group=1, grad_group_stride=E)
        ops.sparse_sgd_rows(grp, grad[:, :D], P, 0.1, grad_group=1, grad_group_stride=E, partials=pp)
    t_sort = timeit(sorted_path)
    print("%-46s one launch %6.1f us   sort + partials + rows %6.1f us" % (name, t_small, t_sort))
