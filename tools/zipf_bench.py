#!/usr/bin/env python3
"""Zipf-distributed ids (SURVEY.md §8(d) distribution Z): duplicates concentrate on a few hot rows — the stress
case for the SelectedRows merge / lazy optimizer.  Reports ids_group, sparse_adam_rows and a full train step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from paddlerec_amd import ops
from paddlerec_amd.deepfm import DeepFMLayer
DEV = "cuda"
B, S, Dn, D, NT = 65536, 26, 13, 16, 1_000_000


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


rng = np.random.default_rng(20250404)
so = torch.arange(S, dtype=torch.int64, device=DEV) * NT
m = DeepFMLayer(NT * S, D, Dn, S, [400, 400, 400], device=DEV, slot_offset=so)
for name, alpha in (("uniform", None), ("zipf a=1.05", 1.05), ("zipf a=1.5", 1.5)):
    if alpha is None:
        ids = rng.integers(1, NT, (B, S), dtype=np.int64)
    else:
        perm = rng.permutation(NT)
        ids = np.clip(perm[np.minimum(rng.zipf(alpha, size=(B, S)), NT - 1)], 1, NT - 1).astype(np.int64)
    ids[rng.random((B, S)) < 0.03] = 0
    t_ids = torch.as_tensor(ids).to(DEV)
    dense = torch.rand(B, Dn, device=DEV)
    label = (torch.rand(B, 1, device=DEV) < 0.25).long()
    ws = ops.Workspace(DEV)
    groups, _ = ops.ids_group(t_ids, NT * S, 0, ws, so)
    U, nv = groups.n_uniq.tolist()[:2]
    seg = groups.seg_offset[: U + 1].long()
    longest = int((seg[1:] - seg[:-1]).max())
    rg = torch.randn(B * S, D, device=DEV) * 1e-3
    m._ensure_sparse_state()
    st = m.sparse_state
    t_adam0 = timeit(lambda: ops.sparse_adam_rows(groups, rg, 1, m.fm.embedding, st["m"], st["v"], 1)) \
        if longest < 5000 or os.environ.get("ZIPF_PLAIN") else float("nan")
    pp = ops.segment_partials(groups, rg, D)
    t_pp = timeit(lambda: ops.segment_partials(groups, rg, D, out=pp))
    t_adam = timeit(lambda: ops.sparse_adam_rows(groups, rg, 1, m.fm.embedding, st["m"], st["v"], 1, partials=pp))
    t_grp = timeit(lambda: ops.ids_group(t_ids, NT * S, 0, ws, so, None, groups))
    t_step = timeit(lambda: m.train_step(t_ids, dense, label, lr=1e-3))
    print("%-12s unique rows %8d  longest segment %7d | ids_group %.3f ms  segment_partials %.3f ms  sparse_adam %.3f ms "
          "(position by position: %.3f ms)  train step %.3f ms" % (name, U, longest, t_grp, t_pp, t_adam, t_adam0, t_step))
