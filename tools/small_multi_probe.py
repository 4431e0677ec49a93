#!/usr/bin/env python3
"""Where the time of DIN's seven-table one-launch merge goes (rec_sparse_sgd_small_multi at the shipped batch: 32 samples x 152
history positions): the launch over all seven jobs and over subsets of them, with the ids of din_small_bench.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from paddlerec_amd import ops  # noqa: E402

DEV = "cuda"
B, T, Ei, Ec = 32, 152, 64, 64
E = Ei + Ec


def timeit(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n)
    return best * 1e3


def main():
    g = torch.Generator(device=DEV).manual_seed(1)
    hi = torch.randint(0, 63001, (B, T), device=DEV, generator=g)
    hc = torch.randint(0, 801, (B, T), device=DEV, generator=g)
    ti = torch.randint(0, 63001, (B, 1), device=DEV, generator=g)
    tc = torch.randint(0, 801, (B, 1), device=DEV, generator=g)
    tis, tcs = ti.expand(B, T).contiguous(), tc.expand(B, T).contiguous()
    rnd = lambda *s: torch.rand(*s, device=DEV, generator=g) * 1e-3
    dq, dh, de0, dz = rnd(B, T, E), rnd(B, T, E), rnd(B, 2 * E), rnd(B, 1)
    tabs = {k: rnd(n, d) for k, (n, d) in dict(tis=(63001, Ei), tcs=(801, Ec), hi=(63001, Ei), hc=(801, Ec), ti=(63001, Ei),
                                                tc=(801, Ec), b=(63001, 1)).items()}
    jobs = {"target_item_seq": (tis.reshape(-1), dq, tabs["tis"], 1, E),
            "target_cat_seq": (tcs.reshape(-1), dq[:, :, Ei:], tabs["tcs"], 1, E),
            "hist_item_seq": (hi.reshape(-1), dh, tabs["hi"], 1, E),
            "hist_cat_seq": (hc.reshape(-1), dh[:, :, Ei:], tabs["hc"], 1, E),
            "target_item": (ti.reshape(-1), de0[:, E:], tabs["ti"], 1, 2 * E),
            "target_cat": (tc.reshape(-1), de0[:, E + Ei:], tabs["tc"], 1, 2 * E),
            "item_b": (ti.reshape(-1), dz, tabs["b"], 1, 1)}
    st = ops.new_status(DEV)
    sets = [("all seven", list(jobs)), ("target_*_seq (one row x 152 per sample)", ["target_item_seq", "target_cat_seq"]),
            ("target_item_seq", ["target_item_seq"]), ("hist_item_seq", ["hist_item_seq"]), ("hist_cat_seq", ["hist_cat_seq"]),
            ("hist_*", ["hist_item_seq", "hist_cat_seq"]), ("the three 32-lookup tables", ["target_item", "target_cat", "item_b"])]
    print("REC_SMALL_FP=%s" % os.environ.get("REC_SMALL_FP", "1"))
    for name, keys in sets:
        js = [jobs[k] for k in keys]
        t = timeit(lambda: ops.sparse_sgd_small_multi(js, 1e-3, st))
        print("  %-45s %7.1f us" % (name, t), flush=True)
    assert int(st.item()) == 0


if __name__ == "__main__":
    main()
