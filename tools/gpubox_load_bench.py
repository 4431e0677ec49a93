#!/usr/bin/env python3
"""load_into_memory of the gpubox pass loop (paddlerec_amd/gpubox.py InMemoryReader) on a synthetic pass in the
slot_dnn line format (`feasign:slot` tokens, ~450 per line, ~10 KB of text per line): whole-file C parse + batches cut
by index arithmetic, against the line-list loader it replaced (split lines in python, re-join per batch, parse per batch).

    python tools/gpubox_load_bench.py [--lines 40000 --batch 4096 --files 4]"""
import argparse, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from paddlerec_amd import gpubox, reader as rd


def old_loader(files, B, S):
    pending, out = [], []
    for path in files:
        with open(path, "rb") as f:
            pending += [ln for ln in f.read().split(b"\n") if ln.strip()]
        while len(pending) >= B:
            chunk, pending = pending[:B], pending[B:]
            data = b"\n".join(chunk) + b"\n"
            values, lod, base, n = rd.parse_feasign_slots(data, 2, S, 0, 0)
            lv, llod, _, _ = rd.parse_feasign_slots(data, 1, 1, 0, 0)
            out.append((values, lod, base, lv[llod[0, :-1]].reshape(n, 1).clamp_(0, 1)))
    return out


ap = argparse.ArgumentParser()
ap.add_argument("--lines", type=int, default=40000)
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--files", type=int, default=4)
ap.add_argument("--slots", type=int, default=408)
args = ap.parse_args()
rng = np.random.default_rng(3)
with tempfile.TemporaryDirectory() as d:
    files, per = [], args.lines // args.files
    block = []
    for i in range(256):                              # 256 distinct lines, repeated
        slots = np.sort(rng.choice(np.arange(2, args.slots + 2), size=140, replace=False))
        toks = ["%d:1" % (i & 1)]
        for s in slots:
            for _ in range(int(rng.integers(1, 6))):
                toks.append("%d:%d" % (int(rng.integers(1, 2 ** 63)), s))
        block.append(" ".join(toks))
    text = ("\n".join(block) + "\n").encode()
    for i in range(args.files):
        p = os.path.join(d, "part-%02d" % i)
        with open(p, "wb") as f:
            for _ in range(per // 256):
                f.write(text)
        files.append(p)
    size = sum(os.path.getsize(p) for p in files)
    n_lines = (per // 256) * 256 * args.files
    t0 = time.perf_counter(); r = gpubox.InMemoryReader(files, args.batch, args.slots); nb = r.load_into_memory()
    t_new = time.perf_counter() - t0
    if os.environ.get("REC_PROFILE"):
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable(); gpubox.InMemoryReader(files, args.batch, args.slots).load_into_memory(); pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(10)
    t0 = time.perf_counter(); old = old_loader(files, args.batch, args.slots); t_old = time.perf_counter() - t0
    assert nb == len(old) and all((a[0] == b[0]).all() and (a[1] == b[1]).all() for a, b in zip(r.batches[:2], old[:2]))
    print("pass: %d lines, %.2f GB of text, batch %d -> %d batches" % (n_lines, size / 1e9, args.batch, nb))
    print("whole-file parse + index cuts: %.2f s  = %.0f k lines/s, %.2f GB/s" % (t_new, n_lines / t_new / 1e3, size / t_new / 1e9))
    print("line-list loader (replaced):   %.2f s  = %.0f k lines/s, %.2f GB/s" % (t_old, n_lines / t_old / 1e3, size / t_old / 1e9))
