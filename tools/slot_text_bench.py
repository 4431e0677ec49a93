#!/usr/bin/env python3
"""Where an epoch of the L-trainer run goes on the host: reading a slot-text file, parsing it (all threads / 1 thread),
pinned staging, host->device copies.  python tools/slot_text_bench.py [--lines 500000]"""
import argparse, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paddlerec_amd import reader
from tools.trainer_bench import write_slot_text

ap = argparse.ArgumentParser()
ap.add_argument("--lines", type=int, default=500000)
args = ap.parse_args()
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "part-00")
    write_slot_text(path, args.lines, 1000001, 1)
    size = os.path.getsize(path)
    t0 = time.perf_counter(); data = open(path, "rb").read(); t_read = time.perf_counter() - t0
    for thr in (0, 64, 16, 1):
        best = 1e9
        for _ in range(3 if thr != 1 else 1):
            t0 = time.perf_counter()
            label, ids, dense = reader.parse_slot_text(data, 26, 13, False, thr, pinned=True)
            best = min(best, time.perf_counter() - t0)
        print("parse_slot_text threads=%-3d: %6.1f ms  %5.2f M lines/s  %5.2f GB/s of text" %
              (thr, best * 1e3, args.lines / best / 1e6, size / best / 1e9))
    print("file read (page cache): %.1f ms for %.0f MB = %.2f GB/s" % (t_read * 1e3, size / 1e6, size / t_read / 1e9))
    if torch.cuda.is_available():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = [t.to("cuda", non_blocking=True) for t in (label, ids, dense)]
        torch.cuda.synchronize()
        t_h2d = time.perf_counter() - t0
        nb = sum(t.numel() * t.element_size() for t in (label, ids, dense))
        print("pinned -> device: %.1f ms for %.0f MB = %.1f GB/s" % (t_h2d * 1e3, nb / 1e6, nb / t_h2d / 1e9))
