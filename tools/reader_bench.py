#!/usr/bin/env python3
"""Host parser throughput (rows R/H): engine C++ parser vs the per-line Python restatement of the reference reader."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from paddlerec_amd import reader as rd
from test_reader import _synthetic_slot_lines, _synthetic_tsv
from oracle import deepfm_ref as R
lines = _synthetic_slot_lines(20000, 1)
data = ("\n".join(lines * 20) + "\n").encode()
print("host cores:", os.cpu_count())
for th in (1, 4, 16, 0):
    t = time.perf_counter(); l, i, d = rd.parse_slot_text(data, threads=th); dt = time.perf_counter() - t
    print("slot text  threads=%-2d: %.2f M lines/s (%.0f MB/s)" % (th, l.shape[0] / dt / 1e6, len(data) / dt / 1e6))
tl = _synthetic_tsv(20000, 2)
data = ("\n".join(tl * 20) + "\n").encode()
for th in (1, 16, 0):
    t = time.perf_counter(); l, i, d = rd.parse_criteo_tsv(data, threads=th); dt = time.perf_counter() - t
    print("criteo tsv threads=%-2d: %.2f M lines/s (26 xxh32 hashes per line)" % (th, l.shape[0] / dt / 1e6))
t = time.perf_counter()
for ln in lines[:5000]:
    R.parse_slot_line(ln)
print("python per-line restatement of criteo_reader.py: %.3f M lines/s" % (5000 / (time.perf_counter() - t) / 1e6))
