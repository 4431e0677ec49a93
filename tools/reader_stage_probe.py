#!/usr/bin/env python3
"""Where the L-trainer's reader spends a file: mmap + parse (pinned / pageable outputs, first and repeated call),
host -> device copies of its batches.  One 3.28 M-line slot-text file (1.8 GB), as tools/trainer_bench.py writes them."""
import mmap
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from paddlerec_amd import reader  # noqa: E402
from tools.trainer_bench import write_slot_text  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3276800
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "part-0")
    write_slot_text(path, n, 1000000, 5)
    size = os.path.getsize(path)
    for pinned in (True, True, True, False, False):
        t0 = time.perf_counter()
        with open(path, "rb") as f:
            mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
            t1 = time.perf_counter()
            label, ids, dense = reader.parse_slot_text(mm, 26, 13, False, 0, pinned=pinned)
            t2 = time.perf_counter()
            reader._close_mmap(mm)
        print("pinned=%s: mmap %.1f ms, parse %.1f ms (%.1f M lines/s, %.1f GB/s of text)" %
              (pinned, 1e3 * (t1 - t0), 1e3 * (t2 - t1), n / (t2 - t1) / 1e6, size / (t2 - t1) / 1e9), flush=True)
        if torch.cuda.is_available():
            B = 65536
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            outs = []
            for lo in range(0, n - B + 1, B):
                outs.append(tuple(t[lo:lo + B].to("cuda", non_blocking=True) for t in (label.view(-1, 1), ids, dense)))
            torch.cuda.synchronize()
            print("   %d batches host -> device: %.1f ms (%.1f GB/s)" % (len(outs), 1e3 * (time.perf_counter() - t3),
                  (label.nbytes + ids.nbytes + dense.nbytes) / (time.perf_counter() - t3) / 1e9), flush=True)
            del outs
        del label, ids, dense
