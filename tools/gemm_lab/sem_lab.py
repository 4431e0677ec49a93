#!/usr/bin/env python3
"""The arrival-counter primitive of a one-launch split-K, alone (tools/gemm_lab/sem_lab.hip): correctness over many launches
with data that changes every launch, and the time per launch of  2 = two launches,  0 = fences,  1 = L2-bypassing accesses.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared tools/gemm_lab/sem_lab.hip -o tools/gemm_lab/_build/libsemlab.so
    python tools/gemm_lab/sem_lab.py [--launches 2000]"""
import argparse
import ctypes as C
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--launches", type=int, default=2000)
    ap.add_argument("--lib", default=os.path.join(HERE, "_build", "libsemlab.so"))
    args = ap.parse_args()
    lab = C.CDLL(args.lib)
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: C.c_void_p(t.data_ptr())
    for tiles, nblk, length in ((40, 5, 64 * 80), (25, 8, 80 * 80), (160, 3, 64 * 80), (8, 16, 64 * 80)):
        src = torch.randint(-8, 8, (nblk, tiles, length), device="cuda").float()      # small integers: exact sums
        part = torch.zeros(nblk, tiles, length, device="cuda")
        out = torch.zeros(tiles, length, device="cuda")
        line = "tiles %3d x %2d slices x %5d floats:" % (tiles, nblk, length)
        for mode in (2, 0, 1):
            bad = 0
            for r in range(args.launches):
                rc = lab.lab_sem(mode, tiles, nblk, length, p(src), p(part), p(out), r % 97, C.c_void_p(st))
                assert rc == 0
                if r % 50 == 49 or r < 5:           # (checking costs a sync: sampled, plus the first launches)
                    want = src.sum(0) + float(nblk * (r % 97))
                    bad += int((out != want).sum().item())
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for r in range(200):
                lab.lab_sem(mode, tiles, nblk, length, p(src), p(part), p(out), r % 97, C.c_void_p(st))
            b.record()
            torch.cuda.synchronize()
            line += "   mode %d: %6.2f us / launch, %d wrong values" % (mode, a.elapsed_time(b) * 1e3 / 200, bad)
        print(line, flush=True)


if __name__ == "__main__":
    main()
