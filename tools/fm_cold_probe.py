#!/usr/bin/env python3
"""What makes fm_fwd_kernel cost ~90 us inside a step and ~70 us in back-to-back repeats?  (r03)
Candidates: (1) the repeats rewrite the SAME 113 MB feat buffer, which then lives in the 256 MB Infinity Cache and
never reaches HBM; (2) TLB reach over the 3.3 GB table.  Times fm_fwd (median of 5 x R launches per event pair):

    same-out     : R back-to-back launches, 4 id batches cycled, ONE output set        (bench.py r02 'repeat')
    rot-out      : the same with 6 output sets cycled (6 x 113 MB > Infinity Cache)
    cold         : each launch preceded by a 1 GB device copy (event pair around the launch only)
    small table  : 26 x 100k rows (333 MB: no TLB pressure), same-out / rot-out / cold
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from paddlerec_amd import ops  # noqa: E402

DEV = "cuda"
B, S, Dn, D = 65536, 26, 13, 16


def run(NT):
    g = torch.Generator(device=DEV).manual_seed(1)
    N = NT * S
    rec = torch.zeros(N, 32, device=DEV)
    rec[:, :17].normal_(0, 0.02, generator=g)
    W, W1 = rec[:, :D], rec[:, D:D + 1]
    dw = torch.randn(1, Dn, D, device=DEV, generator=g) * 0.02
    dw1 = torch.randn(Dn, device=DEV, generator=g) * 0.02
    batches = []
    for _ in range(4):
        ids = torch.randint(1, NT, (B, S), device=DEV, generator=g)
        ids[torch.rand(B, S, device=DEV, generator=g) < 0.03] = 0
        batches.append(ids)
    dense = torch.rand(B, Dn, device=DEV, generator=g)
    so = torch.arange(S, device=DEV, dtype=torch.int64) * NT
    status = ops.new_status(DEV)
    outs = []
    for _ in range(6):
        o = ops.deepfm_fm_fwd(batches[0], dense, W, W1, dw, dw1, 0, so, status, compact=True)
        outs.append(tuple(t.clone() for t in o[:4]))
    big_a = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=DEV)
    big_b = torch.empty_like(big_a)

    def fwd(i, rot):
        ops.deepfm_fm_fwd(batches[i % 4], dense, W, W1, dw, dw1, 0, so, status, outs[i % 6 if rot else 0], compact=True)

    def b2b(rot, R=24, reps=5):
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(R):
                fwd(i, rot)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3 / R)
        return sorted(ts)[len(ts) // 2]

    def cold(reps=9, pre=None):
        ts = []
        for i in range(reps):
            (pre or (lambda: big_b.copy_(big_a)))()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fwd(i, True)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        return sorted(ts)[len(ts) // 2]

    print("table 26 x %8d rows (%5.2f GB): same-out %6.1f us   rot-out %6.1f us   cold(after 1 GB copy) %6.1f us"
          % (NT, N * 128 / 1e9, b2b(False), b2b(True), cold()), flush=True)
    # which half of the copy hurts: a READ-only sweep leaves clean lines in L2 / Infinity Cache, a WRITE-only fill
    # leaves dirty ones that the next kernel's allocations have to push out to HBM
    print("     after a 1 GB read-only sweep (sum) %6.1f us   after a 1 GB write-only fill %6.1f us   after a 256 MB "
          "fill %6.1f us   after a 64 MB fill %6.1f us"
          % (cold(pre=lambda: big_a.sum()), cold(pre=lambda: big_b.fill_(1.0)),
             cold(pre=lambda: big_b[:64 * 1024 * 1024].fill_(1.0)), cold(pre=lambda: big_b[:16 * 1024 * 1024].fill_(1.0))),
          flush=True)


if __name__ == "__main__":
    for nt in (1_000_000, 100_000):
        run(nt)
