#!/usr/bin/env python3
"""L-kernel microbenchmarks: single ops on device-resident inputs, HIP-event timed.
Used to find which part of a kernel's traffic sets its time; not part of bench.py's contract."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from paddlerec_amd import ops  # noqa: E402

DEV = "cuda"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2] * 1e3, ts[0] * 1e3   # median, min (us)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--dim", type=int, default=16)
    args = ap.parse_args()
    B, S, Dn, D = args.batch, 26, 13, args.dim
    g = torch.Generator(device=DEV).manual_seed(1)
    print("device:", torch.cuda.get_device_name(0))
    # reference points: plain copy / fill bandwidth on this box
    x = torch.empty(256 * 1024 * 1024 // 4, device=DEV)
    y = torch.empty_like(x)
    med, mn = timeit(lambda: y.copy_(x))
    print("copy 256MB->256MB: %.1f us  => %.0f GB/s (r+w)" % (med, 2 * x.numel() * 4 / med / 1e3))
    med, mn = timeit(lambda: y.fill_(1.0))
    print("fill 256MB: %.1f us => %.0f GB/s" % (med, x.numel() * 4 / med / 1e3))
    del x, y
    for rows_per_table, tables in ((1_000_000, True), (1_000_000, False), (40_000_000, False)):
        N = rows_per_table * (S if tables else 1)
        W = torch.randn(N, D, device=DEV, generator=g) * 0.02
        W1 = torch.randn(N, 1, device=DEV, generator=g) * 0.02
        dw = torch.randn(1, Dn, D, device=DEV, generator=g) * 0.02
        dw1 = torch.randn(Dn, device=DEV, generator=g) * 0.02
        ids = torch.randint(1, rows_per_table, (B, S), device=DEV, generator=g)
        ids[torch.rand(B, S, device=DEV, generator=g) < 0.03] = 0
        dense = torch.rand(B, Dn, device=DEV, generator=g)
        so = torch.arange(S, device=DEV, dtype=torch.int64) * rows_per_table if tables else None
        F = S + Dn
        fwd_b = B * S * 8 + B * S * D * 4 + B * S * 4 + B * Dn * 4 + B * F * D * 4 + B * 8
        bwd_b = B * S * 8 + B * F * D * 4 + B * 4 + B * S * D * 4 + B * S * D * 4 + B * S * 4
        out = ops.deepfm_fm_fwd(ids, dense, W, W1, dw, dw1, 0, so)
        y1, y2, feat, sum_emb, status = out
        # record layout: one 128-B line per row holds W and W1
        rec = torch.zeros(N, 32, device=DEV)
        rec[:, :D] = W
        rec[:, D:D + 1] = W1
        med, mn = timeit(lambda: ops.deepfm_fm_fwd(ids, dense, rec[:, :D], rec[:, D:D + 1], dw, dw1, 0, so,
                                                   status, (y1, y2, feat, sum_emb)))
        print("N=%d: fm_fwd[record layout] %.1f us (min %.1f) => %.0f GB/s algorithmic"
              % (N, med, mn, fwd_b / med / 1e3))
        cfeat = torch.empty(B, S + 1, D, device=DEV)
        med, mn = timeit(lambda: ops.deepfm_fm_fwd(ids, dense, rec[:, :D], rec[:, D:D + 1], dw, dw1, 0, so,
                                                   status, (y1, y2, cfeat, sum_emb), compact=True))
        print("N=%d: fm_fwd[record layout, compact feat] %.1f us (min %.1f) => %.0f GB/s algorithmic"
              % (N, med, mn, fwd_b / med / 1e3))
        mv = torch.zeros(N, 32, device=DEV)
        med, mn = timeit(lambda: ops.deepfm_fm_fwd(ids, dense, W, W1, dw, dw1, 0, so, status,
                                                   (y1, y2, feat, sum_emb)))
        print("N=%d (%.2f GB): fm_fwd %.1f us (min %.1f) => %.0f GB/s algorithmic"
              % (N, N * D * 4 / 1e9, med, mn, fwd_b / med / 1e3))
        rows = (ids + (so[None] if so is not None else 0)).reshape(-1)
        med, mn = timeit(lambda: ops.emb_gather(rows, W, None, status))
        print("   emb_gather %d rows: %.1f us => %.0f GB/s (r+w of rows)" % (rows.numel(), med,
              2 * rows.numel() * D * 4 / med / 1e3))
        med, mn = timeit(lambda: ops.emb_gather(rows, W1, None, status))
        print("   emb_gather W1 (4 B rows): %.1f us" % med)
        dfeat = torch.randn(B, F, D, device=DEV, generator=g) * 1e-3
        dz = torch.randn(B, 1, device=DEV, generator=g) * 1e-3
        ws = ops.Workspace(DEV)
        o = ops.deepfm_fm_bwd(dense, feat, sum_emb, dfeat, dz, dz, S, ws)
        med, mn = timeit(lambda: ops.deepfm_fm_bwd(dense, feat, sum_emb, dfeat, dz, dz, S, ws, o))
        print("   fm_bwd %.1f us (min %.1f) => %.0f GB/s algorithmic" % (med, mn, bwd_b / med / 1e3))
        med, mn = timeit(lambda: ops.deepfm_fm_bwd(dense, feat, sum_emb, dfeat, dz, dz, S, ws, o, dense_w=dw))
        print("   fm_bwd(recompute dense) %.1f us (min %.1f) => %.0f GB/s algorithmic" % (med, mn, bwd_b / med / 1e3))
        groups, _ = ops.ids_group(ids, N, 0, ws, so)
        med, mn = timeit(lambda: ops.ids_group(ids, N, 0, ws, so, status, groups))
        print("   ids_group %.1f us; uniq=%s" % (med, groups.n_uniq.tolist()))
        M = torch.zeros_like(W)
        V = torch.zeros_like(W)
        med, mn = timeit(lambda: ops.sparse_adam_rows(groups, o[0], 1, W, M, V, 1))
        U = groups.n_uniq[0].item()
        print("   sparse_adam_rows D=%d: %.1f us => %.0f GB/s (U*D*4*6 + n*D*4)" %
              (D, med, (U * D * 4 * 6 + B * S * D * 4) / med / 1e3))
        M1 = torch.zeros_like(W1)
        V1 = torch.zeros_like(W1)
        med, mn = timeit(lambda: ops.sparse_adam_rows(groups, dz, S, W1, M1, V1, 1))
        print("   sparse_adam_rows D=1: %.1f us" % med)
        g2, _ = ops.ids_group(ids, N, 0, ws, so)
        med, mn = timeit(lambda: ops.sparse_adam_rows(g2, o[0], 1, rec[:, :D], mv[:, :D], mv[:, D:2 * D], 1))
        med1, _ = timeit(lambda: ops.sparse_adam_rows(g2, dz, S, rec[:, D:D + 1], rec[:, D + 1:D + 2],
                                                      rec[:, D + 2:D + 3], 1))
        print("   sparse_adam_rows[record layout] D=16: %.1f us, D=1: %.1f us" % (med, med1))
        del W, W1, M, V, M1, V1, rec, mv
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
