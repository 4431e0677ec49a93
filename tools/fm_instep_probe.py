#!/usr/bin/env python3
"""Why is fm_fwd_kernel ~15 us slower inside a training step than in back-to-back repeats?
Times ONE fm_fwd launch (HIP events around it) right after a controlled predecessor, 12 times each, median:

    repeat      : the previous launch was fm_fwd itself (what time_fm_pair / rocprof's repeats see)
    idle        : 3 ms of host sleep before it (clock / power state after an idle gap)
    gemm        : the step's MLP forward + backward GEMM chain before it (MFMA phase, ~2 ms; no table traffic)
    sparse_adam : the step's sparse_adam_record before it (leaves ~400 MB of dirty table lines behind)
    copy        : a 512 MB device copy before it (dirty lines in L2 / Infinity Cache, no table traffic)
    step        : a whole train_step before it (the in-step population)
    step+touch  : a whole train_step, then a read sweep of 256 MB (evicts the dirty lines), then fm_fwd

Prints one line per condition; run on the GPU box (gpurun)."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

import bench  # noqa: E402
from paddlerec_amd.deepfm import DeepFMLayer  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    B, S, Dn, D, fc = 65536, 26, 13, 16, [400, 400, 400]
    rows = 1_000_000
    so = torch.arange(S, dtype=torch.int64, device=dev) * rows
    torch.manual_seed(20250404)
    m = DeepFMLayer(rows * S, D, Dn, S, fc, device=dev, slot_offset=so)
    batches = bench.make_batches(4, B, S, Dn, rows, dev, 20250404)
    for i in range(14):
        m.train_step(*batches[i % 4], lr=1e-3)
    torch.cuda.synchronize()
    k = m.k
    big_a = torch.empty(128 * 1024 * 1024, dtype=torch.float32, device=dev)     # 512 MB
    big_b = torch.empty_like(big_a)
    sweep = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=dev)      # 256 MB
    mlp_w, mlp_dw = m._mlp_weights()
    state = {"i": 0}

    def fm_fwd():
        ids, dense, _ = batches[state["i"] % 4]
        state["i"] += 1
        return m._fm_fwd(m._concat_ids(ids), dense)

    y1, y2, feat, sum_emb, _ = fm_fwd()
    dz = torch.randn(B, 1, device=dev) * 1e-3

    def gemm_chain():
        y_dnn, acts = k.mlp_forward(feat.view(B, -1), mlp_w, m.mlp_b, m.ws_mlp)
        d_flat, fin = k.mlp_backward(dz, acts, mlp_w, mlp_dw, m.mlp_db, m.ws_mlp, defer_first=True)
        fin()

    def pre_step():
        m.train_step(*batches[(state["i"] + 1) % 4], lr=1e-3)

    def pre_adam():
        # a step leaves its groups / row_grad behind: re-running the record update on them is the same traffic
        st = m.sparse_state
        k.sparse_adam_record(m._groups, m._rg, dz, S, m.fm.rec, st["mv"], D, m.step_count, 1e-3,
                             v_offset=16, partials=m._pp, partials1=m._pp1)

    conds = [
        ("repeat", lambda: fm_fwd()),
        ("idle", lambda: (torch.cuda.synchronize(), time.sleep(0.003))),
        ("gemm", gemm_chain),
        ("sparse_adam", pre_adam),
        ("copy", lambda: big_b.copy_(big_a)),
        ("step", pre_step),
        ("step+touch", lambda: (pre_step(), sweep.sum())),
    ]
    print("fm_fwd_kernel after a controlled predecessor (B %d, 26M-row table), us: median [min .. max] of 12" % B)
    for name, pre in conds:
        ts = []
        for _ in range(12):
            pre()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fm_fwd()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        ts.sort()
        print("  %-12s %6.1f  [%6.1f .. %6.1f]" % (name, ts[len(ts) // 2], ts[0], ts[-1]), flush=True)


if __name__ == "__main__":
    main()
