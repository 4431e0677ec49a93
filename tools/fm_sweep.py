#!/usr/bin/env python3
"""Launch-geometry sweep of the two FM kernels on the bench shape (BASELINE configs[1], record layout, compact feat).

    python tools/fm_sweep.py            # parent: one child process per knob setting (the knobs are read once per process)
    python tools/fm_sweep.py --child    # child: times the kernels under the REC_FM_* environment it was given

Timing = R back-to-back launches between ONE pair of HIP events (what rocprofv3's kernel duration sees, plus the
~1.5 us launch boundary), median of 5 repeats; 4 different id batches are cycled so no launch re-reads the lines of
the previous one from L2 / Infinity Cache (the table is 3.3 GB)."""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    import torch
    from paddlerec_amd import ops
    DEV = "cuda"
    B, S, Dn, D, NT = 65536, 26, 13, 16, 1_000_000
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
    out = ops.deepfm_fm_fwd(batches[0], dense, W, W1, dw, dw1, 0, so, status, compact=True)
    y1, y2, feat, sum_emb, _ = out
    dfeat = torch.randn(B, S + 1, D, device=DEV, generator=g) * 1e-3
    dz = torch.randn(B, 1, device=DEV, generator=g) * 1e-3
    ws = ops.Workspace(DEV)
    o = ops.deepfm_fm_bwd(dense, feat, sum_emb, dfeat, dz, dz, S, ws, dense_w=dw, compact=True)

    def timeit(fn, R=40, reps=5):
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(R):
                fn(i)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3 / R)
        ts.sort()
        return ts[len(ts) // 2]

    fwd = timeit(lambda i: ops.deepfm_fm_fwd(batches[i % 4], dense, W, W1, dw, dw1, 0, so, status,
                                             (y1, y2, feat, sum_emb), compact=True))
    bwd = timeit(lambda i: ops.deepfm_fm_bwd(dense, feat, sum_emb, dfeat, dz, dz, S, ws, o, dense_w=dw, compact=True))
    # reference points on this box
    x = torch.empty(256 * 1024 * 1024 // 4, device=DEV)
    y = torch.empty_like(x)
    cp = timeit(lambda i: y.copy_(x), R=10)
    print(json.dumps(dict(fwd_us=fwd, bwd_us=bwd, copy256MB_us=cp,
                          env={k: v for k, v in os.environ.items() if k.startswith("REC_FM_")})))


def main():
    if "--child" in sys.argv:
        return child()
    configs = [{}]
    for nt in (0, 1):
        for fb in (0, 2, 3):
            for bb in (0, 2):
                if nt or fb or bb:
                    configs.append({"REC_FM_NT": str(nt), "REC_FM_FWD_BPC": str(fb), "REC_FM_BWD_BPC": str(bb)})
    fwd_b = 65536 * 4532
    bwd_b = 65536 * 6140
    for c in configs:
        env = dict(os.environ)
        env.update(c)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True,
                           text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line:
            print("FAILED", c, r.stderr[-300:])
            continue
        d = json.loads(line[-1])
        print("%-60s fwd %6.1f us (%4.2f TB/s alg)  bwd %6.1f us (%4.2f TB/s alg)  pair frac %.3f   copy %5.1f us"
              % (c, d["fwd_us"], fwd_b / d["fwd_us"] / 1e6, d["bwd_us"], bwd_b / d["bwd_us"] / 1e6,
                 (fwd_b + bwd_b) / (d["fwd_us"] + d["bwd_us"]) / 1e6 / 8.0, d["copy256MB_us"]), flush=True)


if __name__ == "__main__":
    main()
