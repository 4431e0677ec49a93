#!/usr/bin/env python3
"""A/B of the LDS-DMA GEMM kernel (REC_GEMM_GLDS) on the DeepFM MLP shapes: each variant in its own child process (the
knob is read once), R back-to-back launches per HIP-event pair, median of 5; forward (bias+ReLU, B [K,N]), dX with the
ReLU mask (B [N,K]) and the plain layer-0 dX.  TFLOP/s = 2 M N K / t."""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    import torch
    from paddlerec_amd import ops
    DEV = "cuda"
    M = 65536
    g = torch.Generator(device=DEV).manual_seed(1)
    rnd = lambda *s: torch.rand(*s, device=DEV, generator=g) - 0.5
    ws = ops.Workspace(DEV)
    out = {}
    cases = [("fwd L0 65536x400x432 bias_relu", 400, 432, False, "bias_relu"),
             ("fwd L1 65536x400x400 bias_relu", 400, 400, False, "bias_relu"),
             ("dX  L1 65536x400x400 relu_mask", 400, 400, True, "relu_mask"),
             ("dX  L0 65536x432x400 none", 432, 400, True, "none"),
             ("dX  65536x416x400 none (26 x 16)", 416, 400, True, "none")]
    for name, N, K, tb, epi in cases:
        # three operand sets cycled: 3 x (A + C) > Infinity Cache, as in the step
        sets = [(rnd(M, K), rnd(N, K) if tb else rnd(K, N), rnd(N), rnd(M, N), torch.empty(M, N, device=DEV))
                for _ in range(3)]

        def run(i):
            A, B, bias, X0, C = sets[i % 3]
            ops.gemm(A, B, ws, trans_b=tb, epilogue=epi, bias=bias if epi == "bias_relu" else None,
                     aux0=X0 if epi == "relu_mask" else None, out=C)
        for i in range(6):
            run(i)
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(12):
                run(i)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / 12)
        t = sorted(ts)[2]
        out[name] = (t * 1e3, 2.0 * M * N * K / (t * 1e-3) / 1e12)
    print(json.dumps(out))


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
    else:
        res = {}
        for tag, env in (("register-staged (REC_GEMM_GLDS=0)", {"REC_GEMM_GLDS": "0"}),
                         ("LDS-DMA ring (all shapes)", {"REC_GEMM_GLDS_80": "1"})):
            r = subprocess.run([sys.executable, __file__, "--child"], env=dict(os.environ, **env), capture_output=True,
                               text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(tag, "FAILED", r.stderr[-800:])
                continue
            res[tag] = json.loads(line[-1])
        for case in next(iter(res.values())):
            print("%-36s" % case, "   ".join("%s: %6.1f us %6.1f TF" % (k[:16], v[case][0], v[case][1]) for k, v in res.items()))
