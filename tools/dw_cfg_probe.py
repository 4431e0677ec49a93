#!/usr/bin/env python3
"""Weight-gradient GEMM (A given as [K, M], K = batch) of wide layers: which tile configuration wins?
REC_GEMM_FORCE_CFG is read once per process, so this script runs itself once per configuration.
    python tools/dw_cfg_probe.py            # all configurations, shapes of DCN-v2 (1560^2) and slot_dnn layer 0"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NAMES = {-1: "planner", 0: "128x80", 1: "256x80", 2: "256x128", 3: "128x128", 4: "80x80", 5: "64x80"}
DW = [(1560, 1560, 65536), (3672, 512, 65536), (768, 768, 65536), (1560, 768, 65536), (400, 400, 65536), (416, 400, 65536)]
FWD = [(65536, 1560, 1560), (65536, 768, 1560), (65536, 512, 3672), (65536, 400, 416), (65536, 256, 512)]
SHAPES = [s + (True,) for s in DW] + [s + (False,) for s in FWD]
if os.environ.get("PROBE_ONLY"):       # e.g. PROBE_ONLY=400x400x65536 under rocprofv3
    SHAPES = [s for s in SHAPES if "%dx%dx%d" % s[:3] == os.environ["PROBE_ONLY"]]
if len(sys.argv) > 1:
    import torch
    from paddlerec_amd import ops
    g = torch.Generator(device="cuda").manual_seed(1)
    ws = ops.Workspace("cuda")
    out = []
    for M, N, K, ta in SHAPES:
        a = torch.randn((K, M) if ta else (M, K), device="cuda", generator=g)
        b = torch.randn(K, N, device="cuda", generator=g)
        c = torch.empty(M, N, device="cuda")
        db = torch.empty(N, device="cuda")
        bias = torch.zeros(N, device="cuda")
        fn = (lambda: ops.gemm(a, b, ws, trans_a=True, out=c, b_colsum=db)) if ta else \
             (lambda: ops.gemm(a, b, ws, bias=bias, epilogue="bias_relu", out=c))
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 5
        out.append("%6.2f ms %5.1f TF" % (t, 2.0 * M * N * K / t / 1e9) + ("\n         " if (M, N, K) == DW[-1] else ""))
    print("%-8s " % NAMES[int(sys.argv[1])] + "   ".join(out))
else:
    print("weight gradients (A as [K, M]): " + "  ".join("%dx%dx%d" % s for s in DW))
    print("forward (bias + relu):          " + "  ".join("%dx%dx%d" % s for s in FWD))
    for cfg in (-1, 4, 3, 2, 1, 0):
        env = dict(os.environ)
        if cfg >= 0:
            env["REC_GEMM_FORCE_CFG"] = str(cfg)
        r = subprocess.run([sys.executable, __file__, str(cfg)], env=env, capture_output=True, text=True)
        print("\n".join((r.stdout.strip().splitlines() or [r.stderr[-300:]])[-2:]))
