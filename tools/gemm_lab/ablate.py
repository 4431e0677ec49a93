#!/usr/bin/env python3
"""Runs tools/gemm_lab/ablate.hip: the engine's GEMM loop with one piece removed per variant, on the hot shapes.
One table per shape: time of every variant and what removing the piece saved.  ABL 0 is checked against float64 and
printed beside the engine's rec_gemm_f32 (same shape, interior tiles).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared tools/gemm_lab/ablate.hip -o tools/gemm_lab/_build/libgemmablate.so
    python tools/gemm_lab/ablate.py [--iters 10]"""
import argparse
import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch  # noqa: E402

from paddlerec_amd import ops  # noqa: E402

DEV = "cuda"
B = 65536
NAMES = ["nothing removed", "all blocks load tile (0,0)", "no global loads", "no LDS stores", "no barrier",
         "no fragment reads", "no loads/stores/barrier", "MFMAs only", "2 tiles in flight, stores fenced",
         "2 tiles in flight, stores in MFMAs"]
SHAPES = [   # name, cfg, form (0 A@B, 1 A@B^T, 2 A^T@B), M, N, K, splits
    ("DeepFM forward 256x80", 0, 0, B, 400, 416, 1),
    ("DeepFM dX 256x80 (B as [N,K])", 0, 1, B, 400, 400, 1),
    ("DeepFM dW 80x80 split-K 40", 1, 2, 400, 400, 65280, 40),
    ("slot_dnn layer 0 256x128", 2, 0, B, 512, 3680, 1),
    ("1536^2 dW 128x128 split-K 8", 3, 2, 1536, 1536, B, 8),
]


def timeit(fn, iters):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    lib = C.CDLL(os.path.join(HERE, "_build", "libgemmablate.so"))
    g = torch.Generator(device=DEV).manual_seed(1)
    ws = ops.Workspace(DEV)
    for name, cfg, form, M, N, K, splits in SHAPES:
        ta, tb = form == 2, form == 1
        A = torch.rand((K, M) if ta else (M, K), device=DEV, generator=g) * 2 - 1
        Bm = torch.rand((N, K) if tb else (K, N), device=DEV, generator=g) * 2 - 1
        out = torch.empty(M, N, device=DEV)
        part = torch.empty(splits, M, N, device=DEV) if splits > 1 else out
        flops = 2.0 * M * N * K

        def run(abl):
            return lib.ablate_gemm(cfg, form, abl, C.c_int64(M), N, K, C.c_void_p(A.data_ptr()), C.c_int64(A.stride(0)),
                                   C.c_void_p(Bm.data_ptr()), C.c_int64(Bm.stride(0)), C.c_void_p(out.data_ptr()),
                                   C.c_int64(N), splits, C.c_void_p(part.data_ptr()),
                                   C.c_void_p(torch.cuda.current_stream().cuda_stream))
        rc = run(0)
        torch.cuda.synchronize()
        if rc != 0:
            print("== %s: unsupported (rc %d)" % (name, rc))
            continue
        ref = (A.t() if ta else A).double() @ (Bm.t() if tb else Bm).double()
        got = part.double().sum(0) if splits > 1 else out.double()
        err = float((got - ref).abs().max()) / float(ref.abs().max())
        ms_e = timeit(lambda: ops.gemm(A, Bm, ws, trans_a=ta, trans_b=tb, out=out), args.iters)
        print("== %s  M=%d N=%d K=%d   engine rec_gemm_f32 (with its split-K reduce): %.3f ms %.1f TF;  ABL 0 err %.1e"
              % (name, M, N, K, ms_e, flops / ms_e / 1e9, err), flush=True)
        # clocks ramp over the first tens of milliseconds of load: warm up on ABL 0, then time the variants round-robin
        # (three rounds, median) so that no variant owes its number to its place in the order
        ok = []
        for abl in range(10):
            if run(abl) != 0:
                continue
            if abl >= 8:        # the candidates must be right
                torch.cuda.synchronize()
                got = part.double().sum(0) if splits > 1 else out.double()
                e = float((got - ref).abs().max()) / float(ref.abs().max())
                if not e < 1e-5:
                    print("   ABL %d %-34s WRONG (err %.2e)" % (abl, NAMES[abl], e), flush=True)
                    continue
            ok.append(abl)
        for _ in range(int(150.0 / max(ms_e, 0.05))):     # ~150 ms of load
            run(0)
        torch.cuda.synchronize()
        times = {abl: [] for abl in ok}
        times["engine"] = []
        for _ in range(3):
            for abl in ok:
                times[abl].append(timeit(lambda: run(abl), args.iters))
            times["engine"].append(timeit(lambda: ops.gemm(A, Bm, ws, trans_a=ta, trans_b=tb, out=out), args.iters))
        med = {k: sorted(v)[1] for k, v in times.items()}
        print("   engine rec_gemm_f32, same protocol     %.3f ms  %6.1f TF" % (med["engine"], flops / med["engine"] / 1e9))
        for abl in ok:
            print("   ABL %d %-34s %.3f ms  %6.1f TF-equivalent   %+5.1f %% vs ABL 0   (rounds: %s)"
                  % (abl, NAMES[abl], med[abl], flops / med[abl] / 1e9, (med[abl] / med[0] - 1) * 100,
                     " ".join("%.3f" % t for t in times[abl])), flush=True)


if __name__ == "__main__":
    main()
