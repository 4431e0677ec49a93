#!/usr/bin/env python3
"""Ranks the GEMM lab configurations (tools/gemm_lab/gemm_lab.hip) on the hot path's shapes.  One table per shape:
TFLOP/s of every configuration that passes the float64 check, and of the engine's rec_gemm_f32 beside them.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared tools/gemm_lab/gemm_lab.hip -o tools/gemm_lab/_build/libgemmlab.so
    python tools/gemm_lab/run.py [--shapes fwd0,dw0,...] [--iters 10]"""
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
SHAPES = {   # name: (M, N, K, ta, tb, splits)
    "fwd0": (B, 400, 432, 0, 0, 1),      # layer 0 on the folded input
    "fwd1": (B, 400, 400, 0, 0, 1),
    "dx0": (B, 432, 400, 0, 1, 1),       # g @ W0'^T  (B given as [N,K])
    "dx1": (B, 400, 400, 0, 1, 1),
    "dw0": (432, 400, B, 1, 0, 64),      # feat'^T dZ (A given as [K,M]), split-K
    "dw1": (400, 400, B, 1, 0, 64),
    "cross": (B, 1560, 1560, 0, 0, 1),   # CrossNetV2 layer
    "slot0": (B, 512, 3672, 0, 0, 1),    # slot_dnn layer 0
    "sq4096": (4096, 4096, 4096, 0, 0, 1),
}


def timeit(fn, iters):
    fn()
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
    ap.add_argument("--shapes", default=",".join(SHAPES))
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--splits", default="")     # e.g. "32,64,128" to sweep split-K on the dW shapes
    ap.add_argument("--engine-only", action="store_true")
    args = ap.parse_args()
    lib = C.CDLL(os.path.join(HERE, "_build", "libgemmlab.so"))
    lib.lab_config_name.restype = C.c_char_p
    ncfg = lib.lab_num_configs()
    g = torch.Generator(device=DEV).manual_seed(1)
    ws_e = ops.Workspace(DEV)
    for name in args.shapes.split(","):
        M, N, K, ta, tb, splits0 = SHAPES[name]
        A = torch.rand((K, M) if ta else (M, K), device=DEV, generator=g) * 2 - 1
        Bm = torch.rand((N, K) if tb else (K, N), device=DEV, generator=g) * 2 - 1
        ref = ((A.t() if ta else A).double() @ (Bm.t() if tb else Bm).double())
        scale = float(ref.abs().max())
        flops = 2.0 * M * N * K
        out = torch.empty(M, N, device=DEV)
        ms = timeit(lambda: ops.gemm(A, Bm, ws_e, trans_a=bool(ta), trans_b=bool(tb), out=out), args.iters)
        err = float((out.double() - ref).abs().max()) / scale
        print("== %-7s M=%d N=%d K=%d ta=%d tb=%d   engine rec_gemm_f32: %.3f ms  %.1f TF  (err %.1e)"
              % (name, M, N, K, ta, tb, ms, flops / ms / 1e9, err), flush=True)
        if args.engine_only:
            continue
        split_list = [splits0] if splits0 == 1 or not args.splits else [int(x) for x in args.splits.split(",")]
        for cfg in range(ncfg):
            cname = lib.lab_config_name(cfg).decode()
            for sp in split_list:
                nfl = C.c_int64(0)
                lib.lab_workspace_floats(C.c_int64(M), C.c_int64(N), sp, C.byref(nfl))
                wsl = torch.empty(max(nfl.value, 1), device=DEV)
                out.zero_()

                def run():
                    return lib.lab_gemm(cfg, ta, tb, C.c_int64(M), N, K, C.c_void_p(A.data_ptr()),
                                        C.c_int64(A.stride(0)), C.c_void_p(Bm.data_ptr()), C.c_int64(Bm.stride(0)),
                                        C.c_void_p(out.data_ptr()), C.c_int64(N), sp, C.c_void_p(wsl.data_ptr()),
                                        C.c_void_p(torch.cuda.current_stream().cuda_stream))
                rc = run()
                if rc != 0:
                    continue
                torch.cuda.synchronize()
                err = float((out.double() - ref).abs().max()) / scale
                if not err < 1e-5:
                    print("   cfg %2d %-28s splits %3d  WRONG (err %.2e)" % (cfg, cname, sp, err), flush=True)
                    continue
                ms = timeit(run, args.iters)
                print("   cfg %2d %-28s splits %3d  %.3f ms  %6.1f TF" % (cfg, cname, sp, ms, flops / ms / 1e9),
                      flush=True)
        ms = timeit(lambda: ops.gemm(A, Bm, ws_e, trans_a=bool(ta), trans_b=bool(tb), out=out), args.iters)
        print("   engine again (after the lab configs): %.3f ms  %.1f TF" % (ms, flops / ms / 1e9), flush=True)


if __name__ == "__main__":
    main()
