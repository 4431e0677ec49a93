#!/usr/bin/env python3
"""csrc/gemm_wave80.h (one wave per 80 x 80 tile and K slice) against the engine's weight-gradient GEMMs.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ipaddlerec_amd/csrc -fPIC -shared tools/gemm_lab/wave80_lab.hip \
        -o tools/gemm_lab/_build/libwave80lab.so
    python tools/gemm_lab/wave80_lab.py [--iters 20] [--rounds 3]"""
import argparse
import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch  # noqa: E402

from paddlerec_amd import ops  # noqa: E402

SHAPES = [("dW_1 400x400", 400, 400, 65536), ("dW 480x400", 480, 400, 65536), ("dW 400x400 K 8192", 400, 400, 8192),
          ("dW 1600x1600", 1600, 1600, 65536)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--splits", type=int, default=0)
    ap.add_argument("--lib", default=os.path.join(HERE, "_build", "libwave80lab.so"))
    args = ap.parse_args()
    lab = C.CDLL(args.lib)
    g = torch.Generator(device="cuda").manual_seed(1)
    rnd = lambda *s: torch.rand(*s, device="cuda", generator=g) - 0.5
    ws = ops.Workspace("cuda")
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
    for name, M, N, K in SHAPES:
        X, dZ = rnd(K, M), rnd(K, N)
        ref, db_ref = torch.empty(M, N, device="cuda"), torch.empty(N, device="cuda")
        ops.gemm(X, dZ, ws, trans_a=True, out=ref, b_colsum=db_ref)
        want = (X.double().t() @ dZ.double())
        splits = args.splits or lab.lab_dw80_splits(C.c_int64(M), N, K)
        part = torch.empty(splits, M, N, device="cuda")
        cpart = torch.empty(splits, N, device="cuda")
        line = "%-20s splits %3d  blocks %5d" % (name, splits, splits * (M // 80) * (N // 80))
        for perm in (0, 1):
            part.fill_(float("nan"))
            rc = lab.lab_dw80(C.c_int64(M), N, K, p(X), C.c_int64(M), p(dZ), C.c_int64(N), C.c_int64(N), splits, p(part),
                              p(cpart), perm, C.c_void_p(st))
            assert rc == 0, rc
            got = part.double().sum(0)
            err = float((got - want).abs().max() / want.abs().max())
            err_ref = float((ref.double() - want).abs().max() / want.abs().max())
            db_err = float((cpart.double().sum(0) - dZ.double().sum(0)).abs().max())
            line += "  perm %d err %.1e (tiled %.1e, db %.1e)" % (perm, err, err_ref, db_err)
        print(line, flush=True)
        fl = 2.0 * M * N * K

        def tiled():
            ops.gemm(X, dZ, ws, trans_a=True, out=ref, b_colsum=db_ref)
        fns = [("tiled+reduce", tiled)]
        for perm in (0, 1):
            fns.append(("wave80 perm %d" % perm, lambda perm=perm: lab.lab_dw80(
                C.c_int64(M), N, K, p(X), C.c_int64(M), p(dZ), C.c_int64(N), C.c_int64(N), splits, p(part), p(cpart), perm,
                C.c_void_p(st))))
        best = {}
        for _ in range(args.rounds):
            for nm, fn in fns:
                fn()
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(args.iters):
                    fn()
                b.record()
                torch.cuda.synchronize()
                t = a.elapsed_time(b) / args.iters
                best[nm] = min(best.get(nm, 1e9), t)
        print("    " + "   ".join("%s %.1f us %.1f TF" % (nm, t * 1e3, fl / t / 1e9) for nm, t in best.items()), flush=True)


if __name__ == "__main__":
    main()
