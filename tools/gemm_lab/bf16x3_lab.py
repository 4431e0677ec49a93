#!/usr/bin/env python3
"""csrc/gemm_bf16x3.h (f32 GEMM as six bf16 MFMAs per product, exact three-way operand split) built alone, against the
engine's exact-f32 tiled kernels and a float64 product.

    bash tools/gemm_lab/build_x3.sh && python tools/gemm_lab/bf16x3_lab.py [--iters 20] [--rounds 3] [--data rand|ones|zeros]"""
import argparse
import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch  # noqa: E402

from paddlerec_amd import ops  # noqa: E402

EPI = {"none": 0, "bias": 1, "bias_relu": 2, "relu_mask": 3}
SHAPES = [("fwd0", 65536, 400, 432, False, "bias_relu"), ("fwd1", 65536, 400, 400, False, "bias_relu"),
          ("dx1", 65536, 400, 400, True, "relu_mask"), ("dx0'", 65536, 400, 400, True, "none"),
          ("dx0", 65536, 432, 400, True, "none"), ("dcn", 65536, 1560, 1560, False, "bias_relu"),
          ("slot", 65536, 512, 3680, False, "bias_relu"), ("edge", 1000, 396, 104, False, "bias"),
          ("small", 512, 400, 400, False, "bias_relu")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--data", default="rand")
    ap.add_argument("--lib", default=os.path.join(HERE, "_build", "libx3lab.so"))
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    lab = C.CDLL(args.lib)
    lab.lab_x3_image_bytes.restype = C.c_size_t
    g = torch.Generator(device="cuda").manual_seed(1)
    if args.data == "rand":
        rnd = lambda *s: torch.rand(*s, device="cuda", generator=g) - 0.5
    elif args.data == "normal":      # wide dynamic range: products of very different magnitude in one sum
        rnd = lambda *s: torch.randn(*s, device="cuda", generator=g) * torch.exp(4 * torch.randn(*s, device="cuda", generator=g))
    else:
        rnd = lambda *s: torch.full(s, 1.0 if args.data == "ones" else 0.0, device="cuda")
    ws = ops.Workspace("cuda")
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
    for name, M, N, K, tb, epi in SHAPES:
        if args.only and name not in args.only.split(","):
            continue
        A, B, bias, X0 = rnd(M, K), (rnd(N, K) if tb else rnd(K, N)), rnd(N), rnd(M, N)
        C0, C1 = torch.zeros(M, N, device="cuda"), torch.full((M, N), 7.0, device="cuda")
        bias_ = bias if epi.startswith("bias") else None
        aux_ = X0 if epi == "relu_mask" else None
        img = torch.zeros(lab.lab_x3_image_bytes(K, N), dtype=torch.uint8, device="cuda")

        def split():
            rc = lab.lab_x3_split(p(B), C.c_int64(B.stride(0)), K, N, int(tb), p(img), C.c_void_p(st))
            assert rc == 0, rc

        def tiled():
            ops.gemm(A, B, ws, trans_b=tb, epilogue=epi, bias=bias_, aux0=aux_, out=C0)

        def x3():
            rc = lab.lab_x3_gemm(C.c_int64(M), N, K, p(A), C.c_int64(A.stride(0)), p(img), p(C1), C.c_int64(N), EPI[epi],
                                 p(bias_), p(aux_), N, C.c_void_p(st))
            assert rc == 0, rc
        split()
        best = {}
        for _ in range(args.rounds):
            for nm, fn in (("tiled", tiled), ("x3", x3), ("split", split)):
                fn()
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _i in range(args.iters):
                    fn()
                b.record()
                torch.cuda.synchronize()
                best[nm] = min(best.get(nm, 1e9), a.elapsed_time(b) / args.iters * 1e3)
        # float64 product (row blocks: 65536 x 400 doubles are small, the [M,K] x [K,N] product is cheap on the GPU)
        Bd = (B.t() if tb else B).double()
        ref = A.double() @ Bd
        mag = A.double().abs() @ Bd.abs()               # sum |a||b| per output: the scale rounding errors live on
        if epi.startswith("bias"):
            ref = ref + bias.double()
        if epi == "bias_relu":
            ref = ref.clamp_min(0)
        if epi == "relu_mask":
            ref = torch.where(X0 > 0, ref, torch.zeros_like(ref))
        scale = float(ref.abs().max())
        e0, e1 = (C0.double() - ref).abs(), (C1.double() - ref).abs()
        fl = 2.0 * M * N * K
        print("%-5s M=%d N=%d K=%d tb=%d %-9s tiled %6.1f us %6.1f TF | x3 %6.1f us %6.1f TF-equiv (x%.2f) | split %4.1f us | "
              "max err / scale: f32 %.2e  x3 %.2e | max err / sum|a||b|: f32 %.2e  x3 %.2e"
              % (name, M, N, K, tb, epi, best["tiled"], fl / best["tiled"] / 1e6, best["x3"], fl / best["x3"] / 1e6,
                 best["tiled"] / best["x3"], best["split"], float(e0.max()) / scale, float(e1.max()) / scale,
                 float((e0 / mag).max()), float((e1 / mag).max())), flush=True)


def dw_main(args):
    """the weight-gradient form through the engine (rec_gemm_f32, trans_a, b_colsum): switch off / on"""
    g = torch.Generator(device="cuda").manual_seed(2)
    rnd = lambda *s: torch.rand(*s, device="cuda", generator=g) - 0.5
    for name, rows, kin, nout in (("dw1", 65536, 400, 400), ("dw0", 65536, 432, 400), ("dwr", 8192 + 64, 400, 336)):
        X, G = rnd(rows, kin), rnd(rows, nout)
        out, cs = {}, {}
        best = {}
        for v in ("0", "1"):
            os.environ["REC_GEMM_BF16X3"] = v
            ws = ops.Workspace("cuda")
            C_, b_ = torch.zeros(kin, nout, device="cuda"), torch.zeros(nout, device="cuda")
            fn = lambda: ops.gemm(X, G, ws, trans_a=True, out=C_, b_colsum=b_)
            for _ in range(args.rounds):
                fn()
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _i in range(args.iters):
                    fn()
                b.record()
                torch.cuda.synchronize()
                best[v] = min(best.get(v, 1e9), a.elapsed_time(b) / args.iters * 1e3)
            out[v], cs[v] = C_.double(), b_.double()
        ref = X.double().t() @ G.double()
        mag = X.double().abs().t() @ G.double().abs()
        cref = G.double().sum(0)
        fl = 2.0 * rows * kin * nout
        print("%-4s rows=%d kin=%d nout=%d  f32 %6.1f us %6.1f TF | x3 %6.1f us %6.1f TF-equiv (x%.2f) | max err / sum|a||b|: "
              "f32 %.2e x3 %.2e | colsum max err / sum|g|: f32 %.2e x3 %.2e | differ=%s"
              % (name, rows, kin, nout, best["0"], fl / best["0"] / 1e6, best["1"], fl / best["1"] / 1e6, best["0"] / best["1"],
                 float(((out["0"] - ref).abs() / mag).max()), float(((out["1"] - ref).abs() / mag).max()),
                 float(((cs["0"] - cref).abs() / G.double().abs().sum(0)).max()),
                 float(((cs["1"] - cref).abs() / G.double().abs().sum(0)).max()), not torch.equal(out["0"], out["1"])), flush=True)


def dwk_main(args):
    """the weight-gradient KERNEL alone in a lab build (--lib): where its time goes (products 1, no loads)"""
    lab = C.CDLL(args.lib)
    g = torch.Generator(device="cuda").manual_seed(2)
    rnd = lambda *s: torch.rand(*s, device="cuda", generator=g) - 0.5
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: C.c_void_p(t.data_ptr())
    for name, rows, kin, nout in (("dw1", 65536, 400, 400), ("dw0", 65536, 432, 400)):
        X, G = rnd(rows, kin), rnd(rows, nout)
        P = torch.zeros(64, kin, nout, device="cuda")
        cp = torch.zeros(64, nout, device="cuda")
        sl = C.c_int(0)
        fn = lambda: lab.lab_x3_dw(kin, nout, C.c_int64(rows), p(X), C.c_int64(kin), p(G), C.c_int64(nout), p(P), C.c_int64(nout),
                                   p(cp), C.byref(sl), C.c_void_p(st))
        assert fn() == 0
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(args.rounds):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _i in range(args.iters):
                fn()
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / args.iters * 1e3)
        ref = X.double().t() @ G.double()
        got = P[: sl.value].double().sum(0)
        err = float(((got - ref).abs() / (X.double().abs().t() @ G.double().abs())).max())
        print("%-4s kernel alone %6.1f us  (%d slices)  max err / sum|a||b| %.2e   [%s]" % (name, best, sl.value, err, os.path.basename(args.lib)),
              flush=True)


if __name__ == "__main__":
    if "--dwk" in sys.argv:
        sys.argv.remove("--dwk")
        ap = argparse.ArgumentParser()
        ap.add_argument("--iters", type=int, default=20)
        ap.add_argument("--rounds", type=int, default=2)
        ap.add_argument("--lib", default=os.path.join(HERE, "_build", "libx3lab.so"))
        dwk_main(ap.parse_args())
        sys.exit(0)
    if "--dw" in sys.argv:
        sys.argv.remove("--dw")
        ap = argparse.ArgumentParser()
        ap.add_argument("--iters", type=int, default=20)
        ap.add_argument("--rounds", type=int, default=2)
        dw_main(ap.parse_args())
        sys.exit(0)
    main()
