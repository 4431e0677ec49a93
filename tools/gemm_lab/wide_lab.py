#!/usr/bin/env python3
"""gemm_f32_pipe_kernel<256, 80> on 8 waves of 32 x 80 (shipped) against 4 waves of 64 x 80 (tools/gemm_lab/wide_lab.hip).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ipaddlerec_amd/csrc -fPIC -shared tools/gemm_lab/wide_lab.hip \
        -o tools/gemm_lab/_build/libwidelab.so
    python tools/gemm_lab/wide_lab.py [--iters 20] [--rounds 3]"""
import argparse
import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch  # noqa: E402

EPI = {"none": 0, "bias": 1, "bias_relu": 2, "relu_mask": 3}
SHAPES = [("fwd1  65536x400x400 bias_relu", 65536, 400, 400, False, "bias_relu"),
          ("dX_1  65536x400x400 relu_mask", 65536, 400, 400, True, "relu_mask"),
          ("fwd0' 65536x400x432 bias_relu", 65536, 400, 432, False, "bias_relu"),
          ("dX    65536x3680x512 none", 65536, 3680, 512, True, "none")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--variants", default="0,1")
    ap.add_argument("--probe", action="store_true", help="what bounds the 400-wide GEMM: longer K (per-tile fixed cost) and more rows (rounds of resident blocks)")
    ap.add_argument("--lib", default=os.path.join(HERE, "_build", "libwidelab.so"))
    args = ap.parse_args()
    lab = C.CDLL(args.lib)
    variants = [int(v) for v in args.variants.split(",")]
    g = torch.Generator(device="cuda").manual_seed(1)
    rnd = lambda *s: torch.rand(*s, device="cuda", generator=g) - 0.5
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
    shapes = SHAPES
    if args.probe:
        shapes = [("M 65536  N 400 K 400", 65536, 400, 400, False, "bias_relu"), ("M 65536  N 400 K 1600", 65536, 400, 1600, False, "bias_relu"),
                  ("M 655360 N 400 K 400", 655360, 400, 400, False, "bias_relu"), ("M 102400 N 400 K 400 (1 round x 2000 tiles / 512 slots)", 102400, 400, 400, False, "bias_relu"),
                  ("M 26112  N 400 K 400 (510 tiles: one round)", 26112, 400, 400, False, "bias_relu"),
                  ("M 65536  N 800 K 400", 65536, 800, 400, False, "bias_relu"), ("M 65536  N 1600 K 400", 65536, 1600, 400, False, "bias_relu")]
    for name, M, N, K, tb, epi in shapes:
        A, B, bias, X0 = rnd(M, K), (rnd(N, K) if tb else rnd(K, N)), rnd(N), rnd(M, N)
        outs = {v: torch.zeros(M, N, device="cuda") for v in variants}
        bias_ = bias if epi.startswith("bias") else None
        aux_ = X0 if epi == "relu_mask" else None

        def run(v):
            rc = lab.lab_wide(v, C.c_int64(M), N, K, p(A), C.c_int64(K), p(B), C.c_int64(B.stride(0)), p(outs[v]), C.c_int64(N),
                              int(tb), EPI[epi], p(bias_), p(aux_), N, C.c_void_p(st))
            assert rc == 0, rc
        best = {}
        for _ in range(args.rounds):
            for v in variants:
                run(v)
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(args.iters):
                    run(v)
                b.record()
                torch.cuda.synchronize()
                best[v] = min(best.get(v, 1e9), a.elapsed_time(b) / args.iters)
        same = all(torch.equal(outs[variants[0]], outs[v]) for v in variants[1:])
        ref = (A.double() @ (B.double().t() if tb else B.double()))[:64]
        got = outs[variants[0]][:64].double()
        if epi == "bias_relu":
            ref = torch.relu(ref + bias.double())
        elif epi == "relu_mask":
            ref = torch.where(X0[:64] > 0, ref, torch.zeros_like(ref))
        err = float((got - ref).abs().max() / ref.abs().max())
        fl = 2.0 * M * N * K
        print("%-32s %s   identical %s  err %.1e" % (name, "   ".join("v%d %.1f us %.1f TF" % (v, t * 1e3, fl / t / 1e9)
                                                                    for v, t in best.items()), same, err), flush=True)


if __name__ == "__main__":
    main()
