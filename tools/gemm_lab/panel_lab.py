#!/usr/bin/env python3
"""csrc/gemm_panel.h built alone (seconds instead of the library's minutes) against the engine's tiled kernels.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ipaddlerec_amd/csrc -fPIC -shared tools/gemm_lab/panel_lab.hip \
        -o tools/gemm_lab/_build/libpanellab.so
    python tools/gemm_lab/panel_lab.py [--iters 20] [--rounds 3]"""
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
          ("dx1", 65536, 400, 400, True, "relu_mask"), ("dx0", 65536, 432, 400, True, "none")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--data", default="rand", help="rand | ones | zeros: a power / clock probe (same instructions, different toggling)")
    ap.add_argument("--lib", default=os.path.join(HERE, "_build", "libpanellab.so"))
    args = ap.parse_args()
    lab = C.CDLL(args.lib)
    os.environ["REC_GEMM_PANEL"] = "0"
    g = torch.Generator(device="cuda").manual_seed(1)
    if args.data == "rand":
        rnd = lambda *s: torch.rand(*s, device="cuda", generator=g) - 0.5
    elif args.data == "coarse":        # random signs and magnitudes, two mantissa bits
        rnd = lambda *s: torch.round((torch.rand(*s, device="cuda", generator=g) - 0.5) * 8) / 8
    else:
        rnd = lambda *s: torch.full(s, 1.0 if args.data == "ones" else 0.0, device="cuda")
    ws = ops.Workspace("cuda")
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
    for name, M, N, K, tb, epi in SHAPES:
        A, B, bias, X0 = rnd(M, K), (rnd(N, K) if tb else rnd(K, N)), rnd(N), rnd(M, N)
        C0, C1 = torch.zeros(M, N, device="cuda"), torch.zeros(M, N, device="cuda")
        bias_ = bias if epi.startswith("bias") else None
        aux_ = X0 if epi == "relu_mask" else None

        def tiled():
            os.environ["REC_GEMM_PANEL"] = "0"
            ops.gemm(A, B, ws, trans_b=tb, epilogue=epi, bias=bias_, aux0=aux_, out=C0)

        def panel():
            os.environ["REC_GEMM_PANEL"] = "1"
            rc = lab.lab_panel(C.c_int64(M), N, K, p(A), C.c_int64(A.stride(0)), p(B), C.c_int64(B.stride(0)), p(C1),
                               C.c_int64(N), int(tb), EPI[epi], p(bias_), p(aux_), N, C.c_void_p(st))
            assert rc == 0
        best = {}
        for rnd_i in range(args.rounds):
            for nm, fn in (("tiled", tiled), ("panel", panel)):
                fn()
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(args.iters):
                    fn()
                b.record()
                torch.cuda.synchronize()
                t = a.elapsed_time(b) / args.iters * 1e3
                best[nm] = min(best.get(nm, 1e9), t)
        fl = 2.0 * M * N * K
        print("%-5s M=%d N=%d K=%d tb=%d %-9s tiled %6.1f us %6.1f TF | panel %6.1f us %6.1f TF | identical=%s"
              % (name, M, N, K, tb, epi, best["tiled"], fl / best["tiled"] / 1e6, best["panel"], fl / best["panel"] / 1e6,
                 bool(torch.equal(C0, C1))), flush=True)


if __name__ == "__main__":
    main()
