#!/usr/bin/env python3
"""The row-panel GEMM (csrc/gemm_panel.h) against the tiled kernels on the DeepFM tower's shapes: bit-identity and
time, REC_GEMM_PANEL flipped per call inside one process.   python tools/gemm_panel_probe.py [--iters 20]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from paddlerec_amd import ops  # noqa: E402

SHAPES = [  # name, M, N, K, tb, epilogue
    ("fwd0", 65536, 400, 432, False, "bias_relu"),
    ("fwd1", 65536, 400, 400, False, "bias_relu"),
    ("dx1", 65536, 400, 400, True, "relu_mask"),
    ("dx0", 65536, 432, 400, True, "none"),
    ("fwd1_b32k", 32768, 400, 400, False, "bias_relu"),
    ("fwd1_b128k", 131072, 400, 400, False, "bias"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    g = torch.Generator(device="cuda").manual_seed(1)
    rnd = lambda *s: torch.rand(*s, device="cuda", generator=g) - 0.5
    ws = ops.Workspace("cuda")
    for name, M, N, K, tb, epi in SHAPES:
        A, B, bias, X0 = rnd(M, K), (rnd(N, K) if tb else rnd(K, N)), rnd(N), rnd(M, N)
        outs, times = {}, {}
        for mode in ("0", "1"):
            os.environ["REC_GEMM_PANEL"] = mode
            C = torch.zeros(M, N, device="cuda")
            run = lambda: ops.gemm(A, B, ws, trans_b=tb, epilogue=epi, bias=bias if epi.startswith("bias") else None,
                                   aux0=X0 if epi == "relu_mask" else None, out=C)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.iters):
                run()
            b.record()
            torch.cuda.synchronize()
            times[mode] = a.elapsed_time(b) / args.iters * 1e3
            outs[mode] = C
        ref = (A.double() @ (B.double().t() if tb else B.double()))
        if epi.startswith("bias"):
            ref = ref + bias.double()
        if epi == "bias_relu":
            ref = ref.clamp_min(0)
        if epi == "relu_mask":
            ref = ref * (X0 > 0)
        err = float((outs["1"].double() - ref).abs().max() / ref.abs().max())
        fl = 2.0 * M * N * K
        print("%-11s M=%6d N=%d K=%d tb=%d %-9s tiled %7.1f us %6.1f TF | panel %7.1f us %6.1f TF | identical=%s err=%.1e"
              % (name, M, N, K, tb, epi, times["0"], fl / times["0"] / 1e6, times["1"], fl / times["1"] / 1e6,
                 bool(torch.equal(outs["0"], outs["1"])), err), flush=True)


if __name__ == "__main__":
    main()
