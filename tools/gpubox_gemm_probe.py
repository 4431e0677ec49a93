#!/usr/bin/env python3
"""The three big GEMMs of the gpubox model's layer 0 (slot_dnn: 408 slots x D 9 = 3672 inputs, 512 outputs, batch 65536) on the
engine's kernels, as shipped and with the width padded to 3680 (= 230 x 16 = 46 x 80: whole tiles for every kernel) — what
would a padded activation layout buy?   python tools/gpubox_gemm_probe.py  [REC_GEMM_FORCE_CFG=<n> for one configuration]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from paddlerec_amd import ops  # noqa: E402

DEV = "cuda"
B, H = 65536, 512


def timeit(fn, iters=6):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / iters)
    return best


def main():
    g = torch.Generator(device=DEV).manual_seed(1)
    rnd = lambda *s: torch.rand(*s, device=DEV, generator=g) - 0.5
    ws = ops.Workspace(DEV)
    for W in (3672, 3680):
        x, w0, b0, dz = rnd(B, W), rnd(W, H), rnd(H), rnd(B, H)
        out, dx, dw, db = torch.empty(B, H, device=DEV), torch.empty(B, W, device=DEV), torch.empty(W, H, device=DEV), torch.empty(H, device=DEV)
        fl = 2.0 * B * W * H
        t = timeit(lambda: ops.gemm(x, w0, ws, epilogue="bias_relu", bias=b0, out=out))
        print("width %d  fwd0  [B,%d]x[%d,512]      %7.3f ms %6.1f TF" % (W, W, W, t, fl / t / 1e9))
        t = timeit(lambda: ops.gemm(dz, w0, ws, trans_b=True, out=dx))
        print("width %d  dX_0  [B,512]x[512,%d]     %7.3f ms %6.1f TF" % (W, W, t, fl / t / 1e9))
        for sk in (0, 4, 8):
            t = timeit(lambda: ops.gemm(x, dz, ws, trans_a=True, out=dw, b_colsum=db, split_k=sk))
            print("width %d  dW_0  [%d,B]x[B,512] split %d  %7.3f ms %6.1f TF" % (W, W, sk, t, fl / t / 1e9))


if __name__ == "__main__":
    main()
