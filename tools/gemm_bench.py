#!/usr/bin/env python3
"""rec_gemm_f32 vs the vendor GEMM (torch.mm -> hipBLASLt) on the DeepFM / DCN-v2 MLP shapes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from paddlerec_amd import ops  # noqa: E402

DEV = "cuda"


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


ws = ops.Workspace(DEV)
B = 65536
shapes = [("fwd L0  [B,624]x[624,400]", B, 400, 624, False, False),
          ("fwd L1  [B,400]x[400,400]", B, 400, 400, False, False),
          ("dX  L0  [B,400]x[624,400]^T", B, 624, 400, False, True),
          ("dW  L0  [B,624]^T x[B,400]", 624, 400, B, True, False),
          ("dW  L1  [B,400]^T x[B,400]", 400, 400, B, True, False),
          ("cross   [B,1560]x[1560,1560]", B, 1560, 1560, False, False)]
for name, M, N, K, ta, tb in shapes:
    A = torch.randn((K, M) if ta else (M, K), device=DEV)
    Bm = torch.randn((N, K) if tb else (K, N), device=DEV)
    out = torch.empty(M, N, device=DEV)
    t_rec = timeit(lambda: ops.gemm(A, Bm, ws, trans_a=ta, trans_b=tb, out=out))
    Ae = A.t() if ta else A
    Be = Bm.t() if tb else Bm
    t_ven = timeit(lambda: torch.mm(Ae, Be, out=out))
    fl = 2.0 * M * N * K
    print("%-32s rec %.3f ms (%.1f TF)   vendor %.3f ms (%.1f TF)" %
          (name, t_rec, fl / t_rec / 1e9, t_ven, fl / t_ven / 1e9))
