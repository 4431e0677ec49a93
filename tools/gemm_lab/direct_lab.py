#!/usr/bin/env python3
"""Lab for csrc/gemm_direct.h: builds the kernel alone with -DREC_DIRECT_PF=<n> (and any further -D given) and times the three
GEMM shapes of a DeepFM step at the reference's batch size 512 (forward, dX, dW), 300 launches per HIP-event pair, against
float64.    python tools/gemm_lab/direct_lab.py 8 16 28 ["-DREC_DIRECT_KSPLIT=4 ..."]"""
import ctypes as C
import os
import subprocess
import sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
variants = sys.argv[1:] or ["8"]
dev = "cuda"
rng = np.random.default_rng(0)
shapes = [("fwd  512x400x432", 512, 400, 432, 0, 0, 2), ("dX   512x432x400", 512, 432, 400, 0, 1, 3),
          ("dW   432x400x512", 432, 400, 512, 1, 0, 0), ("din  32x80x512", 32, 80, 512, 0, 0, 2),
          ("floor 512x400x16", 512, 400, 16, 0, 0, 2), ("floor 32x80x16", 32, 80, 16, 0, 0, 2), ("K128 512x400x128", 512, 400, 128, 0, 0, 2),
          ("K256 32x80x256", 32, 80, 256, 0, 0, 2), ("K1024 32x80x1024", 32, 80, 1024, 0, 0, 2)]
for v in variants:
    defs = ["-DREC_DIRECT_PF=" + v] if v.isdigit() else v.split()
    so = os.path.join("/tmp", "direct_%s.so" % "_".join(x.replace("-D", "").replace("=", "") for x in defs))
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I" + REPO + "/include",
                    "-I" + REPO + "/paddlerec_amd/csrc"] + defs + [os.path.join(HERE, "direct_lab.hip"), "-o", so], check=True)
    L = C.CDLL(so)
    L.lab_direct.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 7
    for name, M, N, K, ta, tb, epi in shapes:
        A, B = rng.uniform(-1, 1, (M, K)).astype(np.float32), rng.uniform(-1, 1, (K, N)).astype(np.float32)
        bias, X0 = rng.uniform(-1, 1, N).astype(np.float32), rng.uniform(-1, 1, (M, N)).astype(np.float32)
        At = torch.as_tensor(np.ascontiguousarray(A.T if ta else A)).to(dev)
        Bt = torch.as_tensor(np.ascontiguousarray(B.T if tb else B)).to(dev)
        bt, xt = torch.as_tensor(bias).to(dev), torch.as_tensor(X0).to(dev)
        Ct, cs = torch.empty(M, N, device=dev), torch.empty(N, device=dev)
        st = torch.cuda.current_stream().cuda_stream

        def run():
            return L.lab_direct(M, N, K, ta, tb, epi, At.data_ptr(), Bt.data_ptr(), Ct.data_ptr(), bt.data_ptr(), xt.data_ptr(),
                                cs.data_ptr() if ta else None, st)
        assert run() == 0
        acc = A.astype(np.float64) @ B.astype(np.float64)
        want = np.maximum(acc + bias, 0) if epi == 2 else np.where(X0 > 0, acc, 0) if epi == 3 else acc
        err = np.abs(Ct.cpu().numpy() - want).max()
        bound = (4e-7 * (np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64))).max()
        if ta:
            err = max(err, np.abs(cs.cpu().numpy() - B.astype(np.float64).sum(0)).max() * 1e-2)
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300):
            run()
        e1.record()
        torch.cuda.synchronize()
        print("%-28s %-18s %6.2f us per launch   max err %.2e (bound %.2e)" % (" ".join(defs), name, e0.elapsed_time(e1) / 300 * 1e3, err, bound))
