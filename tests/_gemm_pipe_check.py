"""Run with REC_GEMM_PIPE=1 (tests/test_gemm_gpu.py::test_gemm_pipe_kernel does, in a subprocess — the flag is read once per
process): the opt-in interior-only kernel of csrc/gemm_f32.hip on whole-tile shapes, every epilogue it is built for, against
a float64 reference.  Prints "ok" and the worst relative error."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from paddlerec_amd import ops  # noqa: E402

assert os.environ.get("REC_GEMM_PIPE") == "1"
g = torch.Generator(device="cuda").manual_seed(3)
ws = ops.Workspace("cuda")
worst = 0.0


def rnd(*shape):
    return torch.rand(*shape, device="cuda", generator=g) * 2 - 1


def check(got, ref, what):
    global worst
    err = float((got.double() - ref).abs().max()) / max(float(ref.abs().max()), 1e-30)
    worst = max(worst, err)
    assert err < (3e-5 if what[-1] > 10000 else 1e-5), (what, err)      # f32 sums over K = 65536 reach 1e-5 by themselves


for M, N, K in ((8192, 400, 416), (8192, 160, 16), (16384, 80, 48), (8192, 400, 400)):      # 256x80 tile: tall, N % 80 == 0
    A, W, b = rnd(M, K), rnd(K, N), rnd(N)
    ref = A.double() @ W.double()
    check(ops.gemm(A, W, ws), ref, ("none", M, N, K))
    check(ops.gemm(A, W, ws, epilogue="bias", bias=b), ref + b.double(), ("bias", M, N, K))
    check(ops.gemm(A, W, ws, epilogue="bias_relu", bias=b), (ref + b.double()).clamp_min(0), ("bias_relu", M, N, K))
    Wt, act = rnd(N, K), rnd(M, N)                                         # dX = (dY W^T) * (act > 0), W given as [N, K]
    check(ops.gemm(A, Wt, ws, trans_b=True, epilogue="relu_mask", aux0=act),
          (A.double() @ Wt.double().t()) * (act > 0), ("relu_mask", M, N, K))
# 80x80 tile, split-K, column sums; M = 432 / 144 / 1008: the 144x80 tile (9 waves: DeepFM's layer-0 weight gradient)
for M, N, K in ((400, 400, 8192), (80, 160, 4096), (400, 80, 65536), (240, 400, 1600), (432, 400, 65536), (144, 80, 160),
                (1008, 160, 4096)):
    X, dY = rnd(K, M), rnd(K, N)
    dW, db = torch.empty(M, N, device="cuda"), torch.empty(N, device="cuda")
    ops.gemm(X, dY, ws, trans_a=True, out=dW, b_colsum=db)
    check(dW, X.double().t() @ dY.double(), ("dW", M, N, K))
    check(db, dY.double().sum(0), ("db", M, N, K))
    for split in (1, 3, 8):
        check(ops.gemm(X, dY, ws, trans_a=True, split_k=split), X.double().t() @ dY.double(), ("dW split", split, M, N, K))
torch.cuda.synchronize()
print("ok worst relative error %.2e" % worst)
