"""f32 MFMA GEMM (rec_gemm_f32) and column sums against float64 NumPy.

The instruction is an exact k-ordered f32 fmaf chain; against float64 the error is f32 round-off of a
K-term dot product, bounded here by 4e-7 * sum_k |a_ik||b_kj| (the guide measures 0.75-1.5e-7 at K<=1024).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops(engine_lib):
    from paddlerec_amd import ops as o
    return o


def _mk(rng, *shape):
    return rng.uniform(-1, 1, size=shape).astype(np.float32)


def _check(C, want, bound):
    err = np.abs(C.astype(np.float64) - want)
    assert np.all(err <= bound + 1e-30), "max err %.3e, bound %.3e" % (err.max(), bound.max())


@pytest.mark.parametrize("M,N,K,ta,tb", [
    (128, 128, 16, False, False), (256, 400, 624, False, False), (1000, 400, 400, False, False),
    (300, 1, 400, False, False), (129, 81, 17, False, False), (1, 5, 3, False, False),
    (624, 400, 3000, True, False), (400, 1, 999, True, False), (513, 624, 400, False, True),
    (200, 1560, 1560, False, False), (77, 130, 50, True, True), (64, 256, 40, False, True),
    (5000, 1, 400, False, False), (777, 3, 401, False, True), (400, 1, 5000, True, False),
    (1000, 4, 3000, True, False), (33, 2, 7, False, False)])
def test_gemm_plain(ops, M, N, K, ta, tb):
    # A=I-style asymmetry is covered by random asymmetric operands: a row/col swap cannot pass
    rng = np.random.default_rng(M + N + K)
    A, B = _mk(rng, M, K), _mk(rng, K, N)
    At = torch.as_tensor(np.ascontiguousarray(A.T if ta else A)).to(DEV)
    Bt = torch.as_tensor(np.ascontiguousarray(B.T if tb else B)).to(DEV)
    ws = ops.Workspace(DEV)
    C = ops.gemm(At, Bt, ws, trans_a=ta, trans_b=tb).cpu().numpy()
    want = A.astype(np.float64) @ B.astype(np.float64)
    bound = 4e-7 * (np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64))
    _check(C, want, bound)
    # deterministic (split-K partials are reduced in a fixed order)
    C2 = ops.gemm(At, Bt, ws, trans_a=ta, trans_b=tb).cpu().numpy()
    assert np.array_equal(C, C2)


@pytest.mark.parametrize("split", [1, 2, 7, 64])
def test_gemm_split_k(ops, split):
    rng = np.random.default_rng(split)
    M, N, K = 100, 90, 2000
    A, B = _mk(rng, K, M), _mk(rng, K, N)          # trans_a form: dW = X^T G
    ws = ops.Workspace(DEV)
    C = ops.gemm(torch.as_tensor(A).to(DEV), torch.as_tensor(B).to(DEV), ws, trans_a=True,
                 split_k=split).cpu().numpy()
    want = A.astype(np.float64).T @ B.astype(np.float64)
    _check(C, want, 4e-7 * (np.abs(A).astype(np.float64).T @ np.abs(B).astype(np.float64)))


def test_gemm_epilogues(ops):
    rng = np.random.default_rng(3)
    M, N, K = 300, 400, 200
    A, B, bias = _mk(rng, M, K), _mk(rng, K, N) * 0.1, _mk(rng, N)
    X0, Xl = _mk(rng, M, N), _mk(rng, M, N)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(DEV)
    ws = ops.Workspace(DEV)
    acc = A.astype(np.float64) @ B.astype(np.float64)
    tol = dict(rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ops.gemm(t(A), t(B), ws, epilogue="bias", bias=t(bias)).cpu().numpy(),
                               acc + bias, **tol)
    np.testing.assert_allclose(ops.gemm(t(A), t(B), ws, epilogue="bias_relu", bias=t(bias)).cpu().numpy(),
                               np.maximum(acc + bias, 0), **tol)
    np.testing.assert_allclose(ops.gemm(t(A), t(B), ws, epilogue="relu_mask", aux0=t(X0)).cpu().numpy(),
                               np.where(X0 > 0, acc, 0), **tol)
    np.testing.assert_allclose(                                   # dcn_v2/net.py:225
        ops.gemm(t(A), t(B), ws, epilogue="cross", bias=t(bias), aux0=t(X0), aux1=t(Xl)).cpu().numpy(),
        Xl + X0 * (acc + bias), **tol)
    np.testing.assert_allclose(ops.gemm(t(A), t(B), ws, epilogue="bias_sigmoid", bias=t(bias)).cpu().numpy(),
                               1 / (1 + np.exp(-(acc + bias))), **tol)
    np.testing.assert_allclose(ops.gemm(t(A), t(B), ws, epilogue="bias_tanh", bias=t(bias)).cpu().numpy(),
                               np.tanh(acc + bias), **tol)
    np.testing.assert_allclose(ops.gemm(t(A), t(B), ws, epilogue="add", aux1=t(Xl)).cpu().numpy(),
                               acc + Xl, **tol)
    # epilogue after a split-K reduction
    np.testing.assert_allclose(
        ops.gemm(t(A), t(B), ws, epilogue="bias_relu", bias=t(bias), split_k=4).cpu().numpy(),
        np.maximum(acc + bias, 0), **tol)


@pytest.mark.parametrize("M,N,K,ta,tb", [
    (512, 400, 432, False, False),      # deepfm/config_bigdata.yaml bs 512: layer 0 forward
    (512, 432, 400, False, True),       # ... its dX
    (432, 400, 512, True, False),       # ... its dW (K = the batch), with the bias gradient
    (32, 80, 512, False, False),        # din/config.yaml bs 32: attention MLP
    (80, 40, 32, True, False), (513, 401, 37, False, True), (17, 6, 1000, True, True), (1, 400, 400, False, False)])
def test_gemm_direct_kernel(ops, M, N, K, ta, tb):
    """csrc/gemm_direct.h: the launch-bound sizes in ONE launch, a wave per 16 x 16 tile over the whole K — against float64
    with the same bound as every other exact-f32 path, bit-reproducible, and (trans_a) the fused column sums."""
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    A, B = _mk(rng, M, K), _mk(rng, K, N)
    At = torch.as_tensor(np.ascontiguousarray(A.T if ta else A)).to(DEV)
    Bt = torch.as_tensor(np.ascontiguousarray(B.T if tb else B)).to(DEV)
    ws = ops.Workspace(DEV)
    kw = {}
    if ta and not tb:
        kw["b_colsum"] = torch.empty(N, device=DEV)
    C = ops.gemm(At, Bt, ws, trans_a=ta, trans_b=tb, **kw).cpu().numpy()
    want = A.astype(np.float64) @ B.astype(np.float64)
    bound = 4e-7 * (np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64))
    _check(C, want, bound)
    if kw:
        np.testing.assert_allclose(kw["b_colsum"].cpu().numpy(), B.astype(np.float64).sum(0), rtol=1e-5,
                                   atol=1e-6 * np.abs(B).sum(0).max())
    assert np.array_equal(C, ops.gemm(At, Bt, ws, trans_a=ta, trans_b=tb, **kw).cpu().numpy())
    # an explicit K split takes the tiled kernels + the reduce launch: same result to the bound of both
    C2 = ops.gemm(At, Bt, ws, trans_a=ta, trans_b=tb, split_k=2).cpu().numpy()
    _check(C2, want, bound)
    # strided operands (a column slice of a wider buffer): the float4 path must not be taken on unaligned rows
    if not ta and not tb and K % 4 == 0 and M > 1:
        wide = torch.zeros(M, K + 3, device=DEV)
        wide[:, 1:K + 1] = At
        C3 = ops.gemm(wide[:, 1:K + 1], Bt, ws).cpu().numpy()
        assert np.array_equal(C3, C)


@pytest.mark.parametrize("B,nin,nout,epi", [
    (512, 400, 400, "relu_mask"), (512, 432, 400, "none"),      # deepfm bs 512: hidden layer, layer 0 (one launch for both)
    (32, 80, 40, "dsigmoid"), (32, 256, 80, "none"),            # din bs 32 (K split over the workgroup)
    (512, 1560, 1560, "relu_mask"),                             # not launch-bound: two rec_gemm_f32 calls
    (300, 130, 77, "relu_mask")])                               # odd sizes, unaligned rows of W: falls back as well
def test_linear_backward_pair(ops, B, nin, nout, epi):
    """rec_gemm_f32_pair (ops.linear_backward): dW = X^T G + db and dX = G W^T (+ activation') of one Linear in one call —
    bit-identical to the two rec_gemm_f32 calls whether it goes out as one launch or two, and right against float64."""
    rng = np.random.default_rng(B + nin + nout)
    X, G, W = _mk(rng, B, nin), _mk(rng, B, nout), _mk(rng, nin, nout)
    Xt, Gt, Wt = (torch.as_tensor(a).to(DEV) for a in (X, G, W))
    ws = ops.Workspace(DEV)
    dW, db = torch.empty(nin, nout, device=DEV), torch.empty(nout, device=DEV)
    aux = Xt if epi != "none" else None
    dX = ops.linear_backward(Xt, Gt, Wt, ws, dW, db, epilogue=epi, aux0=aux)
    dW2, db2 = torch.empty_like(dW), torch.empty_like(db)
    ops.gemm(Xt, Gt, ws, trans_a=True, out=dW2, b_colsum=db2)
    dX2 = ops.gemm(Gt, Wt, ws, trans_b=True, epilogue=epi, **(dict(aux0=aux) if aux is not None else {}))
    assert torch.equal(dW, dW2) and torch.equal(db, db2) and torch.equal(dX, dX2)
    acc = G.astype(np.float64) @ W.astype(np.float64).T
    want = np.where(X > 0, acc, 0) if epi == "relu_mask" else acc * X * (1 - X) if epi == "dsigmoid" else acc
    scale = 2.5 if epi == "dsigmoid" else 1.0          # |x (1 - x)| <= 2 on [-1, 1], plus the rounding of the product
    _check(dX.cpu().numpy(), want, scale * 4e-7 * (np.abs(G).astype(np.float64) @ np.abs(W).astype(np.float64).T) + 1e-12)
    _check(dW.cpu().numpy(), X.astype(np.float64).T @ G.astype(np.float64),
           4e-7 * (np.abs(X).astype(np.float64).T @ np.abs(G).astype(np.float64)))


def test_gemm_skinny_paths(ops):
    """One-output Linear layers (N <= 4) take the streaming kernels: forward with epilogues, dW with the fused
    bias gradient."""
    rng = np.random.default_rng(8)
    M, K = 3000, 400
    A, w, b = _mk(rng, M, K), _mk(rng, K, 1) * 0.1, _mk(rng, 1)
    item_b = _mk(rng, M, 1)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(DEV)
    ws = ops.Workspace(DEV)
    acc = A.astype(np.float64) @ w.astype(np.float64)
    tol = dict(rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ops.gemm(t(A), t(w), ws, epilogue="bias", bias=t(b)).cpu().numpy(), acc + b, **tol)
    np.testing.assert_allclose(ops.gemm(t(A), t(w), ws, epilogue="add", bias=t(b), aux1=t(item_b)).cpu().numpy(),
                               acc + b + item_b, **tol)
    np.testing.assert_allclose(ops.gemm(t(A), t(w), ws, epilogue="bias_sigmoid", bias=t(b)).cpu().numpy(),
                               1 / (1 + np.exp(-(acc + b))), **tol)
    G = _mk(rng, M, 1)
    db = torch.empty(1, device=DEV)
    dW = ops.gemm(t(A), t(G), ws, trans_a=True, b_colsum=db)
    np.testing.assert_allclose(dW.cpu().numpy(), A.astype(np.float64).T @ G.astype(np.float64), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(db.cpu().numpy(), G.astype(np.float64).sum(0), rtol=1e-5, atol=1e-4)
    dW2 = ops.gemm(t(A), t(G), ws, trans_a=True, b_colsum=db)
    assert torch.equal(dW, dW2)
    # narrow inputs (DIN's 40-wide head): thread groups share a block's k rows (skinny_dw_narrow_kernel)
    for Kn, Mn, Nn in ((4096, 40, 1), (5001, 128, 3), (1024, 1, 1), (2500, 100, 4), (3000, 129, 2)):
        A2, G2 = _mk(rng, Kn, Mn), _mk(rng, Kn, Nn)
        db2 = torch.empty(Nn, device=DEV)
        got = ops.gemm(t(A2), t(G2), ws, trans_a=True, b_colsum=db2)
        np.testing.assert_allclose(got.cpu().numpy(), A2.astype(np.float64).T @ G2.astype(np.float64), rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(db2.cpu().numpy(), G2.astype(np.float64).sum(0), rtol=1e-5, atol=1e-4)
        assert torch.equal(got, ops.gemm(t(A2), t(G2), ws, trans_a=True, b_colsum=db2))


def test_gemm_strided_views_and_inplace_out(ops):
    rng = np.random.default_rng(4)
    big = torch.as_tensor(_mk(rng, 200, 96)).to(DEV)
    A = big[:, 8:72]                                   # row stride 96, 16-B aligned start
    A1 = big[:, 1:65]                                  # unaligned start -> scalar load path
    B = torch.as_tensor(_mk(rng, 64, 48)).to(DEV)
    ws = ops.Workspace(DEV)
    outbuf = torch.zeros(200, 64, device=DEV)
    for a in (A, A1):
        C = ops.gemm(a, B, ws, out=outbuf[:, :48])
        want = a.double().cpu().numpy() @ B.double().cpu().numpy()
        np.testing.assert_allclose(C.cpu().numpy(), want, rtol=1e-5, atol=1e-5)
    assert float(outbuf[:, 48:].abs().max()) == 0.0   # nothing written outside the N columns


def test_gemm_fused_colsum(ops):
    """dW = X^T G with db = colsum(G) from the same pass (bias gradient of a Linear)."""
    rng = np.random.default_rng(6)
    for Bsz, nin, nout in ((3000, 624, 400), (777, 40, 1), (5000, 130, 200), (100, 400, 400)):
        X, G = _mk(rng, Bsz, nin), _mk(rng, Bsz, nout)
        ws = ops.Workspace(DEV)
        db = torch.empty(nout, device=DEV)
        dW = ops.gemm(torch.as_tensor(X).to(DEV), torch.as_tensor(G).to(DEV), ws, trans_a=True, b_colsum=db)
        np.testing.assert_allclose(dW.cpu().numpy(), X.astype(np.float64).T @ G.astype(np.float64),
                                   rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(db.cpu().numpy(), G.astype(np.float64).sum(0), rtol=1e-5,
                                   atol=1e-6 * np.abs(G).sum(0).max())


def test_colsum(ops):
    rng = np.random.default_rng(5)
    for M, N in ((1, 3), (513, 400), (70000, 400), (1000, 1)):
        G = _mk(rng, M, N)
        got = ops.colsum(torch.as_tensor(G).to(DEV), ops.Workspace(DEV)).cpu().numpy()
        np.testing.assert_allclose(got, G.astype(np.float64).sum(0), rtol=1e-5,
                                   atol=1e-6 * np.abs(G).sum(0).max())


def test_gemm_argument_errors(ops):
    ws = ops.Workspace(DEV)
    a = torch.zeros(4, 5, device=DEV)
    with pytest.raises(Exception, match="inner"):
        ops.gemm(a, torch.zeros(6, 3, device=DEV), ws)
    with pytest.raises(Exception, match="bias"):
        ops.gemm(a, torch.zeros(5, 3, device=DEV), ws, epilogue="bias")
    with pytest.raises(Exception, match="device tensor"):
        ops.gemm(torch.zeros(4, 5), torch.zeros(5, 3), ws)


def test_stream_helpers(ops):
    """rec_stream_spin / concurrent_stream / cu_range_stream: the helpers the train steps place their side work with."""
    import ctypes as C
    import time
    from paddlerec_amd._lib import lib
    main = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    assert lib().rec_stream_spin(2000, C.c_void_p(main.cuda_stream)) == 0
    e1.record()
    torch.cuda.synchronize()
    assert 0.2 <= e0.elapsed_time(e1) <= 200.0                   # the spin really holds the stream (~2 ms)
    s = ops.concurrent_stream(DEV)
    assert isinstance(s, torch.cuda.Stream) and s.cuda_stream != main.cuda_stream
    assert ops.concurrent_stream(DEV) is s                       # probed once per (device, main stream)
    dev = torch.device("cuda", torch.cuda.current_device())
    cs = ops.cu_range_stream(dev, 0, 64)
    assert ops.cu_range_stream(dev, 0, 64) is cs
    with torch.cuda.stream(cs):                                  # kernels run (and finish) on a CU-restricted stream
        a = torch.ones(1 << 20, device=DEV)
        out = ops.colsum(a.view(-1, 1), ops.Workspace(DEV))
    cs.synchronize()
    assert float(out.item()) == float(1 << 20)
    sp = ops.gemm(torch.ones(8, 4, device=DEV), torch.ones(8, 3, device=DEV), ops.Workspace(DEV), trans_a=True, num_cus=64)
    assert torch.equal(sp.cpu(), torch.full((4, 3), 8.0))


def test_gemm_pipe_kernel():
    """The opt-in interior-only kernel (REC_GEMM_PIPE=1, read once per process -> a subprocess): whole-tile shapes of the
    256x80 and 80x80 tiles, the four epilogues it is built for, split-K and column sums, against float64."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, REC_GEMM_PIPE="1")
    r = subprocess.run([sys.executable, os.path.join(here, "_gemm_pipe_check.py")], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and "ok worst relative error" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("M,N,K,tb", [
    (8192, 400, 416, False), (8192, 400, 400, True), (8192, 416, 400, True), (8448, 80, 32, False),
    (8192, 416, 48, False), (16384, 160, 1008, True), (8192, 400, 16 * 3, False), (8192, 208, 32, True),
    (8192, 432, 400, True), (8320, 144, 64, False), (66560, 432, 400, True)])
def test_gemm_glds_kernel(ops, M, N, K, tb, monkeypatch):
    """The LDS-DMA kernel (csrc/gemm_glds.h: 3-stage ring, counted vmcnt, one raw barrier per k-step) takes the tall
    whole-tile problems of the MLP chain: both B forms ([K,N] rotated-row image, [N,K] swizzled b128 image), both block
    shapes (256x80, 128x208), K from 2 to 63 tiles (ring wrap-around, 1-2 tile tails), the four epilogues — against
    float64, and bit-identical across two launches and against REC_GEMM_GLDS=0 up to the fp32 bound."""
    monkeypatch.setenv("REC_GEMM_GLDS_80", "1")       # the 256x80 configuration too (off by default: no faster)
    rng = np.random.default_rng(M + N * 3 + K)
    A, B = _mk(rng, M, K), _mk(rng, K, N)
    bias, X0 = _mk(rng, N), _mk(rng, M, N)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(DEV)
    At, Bt = t(A), t(B.T if tb else B)
    ws = ops.Workspace(DEV)
    want = A.astype(np.float64) @ B.astype(np.float64)
    bound = 4e-7 * (np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64))
    C = ops.gemm(At, Bt, ws, trans_b=tb).cpu().numpy()
    _check(C, want, bound)
    assert np.array_equal(C, ops.gemm(At, Bt, ws, trans_b=tb).cpu().numpy())
    Cb = ops.gemm(At, Bt, ws, trans_b=tb, epilogue="bias", bias=t(bias)).cpu().numpy()
    _check(Cb, want + bias, bound + 1e-6)
    Cr = ops.gemm(At, Bt, ws, trans_b=tb, epilogue="bias_relu", bias=t(bias)).cpu().numpy()
    _check(Cr, np.maximum(want + bias, 0), bound + 1e-6)
    Cm = ops.gemm(At, Bt, ws, trans_b=tb, epilogue="relu_mask", aux0=t(X0)).cpu().numpy()
    _check(Cm, np.where(X0 > 0, want, 0), bound)
    # strided views (row strides larger than the logical width) take the same kernel
    wide = torch.zeros(M, K + 16, device=DEV)
    wide[:, :K] = At
    _check(ops.gemm(wide[:, :K], Bt, ws, trans_b=tb).cpu().numpy(), want, bound)


@pytest.mark.parametrize("M,N,K,tb", [(32768, 400, 400, False), (32768, 400, 432, False), (32768, 400, 400, True),
                                      (32768, 432, 400, True), (65536 + 64 * 40, 400, 48, False)])
def test_gemm_panel_kernel(ops, M, N, K, tb, monkeypatch):
    """The row-panel kernel (csrc/gemm_panel.h: a block owns 64 rows x ALL N columns, persistent over panels, permuted
    accumulator columns, float4 epilogue) is an opt-in experiment (REC_GEMM_PANEL=1; no faster on random data, see
    profiles/r04_gemm_power.txt): against float64 and BIT-IDENTICAL to the tiled kernels (same order of additions), the
    four epilogues, both B forms, a panel count that is not a multiple of the grid."""
    rng = np.random.default_rng(M + N * 3 + K)
    A, B = _mk(rng, M, K), _mk(rng, K, N)
    bias, X0 = _mk(rng, N), _mk(rng, M, N)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(DEV)
    At, Bt = t(A), t(B.T if tb else B)
    ws = ops.Workspace(DEV)
    want = A.astype(np.float64) @ B.astype(np.float64)
    bound = 4e-7 * (np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64))
    for kw, ref, extra in ((dict(), want, 0.0), (dict(epilogue="bias", bias=t(bias)), want + bias, 1e-6),
                           (dict(epilogue="bias_relu", bias=t(bias)), np.maximum(want + bias, 0), 1e-6),
                           (dict(epilogue="relu_mask", aux0=t(X0)), np.where(X0 > 0, want, 0), 0.0)):
        monkeypatch.setenv("REC_GEMM_PANEL", "1")
        Cp = ops.gemm(At, Bt, ws, trans_b=tb, **kw).cpu().numpy()
        monkeypatch.setenv("REC_GEMM_PANEL", "0")
        Ct = ops.gemm(At, Bt, ws, trans_b=tb, **kw).cpu().numpy()
        _check(Cp, ref, bound + extra)
        assert np.array_equal(Cp, Ct), kw.get("epilogue", "none")


@pytest.mark.parametrize("B,N,ld", [(64, 400, 400), (1000, 400, 400), (65536, 400, 400), (4099, 16, 16),
                                    (2048, 512, 512), (777, 128, 132)])
def test_mlp_head_bwd(ops, B, N, ld):
    """Fused backward of a one-logit head (rec_mlp_head_bwd) against float64, and bit-identical on a re-run."""
    rng = np.random.default_rng(B + N)
    act = np.maximum(_mk(rng, B, ld), 0)                       # a ReLU output: about half zeros
    dz, w = _mk(rng, B, 1) * 1e-3, _mk(rng, N, 1)
    ws = ops.Workspace(DEV)
    a_t = torch.as_tensor(act).to(DEV)[:, :N]
    dz_t, w_t = torch.as_tensor(dz).to(DEV), torch.as_tensor(w).to(DEV)
    runs = []
    for _ in range(2):
        dw, db = torch.full((N, 1), 7.0, device=DEV), torch.full((1,), 7.0, device=DEV)
        dx = ops.mlp_head_bwd(a_t, dz_t, w_t, ws, dw, db, relu=True)
        runs.append((dx.cpu().numpy(), dw.cpu().numpy(), db.cpu().numpy()))
    dx, dw, db = runs[0]
    a64 = act[:, :N].astype(np.float64)
    want_dx = (dz.astype(np.float32) * w.reshape(1, N).astype(np.float32)) * (act[:, :N] > 0)     # one fp32 product
    np.testing.assert_array_equal(dx, want_dx.astype(np.float32))
    want_dw = a64.T @ dz.astype(np.float64)
    bound = 1e-6 * (np.abs(a64).T @ np.abs(dz.astype(np.float64))).max() + 1e-30
    assert np.abs(dw - want_dw).max() <= bound
    assert abs(db[0] - dz.astype(np.float64).sum()) <= 1e-6 * np.abs(dz).sum()
    for x, y in zip(runs[0], runs[1]):
        np.testing.assert_array_equal(x, y)
    # relu = 0: no mask
    dw, db = torch.empty(N, 1, device=DEV), torch.empty(1, device=DEV)
    dx2 = ops.mlp_head_bwd(a_t, dz_t, w_t, ws, dw, db, relu=False).cpu().numpy()
    np.testing.assert_array_equal(dx2, (dz * w.reshape(1, N)).astype(np.float32))


def test_mlp_backward_head_path_matches_gemm_path(ops, monkeypatch):
    """mlp_backward with the fused head and with the GEMM chain (REC_MLP_HEAD_FUSED=0): same gradients to 1e-5 of scale."""
    rng = np.random.default_rng(11)
    B, sizes = 4096, [208, 400, 400, 1]
    ws = ops.Workspace(DEV)
    Ws = [torch.as_tensor(_mk(rng, sizes[i], sizes[i + 1]) * 0.1).to(DEV) for i in range(3)]
    bs = [torch.as_tensor(_mk(rng, sizes[i + 1]) * 0.1).to(DEV) for i in range(3)]
    x = torch.as_tensor(_mk(rng, B, 208)).to(DEV)
    dy = torch.as_tensor(_mk(rng, B, 1) * 1e-3).to(DEV)
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("REC_MLP_HEAD_FUSED", flag)
        _, acts = ops.mlp_forward(x, Ws, bs, ws)
        dws = [torch.empty_like(w) for w in Ws]
        dbs = [torch.empty_like(b) for b in bs]
        dx = ops.mlp_backward(dy, acts, Ws, dws, dbs, ws)
        outs.append([dx.cpu().numpy()] + [t.cpu().numpy() for t in dws + dbs])
    for a, b in zip(*outs):
        assert np.abs(a - b).max() <= 1e-5 * np.abs(b).max()


@pytest.mark.parametrize("M,N,K,tb,epi", [
    (8192, 400, 400, False, "bias_relu"), (8192, 400, 432, False, "bias_relu"), (8192, 400, 400, True, "relu_mask"),
    (8200, 396, 104, False, "bias"), (9000, 416, 72, True, "none"), (8192, 368, 400, False, "none"),
    # column blocks: 432 = two blocks of 14 tiles, 512 = two of 16, 1560 = four of 26; the CrossNet epilogues
    (8192, 432, 400, True, "none"), (8192, 512, 200, False, "bias_relu"), (8192, 1560, 136, False, "cross"),
    (8192, 1560, 72, True, "add"), (8320, 768, 1560, False, "bias_relu"),
    # CrossNetMix (dcn_v2/net.py:278-320): low-rank projections (N 256 = one block of 16 tiles), the mixture epilogue
    (8192, 256, 1560, False, "bias_tanh"), (8192, 1560, 256, True, "moe"), (8192, 256, 264, False, "dtanh"),
    (8192, 400, 72, False, "bias_sigmoid"), (8192, 400, 72, True, "dsigmoid")])
def test_gemm_bf16x3(ops, monkeypatch, M, N, K, tb, epi):
    """gemm_bf16x3.h (REC_GEMM_BF16X3=1): f32 operands split exactly into three bf16 terms, six bf16 MFMAs per product,
    f32 accumulate.  Same float64 bar as the exact-f32 kernels (4e-7 of sum |a||b| per output), rows behind a multiple of
    128, a K tail behind a multiple of 32, both weight orientations, the four epilogues; and the switch really switches
    (the results differ from the exact-f32 kernels' in the last bits, both inside the bound)."""
    rng = np.random.default_rng(M + N + K)
    A, B, bias, X0 = _mk(rng, M, K), _mk(rng, K, N), _mk(rng, N), _mk(rng, M, N)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(DEV)
    At, Bt = t(A), t(B.T if tb else B)
    X1 = _mk(rng, M, N)
    rs = _mk(rng, M)
    kw = dict(trans_b=tb, epilogue=epi, bias=t(bias) if epi.startswith("bias") or epi in ("cross", "add", "moe") else None,
              aux0=t(X0) if epi in ("relu_mask", "cross", "add", "moe", "dtanh", "dsigmoid") else None,
              aux1=t(X1) if epi in ("cross", "add", "moe") else None, row_scale=t(rs) if epi == "moe" else None)
    monkeypatch.setenv("REC_GEMM_BF16X3", "0")
    C0 = ops.gemm(At, Bt, ops.Workspace(DEV), **kw).cpu().numpy()
    monkeypatch.setenv("REC_GEMM_BF16X3", "1")
    ws = ops.Workspace(DEV)
    C1 = ops.gemm(At, Bt, ws, **kw).cpu().numpy()
    acc = A.astype(np.float64) @ B.astype(np.float64)
    bound = 4e-7 * (np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64))
    if epi.startswith("bias"):
        acc = acc + bias
        bound = bound + 1.2e-7 * np.abs(acc)                 # the f32 add of the bias
    if epi == "bias_relu":
        acc = np.maximum(acc, 0)
    if epi == "relu_mask":
        acc = np.where(X0 > 0, acc, 0)
    if epi == "cross":                                       # dcn_v2/net.py:225: X_l + X_0 * (X_l W + b)
        acc = X1 + X0.astype(np.float64) * (acc + bias)
        bound = bound * np.abs(X0) + 2.4e-7 * (np.abs(acc) + np.abs(X1)) + 1e-7
    if epi == "add":
        acc = acc + bias + X1.astype(np.float64) + X0
        bound = bound + 3.6e-7 * (np.abs(acc) + np.abs(X1) + np.abs(X0)) + 1e-7
    if epi == "moe":                                         # x_l + x_0 * gate_e * (U_e v + b)
        acc = X1 + X0.astype(np.float64) * rs[:, None] * (acc + bias)
        bound = bound * np.abs(X0 * rs[:, None]) + 3.6e-7 * (np.abs(acc) + np.abs(X1)) + 1e-7
    if epi in ("bias_tanh", "bias_sigmoid"):                 # expf / tanhf of the device: a few ulp of the result
        acc = np.tanh(acc) if epi == "bias_tanh" else 1.0 / (1.0 + np.exp(-acc))
        bound = bound + 1e-6
    if epi == "dtanh":
        acc = acc * (1.0 - X0.astype(np.float64) ** 2)
        bound = bound * np.abs(1.0 - X0.astype(np.float64) ** 2) + 2.4e-7 * np.abs(acc) + 1e-7
    if epi == "dsigmoid":
        acc = acc * X0.astype(np.float64) * (1.0 - X0)
        bound = bound * np.abs(X0 * (1.0 - X0)) + 3.6e-7 * np.abs(acc) + 1e-7
    _check(C1, acc, bound)
    _check(C0, acc, bound)
    assert not np.array_equal(C0, C1), "REC_GEMM_BF16X3=1 did not select the bf16 x 3 kernel"
    assert np.array_equal(C1, ops.gemm(At, Bt, ws, **kw).cpu().numpy())          # deterministic
    # an operand with a wide dynamic range (products of very different magnitude in one sum): every element is split to
    # 2^-25 of ITS OWN magnitude whatever its exponent, so the result is as close to float64 as the exact-f32 kernel's
    # (whose own error on such sums is ~1.5e-6 of sum |a||b|: f32 accumulation, not the operands)
    A2 = (A * np.exp(6 * rng.standard_normal(A.shape))).astype(np.float32)
    want2 = A2.astype(np.float64) @ B.astype(np.float64)
    mag2 = np.abs(A2).astype(np.float64) @ np.abs(B).astype(np.float64)
    C2 = ops.gemm(t(A2), Bt, ws, trans_b=tb).cpu().numpy()
    monkeypatch.setenv("REC_GEMM_BF16X3", "0")
    C2f = ops.gemm(t(A2), Bt, ops.Workspace(DEV), trans_b=tb).cpu().numpy()
    e3, ef = (np.abs(C2 - want2) / mag2).max(), (np.abs(C2f - want2) / mag2).max()
    # both are ~sqrt(K) x 2^-24: the f32 roundings of a running sum that one product dominates (which of the two kernels
    # is closer depends on where that product sits in its accumulation order)
    assert e3 <= 3.0 * ef + 2e-7, (e3, ef)


@pytest.mark.parametrize("rows,kin,nout", [(8192 + 64, 400, 400), (16384, 432, 400), (8192, 336, 416), (12000 - 32, 448, 340)])
def test_gemm_bf16x3_weight_gradient(ops, monkeypatch, rows, kin, nout):
    """dW = X^T G and db = colsum(G) on the bf16 x 3 kernel (gemm_bf16x3_dw_kernel: both operands split and transposed
    on their way into LDS, K split over the chip, the engine's fixed-order reduce): float64 bound of the exact-f32 form,
    rows that do not fill the last slice, output blocks of 13 / 12 / 9 tiles, deterministic."""
    rng = np.random.default_rng(rows + kin)
    X, G = _mk(rng, rows, kin), _mk(rng, rows, nout)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(DEV)
    Xt, Gt = t(X), t(G)
    res = {}
    for v in ("0", "1"):
        monkeypatch.setenv("REC_GEMM_BF16X3", v)
        ws = ops.Workspace(DEV)
        C_, b_ = torch.zeros(kin, nout, device=DEV), torch.zeros(nout, device=DEV)
        ops.gemm(Xt, Gt, ws, trans_a=True, out=C_, b_colsum=b_)
        C2, b2 = torch.zeros(kin, nout, device=DEV), torch.zeros(nout, device=DEV)
        ops.gemm(Xt, Gt, ws, trans_a=True, out=C2, b_colsum=b2)
        assert torch.equal(C_, C2) and torch.equal(b_, b2)
        res[v] = (C_.cpu().numpy(), b_.cpu().numpy())
    want = X.astype(np.float64).T @ G.astype(np.float64)
    bound = 4e-7 * (np.abs(X).astype(np.float64).T @ np.abs(G).astype(np.float64))
    cwant, cbound = G.astype(np.float64).sum(0), 4e-7 * np.abs(G).astype(np.float64).sum(0)
    for v in ("0", "1"):
        _check(res[v][0], want, bound)
        _check(res[v][1], cwant, cbound)
    assert not np.array_equal(res["0"][0], res["1"][0]), "REC_GEMM_BF16X3=1 did not select the bf16 x 3 weight-gradient kernel"


def test_gemm_nonfinite_operands(ops, monkeypatch):
    """Non-finite and extreme operands on both GEMM families (include/recengine.h "ARITHMETIC"):
      exact f32 (REC_GEMM_BF16X3=0): IEEE — an Inf operand gives +-Inf, relu(-Inf) = 0, relu(+Inf) = +Inf;
      bf16 x 3: the second term of an Inf (or of a finite |x| > 3.39e38, which rounds to the bf16 Inf) is Inf - Inf, so the
      row comes out NaN; behind BIAS_RELU the epilogue's fmaxf(NaN, 0) turns that into 0 — also where the exact kernel
      returns +Inf.  NaN operands give NaN in both (0 behind BIAS_RELU in both: fmaxf).  Rows without such an operand are
      untouched: finite and inside the float64 bound in both families.  Operands so small that their third bf16 term is
      subnormal (|x| < 2^-110) lose that term: an absolute error below 2^-126 per product, invisible at the bound."""
    rng = np.random.default_rng(11)
    M, N, K = 8192, 400, 400
    A, B, bias = _mk(rng, M, K), _mk(rng, K, N), _mk(rng, N)
    B[7, :] = np.abs(B[7, :]) + 0.1                       # a column-independent sign for the special operands' products
    A[100, 7] = np.inf
    A[200, 7] = -np.inf
    A[300, 7] = np.nan
    A[400, 7] = 3.4e38                                    # finite in f32, Inf as a bf16 first term
    A[500, :] = (A[500, :] * 2.0 ** -118).astype(np.float32)          # third terms below the bf16 normal range
    special = [100, 200, 300, 400]
    plain = np.ones(M, bool)
    plain[special] = False
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(DEV)
    out = {}
    for fam in ("0", "1"):
        monkeypatch.setenv("REC_GEMM_BF16X3", fam)
        ws = ops.Workspace(DEV)
        out[fam, "none"] = ops.gemm(t(A), t(B), ws).cpu().numpy()
        out[fam, "relu"] = ops.gemm(t(A), t(B), ws, epilogue="bias_relu", bias=t(bias)).cpu().numpy()
    Af = np.where(np.isfinite(A), A, 0).astype(np.float64)
    Af[400, 7] = 0
    want = Af @ B.astype(np.float64)
    bound = 4e-7 * (np.abs(Af) @ np.abs(B).astype(np.float64)) + 1e-37
    for fam in ("0", "1"):                                # rows without a special operand: finite, inside the bound
        C = out[fam, "none"]
        assert np.all(np.isfinite(C[plain]))
        _check(C[plain], want[plain], bound[plain])
        R = out[fam, "relu"]
        assert np.all(np.isfinite(R[plain])) and np.all(R[plain] >= 0)
    # exact f32: IEEE
    e, er = out["0", "none"], out["0", "relu"]
    assert np.all(np.isposinf(e[100])) and np.all(np.isneginf(e[200])) and np.all(np.isnan(e[300]))
    assert np.all(np.isfinite(e[400]) | np.isposinf(e[400]))                     # 3.4e38 x b (|b| <= 1.1): finite or overflow
    assert np.all(np.isposinf(er[100])) and np.all(er[200] == 0) and np.all(er[300] == 0)      # fmaxf(NaN, 0) = 0
    # bf16 x 3: the row of a non-finite / overflowing operand is NaN, and 0 behind the ReLU
    x, xr = out["1", "none"], out["1", "relu"]
    for r in special:
        assert np.all(np.isnan(x[r])), r
        assert np.all(xr[r] == 0), r
    assert not np.array_equal(out["0", "none"][plain], out["1", "none"][plain])     # the switch really switched


@pytest.mark.parametrize("M,N,K", [(8192 + 64, 400, 400), (16384 - 16, 400, 432), (8192, 512, 256), (9000, 368, 400)])
def test_gemm_relu_bits(ops, monkeypatch, M, N, K):
    """rec_gemm_epilogue_args.relu_bits: the BIAS_RELU forward leaves the ReLU mask of its output as bits, the RELU_MASK dX
    GEMM with the same (m, n) reads them instead of the activation — the same result bit for bit (deepfm/net.py:142-174:
    Linear -> ReLU and the ReLU' of its backward); calls without the bit form say so instead of ignoring the argument."""
    monkeypatch.setenv("REC_GEMM_BF16X3", "1")
    rng = np.random.default_rng(M + N)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(DEV)
    X, W, b = t(_mk(rng, M, K)), t(_mk(rng, K, N)), t(_mk(rng, N))
    ws = ops.Workspace(DEV)
    holder = []
    Y = ops.gemm(X, W, ws, epilogue="bias_relu", bias=b, relu_bits=holder)
    assert len(holder) == 1 and holder[0] is not None, "no bit form for a tall forward call"
    assert torch.equal(Y, ops.gemm(X, W, ws, epilogue="bias_relu", bias=b))             # the output itself is unchanged
    assert 0.2 < float((Y > 0).float().mean()) < 0.8
    # the backward of the NEXT layer: dY_in [M, N2] @ W2^T [N2, N] masked by Y > 0
    N2 = 400
    G, W2 = t(_mk(rng, M, N2)), t(_mk(rng, N, N2))
    want = ops.gemm(G, W2, ws, trans_b=True, epilogue="relu_mask", aux0=Y)
    got = ops.gemm(G, W2, ws, trans_b=True, epilogue="relu_mask", aux0=Y, relu_bits=holder[0])
    assert torch.equal(want, got)
    # aux0 may be absent when the bits are given
    d, x, out, need = ops._gemm_prepare(G, W2, False, True, "relu_mask", None, None, None, None, 0, None, None, None, 0, None)
    x.relu_bits = holder[0].data_ptr()
    w = ws.get(need)
    import ctypes as C
    ops.check(ops.lib().rec_gemm_f32(C.byref(d), ops._p(G), ops._p(W2), ops._p(out), C.byref(x), ops._p(w),
                                     C.c_size_t(w.numel()), ops._stream()), "rec_gemm_f32")
    assert torch.equal(want, out)
    # a call without the bit form (a short batch runs on the exact-f32 kernels): Python says None, the C entry refuses
    Xs = X[:512].contiguous()
    h2 = []
    ops.gemm(Xs, W, ws, epilogue="bias_relu", bias=b, relu_bits=h2)
    assert h2 == [None]
    d, x, out, need = ops._gemm_prepare(Xs, W, False, False, "bias_relu", b, None, None, None, 0, None, None, None, 0, None)
    x.relu_bits = holder[0].data_ptr()
    w = ws.get(need)
    rc = ops.lib().rec_gemm_f32(C.byref(d), ops._p(Xs), ops._p(W), ops._p(out), C.byref(x), ops._p(w),
                                C.c_size_t(w.numel()), ops._stream())
    assert rc != 0
    # the switch: REC_RELU_BITS=0 keeps every mask on the activation
    monkeypatch.setenv("REC_RELU_BITS", "0")
    h3 = []
    ops.gemm(X, W, ws, epilogue="bias_relu", bias=b, relu_bits=h3)
    assert h3 == [None]
