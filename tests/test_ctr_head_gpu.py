"""rec_ctr_head_fwd_bwd: the last Linear(n -> 1) + sigmoid + log_loss + mean and the backward of both in one pass over
the last hidden activation (deepfm/net.py:169-174 + dygraph_model.py:76-85) — against float64 torch autograd and
against the separate entry points it replaces (rec_gemm_f32 + rec_sigmoid_logloss + rec_mlp_head_bwd)."""
import pytest
import torch

DEV = "cuda"
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops(engine_lib):
    from paddlerec_amd import ops as o
    return o


def _ref(act, w, b, y1, y2, label, eps, clip, dtype=torch.float64):
    """float64: the reference; float32: the same graph in the kernel's own precision — the measured fp32 floor."""
    a = act.to(dtype).requires_grad_(True)
    wd = w.to(dtype).requires_grad_(True)
    bd = b.to(dtype).requires_grad_(True)
    y = a @ wd + bd
    z = y if y1 is None else (y1.to(dtype) + (y2.to(dtype) if y2 is not None else 0) + y)
    z.retain_grad()
    zc = z.clamp(clip[0], clip[1]) if clip else z
    p = torch.sigmoid(zc)
    t = label.to(dtype).reshape(-1, 1)
    loss = (-t * torch.log(p + eps) - (1 - t) * torch.log(1 - p + eps)).mean()
    loss.backward()
    return p.detach(), z.grad, loss.detach(), a.grad, wd.grad, bd.grad


@pytest.mark.parametrize("B,n,with_fm,clip", [(1, 16, True, None), (37, 400, True, None), (512, 400, True, None),
                                               (65536, 400, True, None), (4099, 512, False, (-15.0, 15.0)),
                                               (300, 128, False, (-0.5, 0.5)), (2048, 260, True, None)])
def test_ctr_head_against_float64_and_the_separate_calls(ops, B, n, with_fm, clip):
    g = torch.Generator(device=DEV).manual_seed(B + n)
    act = torch.relu(torch.randn(B, n, device=DEV, generator=g))
    w = torch.randn(n, 1, device=DEV, generator=g) / n ** 0.5
    b = torch.randn(1, device=DEV, generator=g) * 0.1
    y1 = torch.randn(B, 1, device=DEV, generator=g) * 0.3 if with_fm else None
    y2 = torch.randn(B, 1, device=DEV, generator=g) * 0.3 if with_fm else None
    label = (torch.rand(B, 1, device=DEV, generator=g) < 0.3).to(torch.int64)
    ws = ops.Workspace(DEV)
    dw, db = torch.empty(n, 1, device=DEV), torch.empty(1, device=DEV)
    pred, dz, loss, dx = ops.ctr_head(act, w, b, y1, y2, label, ws, dw, db, clip=clip)
    p_r, dz_r, loss_r, dx_r, dw_r, db_r = _ref(act, w, b, y1, y2, label, 1e-4, clip)
    # dx under the ReLU mask of act (the layer in front ends in a ReLU): autograd's d act is unmasked
    dx_r = torch.where(act > 0, dx_r, torch.zeros_like(dx_r))
    close = lambda x, r, tol: float((x.double() - r).abs().max()) <= tol * max(float(r.abs().max()), 1e-30)
    assert close(pred, p_r, 2e-6) and close(dz, dz_r, 1e-5) and close(loss, loss_r, 2e-6)
    assert close(dx, dx_r, 1e-5)
    # dw / db are sums over the batch of O(1/B) terms of either sign: 1e-5 of the result, or 4 x the distance of the SAME
    # graph in float32 (torch, its own summation order) from the float64 one — the measured fp32 floor
    from helpers import assert_close_floor
    _, _, _, _, dw32, db32 = _ref(act, w, b, y1, y2, label, 1e-4, clip, dtype=torch.float32)
    n64 = lambda x: x.detach().double().cpu().numpy()
    assert_close_floor(n64(dw), n64(dw_r), n64(dw32), rel=1e-5, err_msg="dw")
    assert_close_floor(n64(db).reshape(1), n64(db_r).reshape(1), n64(db32).reshape(1), rel=1e-5, err_msg="db")
    # ... and against the five launches it replaces: same arithmetic per element
    y = ops.gemm(act, w, ws, epilogue="bias", bias=b)
    a1, a2, a3 = (y1, y2, y) if with_fm else (y, None, None)
    pred2, dz2, loss2 = ops.sigmoid_logloss(a1, a2, a3, label, ws, clip=clip)
    assert close(pred, pred2.double(), 1e-6) and close(dz, dz2.double(), 2e-6) and close(loss, loss2.double(), 1e-6)
    if n <= 512 and n % 4 == 0:
        dw2, db2 = torch.empty(n, 1, device=DEV), torch.empty(1, device=DEV)
        dx2 = ops.mlp_head_bwd(act, dz2, w, ws, dw2, db2, relu=True)
        assert close(dx, dx2.double(), 2e-6) and close(dw, dw2.double(), 1e-5) and close(db, db2.double(), 1e-5)
    # deterministic
    dw3, db3 = torch.empty(n, 1, device=DEV), torch.empty(1, device=DEV)
    pred3, dz3, loss3, dx3 = ops.ctr_head(act, w, b, y1, y2, label, ws, dw3, db3, clip=clip)
    assert torch.equal(dw, dw3) and torch.equal(loss, loss3) and torch.equal(dx, dx3) and torch.equal(db, db3)


def test_ctr_head_on_strided_rows(ops):
    """act and dx as column slices of wider buffers (row strides larger than n): the kernel addresses rows by their stride."""
    g = torch.Generator(device=DEV).manual_seed(5)
    B, n = 1000, 400
    wide = torch.relu(torch.randn(B, 416, device=DEV, generator=g))
    act = wide[:, :n]
    w = torch.randn(n, 1, device=DEV, generator=g) / 20
    b = torch.zeros(1, device=DEV)
    label = (torch.rand(B, 1, device=DEV, generator=g) < 0.3).to(torch.int64)
    ws = ops.Workspace(DEV)
    dw, db = torch.empty(n, 1, device=DEV), torch.empty(1, device=DEV)
    dxw = torch.full((B, 432), 7.0, device=DEV)
    out = (torch.empty(B, 1, device=DEV), torch.empty(B, 1, device=DEV), torch.empty(1, device=DEV), dxw[:, :n])
    pred, dz, loss, dx = ops.ctr_head(act, w, b, None, None, label, ws, dw, db, out=out)
    dw2, db2 = torch.empty(n, 1, device=DEV), torch.empty(1, device=DEV)
    pred2, dz2, loss2, dx2 = ops.ctr_head(act.contiguous(), w, b, None, None, label, ws, dw2, db2)
    assert torch.equal(pred, pred2) and torch.equal(dz, dz2) and torch.equal(loss, loss2) and torch.equal(dw, dw2)
    assert torch.equal(dx, dx2) and bool((dxw[:, n:] == 7.0).all())          # nothing written behind column n


def test_ctr_head_argument_errors(ops):
    act = torch.zeros(8, 400, device=DEV)
    w, b = torch.zeros(400, 1, device=DEV), torch.zeros(1, device=DEV)
    label = torch.zeros(8, 1, dtype=torch.int64, device=DEV)
    ws = ops.Workspace(DEV)
    dw, db = torch.empty(400, 1, device=DEV), torch.empty(1, device=DEV)
    with pytest.raises(ops.RecError):                                    # n not a multiple of 4
        ops.ctr_head(act[:, :398].contiguous(), w[:398].contiguous(), b, None, None, label, ws, dw, db)
    with pytest.raises(ops.RecError):                                    # label of another batch
        ops.ctr_head(act, w, b, None, None, label[:4], ws, dw, db)
    with pytest.raises(ops.RecError):                                    # y2 without y1
        ops.ctr_head(act, w, b, None, torch.zeros(8, 1, device=DEV), label, ws, dw, db)
