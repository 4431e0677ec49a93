"""The block-tile FM kernels for narrow rows (paddlerec_amd/csrc/fm_tile.h: emb_dim 9 / 10 / 11 ..., the reference's own
table shapes — deepfm/config.yaml:48-50 D 9, benchmark.yaml:21 D 10) through the C-ABI, against the oracle:
gathered rows bit-exact, FM sums within 1e-5, every layout the dispatcher sends to them — a contiguous [N, D] table
(scalar row loads), 16- and 32-float records (three 16-byte loads per lookup), dense feat and feat at a padded sample
stride, tiles with a ragged tail (B not a multiple of the 32-sample tile), padding ids, out-of-range ids."""
import numpy as np
import pytest
import torch

from helpers import make_deepfm_problem
from oracle import deepfm_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda"
RTOL, ATOL = 1e-5, 2e-7


@pytest.fixture(scope="module")
def ops(engine_lib):
    from paddlerec_amd import ops as o
    return o


def T(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV)


def N_(t):
    return t.detach().cpu().numpy()


def _tables(p, D, rec_floats):
    """(W, W1) as the dispatcher sees them: separate contiguous tensors (rec_floats 0) or views of one record buffer."""
    if not rec_floats:
        return T(p["W"]), T(p["W1"])
    N = p["W"].shape[0]
    rec = torch.zeros(N, rec_floats, device=DEV)
    rec[:, :D] = T(p["W"])
    rec[:, D:D + 1] = T(p["W1"]).reshape(N, 1)
    return rec[:, :D], rec[:, D:D + 1]


@pytest.mark.parametrize("B,D,rec_floats,feat_ld", [
    (1, 9, 16, 0), (31, 9, 32, 0), (32, 9, 0, 0), (33, 10, 16, 0), (257, 9, 16, 400), (1000, 10, 16, 400),
    (70, 7, 8, 0), (65, 11, 16, 432), (129, 5, 8, 0), (2100, 9, 16, 352), (96, 10, 32, 0)])
def test_narrow_fm_fwd_and_bwd_vs_oracle(ops, B, D, rec_floats, feat_ld):
    pr = make_deepfm_problem(B=B, D=D, N=3000, seed=B + D, zipf=(B % 2 == 0))
    p = pr["params"]
    S, Dn = 26, 13
    F = S + Dn
    W, W1 = _tables(p, D, rec_floats)
    out = None
    if feat_ld:
        out = (torch.empty(B, 1, device=DEV), torch.empty(B, 1, device=DEV), torch.full((B, feat_ld), 7.0, device=DEV),
               torch.empty(B, D, device=DEV))
    y1, y2, feat, sum_emb, status = ops.deepfm_fm_fwd(T(pr["ids"]), T(pr["dense"]), W, W1, T(p["dense_w"]),
                                                      T(p["dense_w_one"]), 0, None, out=out, feat_ld=feat_ld)
    assert int(status.item()) == 0
    ry1, ry2, rfeat = R.fm_forward(pr["ids"], pr["dense"], p["W1"], p["W"], p["dense_w_one"], p["dense_w"], 0, None)
    got = N_(feat)
    if feat_ld:
        body, tail = got[:, :F * D], got[:, F * D:]
        assert np.array_equal(body.reshape(B, F, D), rfeat)
        q = (F * D + 3) // 4 * 4                    # whole float4 rows: zeros up to the next multiple of 4, the caller's
        assert np.all(tail[:, :q - F * D] == 0) and np.all(tail[:, q - F * D:] == 7.0)        # buffer beyond it untouched
    else:
        assert np.array_equal(got, rfeat)                                    # gather + multiply: bit-exact
    np.testing.assert_allclose(N_(y1), ry1, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(N_(y2), ry2, rtol=RTOL, atol=ATOL * D)
    np.testing.assert_allclose(N_(sum_emb), rfeat.sum(1), rtol=RTOL, atol=ATOL)

    # ---- backward on the same feat
    rng = np.random.default_rng(B)
    dfeat = (rng.standard_normal((B, F, D)) * 1e-3).astype(np.float32)
    dz = (rng.standard_normal((B, 1)) * 1e-3).astype(np.float32)
    dz2 = (rng.standard_normal((B, 1)) * 1e-3).astype(np.float32)
    if feat_ld:
        dfe = torch.full((B, feat_ld), 3.0, device=DEV)
        dfe[:, :F * D] = T(dfeat.reshape(B, -1))
    else:
        dfe = T(dfeat)
    ws = ops.Workspace(DEV)
    for dense_w in (None, T(p["dense_w"])):          # dense part of feat re-read / recomputed: the same bits
        rg, ddw, ddw1 = ops.deepfm_fm_bwd(T(pr["dense"]), feat, sum_emb, dfe, T(dz), T(dz2), S, ws, dense_w=dense_w,
                                          feat_ld=feat_ld)
        ref = R.fm_backward(pr["ids"], pr["dense"], rfeat, dfeat, dz, dz2, 0, None)
        np.testing.assert_allclose(N_(rg), ref["row_grad"], rtol=RTOL, atol=1e-9)
        ref64 = R.fm_backward(pr["ids"], pr["dense"].astype(np.float64), rfeat.astype(np.float64), dfeat.astype(np.float64),
                              dz.astype(np.float64), dz2.astype(np.float64))
        np.testing.assert_allclose(N_(ddw), ref64["d_dense_w"][0], rtol=RTOL, atol=RTOL * np.abs(ref64["d_dense_w"]).max())
        np.testing.assert_allclose(N_(ddw1), ref64["d_dense_w_one"], rtol=RTOL,
                                   atol=RTOL * np.abs(ref64["d_dense_w_one"]).max())
        rg2, ddw2, _ = ops.deepfm_fm_bwd(T(pr["dense"]), feat, sum_emb, dfe, T(dz), T(dz2), S, ws, dense_w=dense_w,
                                         feat_ld=feat_ld)
        assert torch.equal(rg, rg2) and torch.equal(ddw, ddw2)          # deterministic


def test_narrow_fm_padding_and_out_of_range(ops):
    pr = make_deepfm_problem(B=40, D=9, N=100, seed=3)
    p = pr["params"]
    pr["ids"][:5] = 0
    pr["ids"][7, 3] = 100
    pr["ids"][9, 25] = -4
    W, W1 = _tables(p, 9, 16)
    y1, y2, feat, sum_emb, status = ops.deepfm_fm_fwd(T(pr["ids"]), T(pr["dense"]), W, W1, T(p["dense_w"]),
                                                      T(p["dense_w_one"]), 0, None)
    assert int(status.item()) & 1
    f = N_(feat)
    assert np.all(f[:5, :26] == 0) and np.all(f[7, 3] == 0) and np.all(f[9, 25] == 0)
    ids_ok = pr["ids"].copy()
    ids_ok[7, 3] = 0
    ids_ok[9, 25] = 0
    ry1, ry2, rfeat = R.fm_forward(ids_ok, pr["dense"], p["W1"], p["W"], p["dense_w_one"], p["dense_w"], 0, None)
    assert np.array_equal(f, rfeat)
    np.testing.assert_allclose(N_(y1), ry1, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(N_(y2), ry2, rtol=RTOL, atol=ATOL * 9)


TILE_VS_ROWGROUP = r"""
import os, sys
import numpy as np, torch
REPO = %(repo)r
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from paddlerec_amd import ops
from helpers import make_deepfm_problem
T = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
pr = make_deepfm_problem(B=777, D=9, N=5000, seed=5)
p = pr["params"]
rec = torch.zeros(5000, 16, device="cuda"); rec[:, :9] = T(p["W"]); rec[:, 9:10] = T(p["W1"]).reshape(-1, 1)
out = ops.deepfm_fm_fwd(T(pr["ids"]), T(pr["dense"]), rec[:, :9], rec[:, 9:10], T(p["dense_w"]), T(p["dense_w_one"]), 0, None)
rng = np.random.default_rng(1)
df = T((rng.standard_normal((777, 39, 9)) * 1e-3).astype(np.float32)); dz = T((rng.standard_normal((777, 1)) * 1e-3).astype(np.float32))
bw = ops.deepfm_fm_bwd(T(pr["dense"]), out[2], out[3], df, dz, dz, 26, ops.Workspace("cuda"), dense_w=T(p["dense_w"]))
np.savez(sys.argv[1], y1=out[0].cpu(), y2=out[1].cpu(), feat=out[2].cpu(), sum_emb=out[3].cpu(), rg=bw[0].cpu(), ddw=bw[1].cpu(), ddw1=bw[2].cpu())
"""


def test_tile_kernels_agree_with_the_row_group_kernels(tmp_path, engine_lib):
    """REC_FM_TILE=0 keeps narrow rows on the row-group kernels: same lookups bit for bit, row gradients to the last
    bit or two (one elementwise formula, contracted differently), FM sums and batch sums within fp32 summation-order
    noise."""
    import os
    import subprocess
    import sys
    from conftest import REPO
    res = {}
    for tile in ("1", "0"):
        f = str(tmp_path / ("t%s.npz" % tile))
        r = subprocess.run([sys.executable, "-c", TILE_VS_ROWGROUP % dict(repo=REPO), f], capture_output=True, text=True,
                           env=dict(os.environ, REC_FM_TILE=tile, PYTHONDONTWRITEBYTECODE="1"), timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        res[tile] = np.load(f)
    a, b = res["1"], res["0"]
    assert np.array_equal(a["feat"], b["feat"])
    np.testing.assert_allclose(a["rg"], b["rg"], rtol=1e-5, atol=1e-9)     # same formula, the compilers' own fma choices
    for k in ("y1", "y2", "sum_emb"):
        np.testing.assert_allclose(a[k], b[k], rtol=1e-5, atol=2e-6, err_msg=k)
    for k in ("ddw", "ddw1"):
        np.testing.assert_allclose(a[k], b[k], rtol=1e-5, atol=1e-5 * float(np.abs(b[k]).max()), err_msg=k)
