"""Seeded shape sweep of the row-update kernels added in round 2 against torch on the same device: the lane-per-row lazy
Adam (rows without float4 groups: every width 1..16, aligned and unaligned strides, views with offsets), the one-launch
merge + SGD / record-Adam (any n up to the limit, widths up to 256, strided gradient views, duplicates from none to
one row owning most lookups).  The kernels' own oracle tests use a few shapes; this walks the corners."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _adam_ref(P, M, V, rows, g, t, lr, b1=0.9, b2=0.999, eps=1e-8):
    """float64 lazy Adam on the merged gradient g [U, D] of rows [U] (Paddle's epsilon placement)."""
    lr_t = lr * (1 - b2 ** t) ** 0.5 / (1 - b1 ** t)
    m = b1 * M[rows].double() + (1 - b1) * g
    v = b2 * V[rows].double() + (1 - b2) * g * g
    p = P[rows].double() - lr_t * m / (v.sqrt() + eps * (1 - b2 ** t) ** 0.5)
    return p, m, v


@pytest.mark.parametrize("seed", range(24))
def test_narrow_adam_rows_shape_sweep(engine_lib, seed):
    from paddlerec_amd import ops
    rng = np.random.default_rng(1000 + seed)
    D = int(rng.integers(1, 17))
    N = int(rng.integers(50, 4000))
    n = int(rng.integers(1, 30000))
    stride = D + int(rng.integers(0, 9))
    sstride = D + int(rng.integers(0, 9))
    if rng.random() < 0.5:                                  # record-like: 16-B aligned strides
        stride, sstride = (stride + 3) // 4 * 4, (sstride + 3) // 4 * 4
    off = int(rng.integers(0, 3)) * (4 if stride % 4 == 0 else 1)
    g = torch.Generator(device=DEV).manual_seed(seed)
    Pbuf = torch.randn(N, stride + off, device=DEV, generator=g)
    Mbuf = torch.randn(N, sstride, device=DEV, generator=g) * 1e-2
    Vbuf = torch.rand(N, sstride, device=DEV, generator=g) * 1e-3
    P, M, V = Pbuf[:, off:off + D], Mbuf[:, :D], Vbuf[:, :D]
    P0, M0, V0 = P.clone(), M.clone(), V.clone()
    hot = rng.random() < 0.3
    ids = torch.from_numpy(rng.integers(0, N, size=n)).to(DEV)
    if hot:
        ids[torch.rand(n, device=DEV, generator=g) < 0.6] = int(rng.integers(1, N))
    pitch = D + int(rng.integers(0, 5))
    grad_full = torch.randn(n, pitch, device=DEV, generator=g) * 1e-2
    grad = grad_full[:, :D]
    t = int(rng.integers(1, 50))
    ws = ops.Workspace(DEV)
    groups, status = ops.ids_group(ids, N, 0, ws)
    pp = ops.segment_partials(groups, grad, D, grad_group=1, grad_group_stride=pitch)
    ops.sparse_adam_rows(groups, grad, 1, P, M, V, t, lr=1e-2, grad_group=1, grad_group_stride=pitch, partials=pp)
    live = ids != 0
    gs = torch.zeros(N, D, device=DEV, dtype=torch.float64)
    gs.index_add_(0, ids[live], grad[live].double())
    rows = torch.nonzero(torch.bincount(ids[live], minlength=N) > 0).reshape(-1)
    p, m, v = _adam_ref(P0, M0, V0, rows, gs[rows], t, 1e-2)
    for got, want, full in ((M, m, M0), (V, v, V0), (P, p, P0)):
        torch.testing.assert_close(got[rows].double(), want, rtol=2e-5,
                                   atol=2e-5 * float(want.abs().max().item()) if rows.numel() else 0.0)
        mask = torch.ones(N, dtype=torch.bool, device=DEV)
        mask[rows] = False
        assert torch.equal(got[mask], full[mask])           # untouched rows bit-identical
    if off or stride + off > D:                             # nothing outside the D columns was written
        ref = torch.randn(N, stride + off, device=DEV, generator=torch.Generator(device=DEV).manual_seed(seed))
        assert torch.equal(Pbuf[:, :off], ref[:, :off]) and torch.equal(Pbuf[:, off + D:], ref[:, off + D:])
    assert int(status.item()) == 0


@pytest.mark.parametrize("seed", range(16))
def test_small_merge_shape_sweep(engine_lib, seed):
    from paddlerec_amd import ops
    rng = np.random.default_rng(2000 + seed)
    D = int(rng.choice([1, 3, 16, 17, 64, 65, 128, 200, 256]))
    N = int(rng.integers(2, 5000))
    n = int(rng.choice([1, 63, 64, 65, 1000, 4864, 13312, 15360]))
    g = torch.Generator(device=DEV).manual_seed(seed)
    ids = torch.from_numpy(rng.integers(0, N, size=n)).to(DEV)
    if rng.random() < 0.4 and n > 10:
        ids[torch.rand(n, device=DEV, generator=g) < 0.7] = int(rng.integers(0, N))
    pitch = D + int(rng.integers(0, 4)) * 4
    grad = torch.randn(n, pitch, device=DEV, generator=g)[:, :D]
    pad = 0 if rng.random() < 0.5 else None
    P = torch.randn(N, D, device=DEV, generator=g)
    P0 = P.clone()
    st = ops.sparse_sgd_small(ids, grad, P, 0.25, padding_idx=pad, grad_group=1, grad_group_stride=pitch)
    live = (ids != 0) if pad == 0 else torch.ones_like(ids, dtype=torch.bool)
    want = P0.double()
    want.index_add_(0, ids[live], -0.25 * grad[live].double())
    torch.testing.assert_close(P.double(), want, rtol=1e-5, atol=1e-5 * float(want.abs().max().item()))
    assert int(st.item()) == 0
    if D <= 64:            # the record form: W | W1 | m1 | v1 in one line, moments in a second
        S = int(rng.choice([1, 2, 13, 26]))
        S = S if n >= S else 1
        B = n // S
        ids2 = ids[: B * S].reshape(B, S).contiguous()
        rec = torch.randn(N, (D + 3 + 31) // 32 * 32, device=DEV, generator=g) * 0.1
        Dp = (D + 3) // 4 * 4
        mv = torch.rand(N, (2 * Dp + 31) // 32 * 32, device=DEV, generator=g) * 1e-3
        rec[:, D + 2].abs_()
        rec0, mv0 = rec.clone(), mv.clone()
        rg = torch.randn(B * S, D, device=DEV, generator=g) * 1e-2
        dz = torch.randn(B, device=DEV, generator=g) * 1e-2
        t = int(rng.integers(1, 20))
        st2 = ops.sparse_adam_record_small(ids2, None, 0, rg, dz, S, rec, mv, D, t, lr=1e-2, v_offset=Dp)
        flat = ids2.reshape(-1)
        lv = flat != 0
        gs = torch.zeros(N, D, device=DEV, dtype=torch.float64)
        gs.index_add_(0, flat[lv], rg[lv].double())
        g1 = torch.zeros(N, 1, device=DEV, dtype=torch.float64)
        g1.index_add_(0, flat[lv], dz.repeat_interleave(S)[lv].double().unsqueeze(1))
        rows = torch.nonzero(torch.bincount(flat[lv], minlength=N) > 0).reshape(-1)
        p, m, v = _adam_ref(rec0[:, :D], mv0[:, :D], mv0[:, Dp:Dp + D], rows, gs[rows], t, 1e-2)
        p1, m1, v1 = _adam_ref(rec0[:, D:D + 1], rec0[:, D + 1:D + 2], rec0[:, D + 2:D + 3], rows, g1[rows], t, 1e-2)
        for got, want_ in ((rec[:, :D], p), (mv[:, :D], m), (mv[:, Dp:Dp + D], v), (rec[:, D:D + 1], p1),
                           (rec[:, D + 1:D + 2], m1), (rec[:, D + 2:D + 3], v1)):
            if rows.numel():
                torch.testing.assert_close(got[rows].double(), want_, rtol=2e-5,
                                           atol=2e-5 * float(want_.abs().max().item()))
        mask = torch.ones(N, dtype=torch.bool, device=DEV)
        mask[rows] = False
        assert torch.equal(rec[mask], rec0[mask]) and torch.equal(mv[mask], mv0[mask])
        assert int(st2.item()) == 0


@pytest.mark.parametrize("seed", range(12))
def test_ps_push_narrow_shape_sweep(engine_lib, seed):
    """rec_ps_push_rows on narrow features (lane-per-feature kernel: embedx_dim 1..15 at record offset 1) against the
    accessor oracle: tables holding unborn / embed-only / full features, thresholds that let some features be born
    with and some without their embedx part, show and click inputs, duplicates and padding."""
    from oracle import deepfm_ref as R
    from oracle import ps_ref
    from paddlerec_amd import ops
    rng = np.random.default_rng(3000 + seed)
    D = int(rng.integers(2, 17))                       # looked-up width: embed_w + embedx(D-1)
    N = int(rng.integers(20, 300))
    S = int(rng.integers(1, 5))
    B = int(rng.integers(4, 120))
    thr = float(rng.choice([0.0, 1.5, 10.0]))
    # the two SGD rules differ (embed_sgd_param / embedx_sgd_param are separate blocks of the YAML), the pushed
    # gradient is scaled (batch size), show scaling on / off, embed_w zero-initialised or not
    kw = dict(embedx_threshold=thr, initial_range=1e-2, seed=77 + seed, lr=0.05, embedx_lr=float(rng.choice([0.05, 0.02])),
              initial_g2sum=3.0, embedx_initial_g2sum=float(rng.choice([3.0, 1.0])), bounds=(-10.0, 10.0),
              embedx_bounds=(-0.25, 0.25), embedx_initial_range=2e-2, grad_scale=float(rng.choice([1.0, 8.0])),
              show_scale=bool(rng.integers(0, 2)), embed_zero_init=bool(rng.integers(0, 2)))
    table = ops.PsTable(N, D, DEV, kind="slot", **kw)
    L = table.layout
    lay = dict(embed_off=L.embed_off, embedx_off=L.embedx_off, embedx_dim=L.embedx_dim, stat_off=L.stat_off)
    acc = dict(kw, nonclk_coeff=0.1, click_coeff=1.0)
    rec = np.zeros((N, L.row_stride), np.float32)
    for r in range(1, N):                              # a table with history: a third unborn, a third embed-only, a third full
        kind = rng.integers(0, 3)
        if kind == 0:
            continue
        rec[r, L.embed_off] = rng.standard_normal() * 0.1
        rec[r, L.stat_off:L.stat_off + 4] = [rng.integers(0, 30), rng.integers(0, 3), rng.random() * 0.5, rng.random() * 0.5]
        rec[r, L.stat_off + 1] = min(rec[r, L.stat_off + 1], rec[r, L.stat_off])
        rec[r, L.stat_off + 4] = kind
        rec[r, L.stat_off + 5:L.stat_off + 7] = [rng.random(), rng.integers(0, 9)]      # delta_score, unseen_days
        if kind == 2:
            rec[r, L.embedx_off:L.embedx_off + D - 1] = rng.standard_normal(D - 1) * 0.1
    table.rec.copy_(torch.from_numpy(rec))
    ids = rng.integers(0, N, size=(B, S)).astype(np.int64)
    ids[rng.random((B, S)) < 0.2] = ids[0, 0]          # a popular feature
    grad = (rng.standard_normal((B * S, D)) * 0.05).astype(np.float32)
    show = rng.integers(1, 4, size=B).astype(np.int64)
    click = (rng.random(B) < 0.4).astype(np.int64)
    tids = torch.from_numpy(ids).to(DEV)
    groups, status = ops.ids_group(tids, N, 0, ops.Workspace(DEV))
    ops.ps_push_rows(table, groups, torch.from_numpy(grad).to(DEV), S, show=torch.from_numpy(show).to(DEV),
                     click=torch.from_numpy(click).to(DEV))
    rows = ids.reshape(-1)
    valid = rows != 0
    uniq, merged, _ = R.merge_rows(rows, valid, grad)
    pos_s = np.repeat(show, S).astype(np.float64)
    pos_c = np.repeat(click, S).astype(np.float64)
    dshow = np.array([pos_s[(rows == u) & valid].sum() for u in uniq])
    dclick = np.array([pos_c[(rows == u) & valid].sum() for u in uniq])
    want = rec.copy()
    ps_ref.push_rows(want, lay, uniq, merged[:, 0], merged[:, 1:], dshow, dclick, acc)
    got = table.rec.cpu().numpy()
    so = L.stat_off
    assert np.array_equal(got[:, so:so + 2], want[:, so:so + 2])            # show / click: exact
    assert np.array_equal(got[:, so + 4], want[:, so + 4])                  # feature states: exact
    assert np.array_equal(got[:, so + 6], want[:, so + 6])                  # unseen_days: exact
    np.testing.assert_allclose(got[:, so + 5], want[:, so + 5], rtol=1e-6)  # delta_score
    # weights / g2sums: the device merges duplicate gradients in the same ascending order as the oracle; what is left
    # is the double -> float rounding of the rule (1 ulp)
    np.testing.assert_allclose(got[:, :D], want[:, :D], rtol=2e-6, atol=1e-8)
    np.testing.assert_allclose(got[:, so + 2:so + 4], want[:, so + 2:so + 4], rtol=2e-6, atol=1e-12)
    untouched = np.setdiff1d(np.arange(N), uniq)
    assert np.array_equal(got[untouched], rec[untouched])
    assert int(status.item()) == 0


@pytest.mark.parametrize("case", ["uniform", "zipf", "one_row_owns_most", "all_padding_but_one", "oob"])
@pytest.mark.parametrize("D", [9, 16])
def test_bucket_merge_equals_sorted_merge_bitwise(engine_lib, case, D):
    """One shared table, more than 2048 lookups: rec_sparse_adam_record_small merges by ROW BUCKETS
    (sparse_bucket_kernel).  A row's gradients are added in ascending position, as the sort-based merge
    (rec_ids_group + rec_sparse_adam_record without partials) adds them: records and moments bit-identical — also
    when one row owns most of the batch (its bucket does not fit the LDS list: the block walks global memory), with
    padding ids and with ids outside the table (flagged, not applied)."""
    from paddlerec_amd import ops
    from paddlerec_amd import _lib as L
    rng = np.random.default_rng(77)
    B, S, N = 512, 26, 3000
    n = B * S
    if case == "zipf":
        ids_np = np.minimum(rng.zipf(1.2, size=n), N - 1).astype(np.int64)
    else:
        ids_np = rng.integers(0, N, size=n)
    if case == "one_row_owns_most":
        ids_np[rng.random(n) < 0.8] = 1234
    if case == "all_padding_but_one":
        ids_np[:] = 0
        ids_np[n // 2] = 17
    if case == "oob":
        ids_np[rng.random(n) < 0.01] = N + 5
        ids_np[7] = -3
    ids = torch.from_numpy(ids_np).to(DEV).reshape(B, S)
    g = torch.Generator(device=DEV).manual_seed(3)
    stride = 16 if D == 9 else 32
    Dp = (D + 3) // 4 * 4
    rec = torch.randn(N, stride, device=DEV, generator=g) * 0.1
    mv = torch.rand(N, 32, device=DEV, generator=g) * 1e-3
    rec[:, D + 2].abs_()
    rec_b, mv_b = rec.clone(), mv.clone()
    rg = torch.randn(n, D, device=DEV, generator=g) * 1e-2
    dz = torch.randn(B, device=DEV, generator=g) * 1e-2
    st = ops.sparse_adam_record_small(ids, None, 0, rg, dz, S, rec, mv, D, 5, lr=1e-2, v_offset=Dp)
    ws = ops.Workspace(DEV)
    groups, st_b = ops.ids_group(ids.reshape(-1), N, 0, ws)
    ops.sparse_adam_record(groups, rg, dz, S, rec_b, mv_b, D, 5, lr=1e-2, v_offset=Dp)
    assert torch.equal(rec, rec_b) and torch.equal(mv, mv_b)
    want = L.REC_FLAG_INDEX_OOB if case == "oob" else 0
    assert int(st.item()) == want and int(st_b.item()) == want
