"""Row P — the multi-slot sum-pool kernel and the BenchmarkDNNLayer net built on it (paddlerec_amd/slot_dnn.py;
reference: models/rank/slot_dnn/net.py:55-85, static_model.py:104-112, queuedataset_reader.py:56-82).

Oracle (oracle/slot_dnn_ref.py) pinned to tests/golden/slot_dnn_D9.npz = the reference's unmodified net.py executed
over the paddle shim (sparse_embedding + sequence_pool + clip).  CPU: oracle vs golden, host logic of the mirror with
the oracle-backed operator backend, the feasign hash (C host restatement vs oracle).  `-m gpu`: the HIP kernels
through the C-ABI — pooled sums within 1e-5, counts / segment ids / rows bit-exact."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from helpers import assert_close_scaled, load_golden
from oracle import deepfm_ref as R
from oracle import slot_dnn_ref as M

DEV = "cuda"


def T(a, dev=DEV):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dev)


def _golden_problem():
    g = load_golden("slot_dnn_D9")
    S = int(g["S"])
    samples = [[[int(x) for x in cell.split(",")] for cell in row] for row in g["samples"]]
    values, lod, base = M.csr_from_samples(samples, S)
    n = int(g["n_mlp"])
    return g, samples, values, lod, base, [g["mlp_w%d" % i].copy() for i in range(n)], \
        [g["mlp_b%d" % i].copy() for i in range(n)]


def test_oracle_matches_reference_golden():
    g, _, values, lod, base, mw, mb = _golden_problem()
    o = M.loss_and_grads(values, lod, base, g["label"], g["W"], mw, mb, padding_idx=0, key_mode=0)
    np.testing.assert_allclose(o["pool"], g["pooled"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(o["pred"], g["pred"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(o["loss"], g["loss"], rtol=1e-6)
    assert int(g["n_clipped"]) >= 1                                   # the +-15 clip and its zero gradient are exercised
    for i in range(int(g["n_mlp"])):
        # atol = fp32 summation noise of O(0.1..1) gradients (the x24 weights of the fixture): NumPy and torch
        # reduce in different orders
        np.testing.assert_allclose(o["dws"][i], g["g_mlp_w%d" % i], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(o["dbs"][i], g["g_mlp_b%d" % i], rtol=1e-5, atol=1e-6)
    gW = np.zeros_like(g["gW"])
    gW[o["uniq"]] = o["merged"]
    np.testing.assert_allclose(gW, g["gW"], rtol=1e-5, atol=1e-6)
    assert np.all(g["gW"][0] == 0)                                    # padding row: no gradient


def test_feasign_hash_host_restatement_matches_oracle(engine_lib):
    from paddlerec_amd import ops
    rng = np.random.default_rng(7)
    keys = rng.integers(0, 2 ** 64, size=4000, dtype=np.uint64)
    keys[:6] = [0, 1, 2 ** 63, 2 ** 64 - 1, 2 ** 63 + 5, 12345678901234567890]   # > 2^63 feasigns occur
    for N in (2, 1000, 1_000_003, 10 ** 10):
        got = ops.feasign_rows_host(keys, N)
        assert np.array_equal(got, M.feasign_rows(keys, N))
        assert got[0] == 0 and got[1:].min() >= 1 and got.max() < N
    assert M.mix64(1) == 0xB456BCFC34C2CB2C                           # murmur3 fmix64(1), published vector


def test_ps_init_value_host_matches_oracle(engine_lib):
    from oracle import ps_ref
    for seed, row, d in [(2025, 0, 0), (2025, 17, 3), (1, 10 ** 10 - 1, 16), (2 ** 63 + 9, 123456789, 8)]:
        got = engine_lib.rec_ps_init_value_host(seed, row, d, 1e-4)
        assert np.float32(got) == ps_ref.init_value(seed, row, d, 1e-4)
        assert abs(got) <= 1e-4


def _random_problem(rng, B, S, N, max_len=6, empty_frac=0.2, pad_frac=0.1, feasigns=False):
    samples = []
    for b in range(B):
        row = []
        for s in range(S):
            k = int(rng.integers(0, max_len + 1))
            if rng.random() < empty_frac:
                k = 0
            if feasigns:
                v = [int(x) for x in rng.integers(1, 2 ** 64, size=k, dtype=np.uint64)]
            else:
                v = [int(x) for x in rng.integers(1, N, size=k)]
            v = [0 if rng.random() < pad_frac else x for x in v]
            row.append(v)                                              # may be EMPTY: lod[b] == lod[b+1]
        samples.append(row)
    return samples


def _layer_from_golden(g, mw, mb, device, kernels=None, **kw):
    from paddlerec_amd.slot_dnn import BenchmarkDNNLayer
    m = BenchmarkDNNLayer(int(g["N"]), int(g["D"]), int(g["S"]), [w.shape[1] for w in mw[:-1]], device=device,
                          kernels=kernels, key_mode=0, **kw)
    sd = {"embedding": g["W"]}
    for i, (w, b) in enumerate(zip(mw, mb)):
        sd["linear_%d.weight" % i], sd["linear_%d.bias" % i] = w, b
    m.set_dict(sd)
    return m


def _check_layer_step(device, kernels, Batch):
    g, _, values, lod, base, mw, mb = _golden_problem()
    m = _layer_from_golden(g, mw, mb, device, kernels)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(device)
    mbatch = Batch(t(values), t(lod), t(base))
    pred0 = m.forward(mbatch)
    np.testing.assert_allclose(pred0.cpu().numpy(), g["pred"], rtol=1e-5, atol=1e-6)
    W0 = g["W"].copy()
    loss, pred = m.train_step(mbatch, t(g["label"]), lr=1e-3)
    np.testing.assert_allclose(loss.cpu().numpy()[0], g["loss"], rtol=1e-5)
    for i in range(int(g["n_mlp"])):
        assert_close_scaled(m.mlp_dw[i].cpu().numpy(), g["g_mlp_w%d" % i], 1e-5)
        assert_close_scaled(m.mlp_db[i].cpu().numpy(), g["g_mlp_b%d" % i], 1e-5)
    # lazy Adam, step 1, on exactly the rows the reference's sparse gradient touches
    o = M.loss_and_grads(values, lod, base, g["label"], W0, mw, mb, 0, 0)
    Wn, Mn, Vn = W0.copy(), np.zeros_like(W0), np.zeros_like(W0)
    R.adam_update_rows(Wn, Mn, Vn, o["uniq"], o["merged"], 1, lr=1e-3)
    got = m.embedding.cpu().numpy()
    assert_close_scaled(m.sparse_state["m"].cpu().numpy(), Mn, 1e-5)
    np.testing.assert_allclose(got, Wn, rtol=1e-5, atol=2e-6)
    untouched = np.setdiff1d(np.arange(W0.shape[0]), o["uniq"])
    assert np.array_equal(got[untouched], W0[untouched])
    assert np.array_equal(m.last_counts.cpu().numpy(), o["counts"])
    assert int(m.status.item()) == 0


def test_layer_host_logic_on_cpu_backend():
    """Orchestration of the mirror (buffers, grad-index plumbing, optimizer bookkeeping) with the oracle-backed
    operator stand-in — no GPU, no HIP kernel."""
    import cpu_kernels
    _check_layer_step("cpu", cpu_kernels, cpu_kernels.MultislotBatch)


def test_layer_ps_accessor_host_logic_on_cpu_backend():
    import cpu_kernels
    _check_ps_layer("cpu", cpu_kernels, cpu_kernels.MultislotBatch)


def _check_ps_layer(device, kernels, Batch):
    """sparse_optimizer='ps': lazy birth at the first pull, AdaGrad rule per part, show/click from the inputs,
    embedx created only once the score reaches embedx_threshold — three steps against oracle/ps_ref.py."""
    from oracle import ps_ref
    from paddlerec_amd.slot_dnn import BenchmarkDNNLayer
    g, _, values, lod, base, mw, mb = _golden_problem()
    N, D, S = int(g["N"]), int(g["D"]), int(g["S"])
    accp = dict(lr=0.05, initial_g2sum=3.0, bounds=(-10.0, 10.0), initial_range=1e-2, embedx_threshold=1.5,
                nonclk_coeff=0.1, click_coeff=1.0, seed=99)
    m = BenchmarkDNNLayer(N, D, S, [w.shape[1] for w in mw[:-1]], device=device, kernels=kernels, key_mode=0,
                          sparse_optimizer="ps", accessor=accp)
    sd = {}
    for i, (w, b) in enumerate(zip(mw, mb)):
        sd["linear_%d.weight" % i], sd["linear_%d.bias" % i] = w / 24.0, b     # un-saturate: real gradients flow
    m.set_dict(sd)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(device)
    mbatch = Batch(t(values), t(lod), t(base))
    L = m.table.layout
    lay = dict(embed_off=L.embed_off, embedx_off=L.embedx_off, embedx_dim=L.embedx_dim, stat_off=L.stat_off)
    rec = np.zeros((N, L.row_stride), np.float32)
    mwn = [w / 24.0 for w in mw]
    label = g["label"]
    live = values != 0
    for step in range(3):
        # what the pull shows: born rows their weights, unborn rows their creation values
        Wv = np.stack([ps_ref.pull_value(rec, lay, r, accp, D) for r in range(N)])
        o = M.loss_and_grads(values, lod, base, label, Wv, mwn, mb, 0, 0)
        loss, _ = m.train_step(mbatch, t(label), lr=1e-3)
        np.testing.assert_allclose(loss.cpu().numpy()[0], o["loss"], rtol=2e-5)
        U = len(o["uniq"])
        dshow, dclick = np.zeros(U), np.zeros(U)
        pos = {int(r): i for i, r in enumerate(o["uniq"])}
        for k in np.nonzero(live)[0]:
            dshow[pos[int(o["rows"][k])]] += 1
            dclick[pos[int(o["rows"][k])]] += int(label[o["seg"][k] // S, 0])
        # the layer pushes the gradient of the SUMMED loss (scale_sparse_grad): grad_scale = batch size
        ps_ref.push_rows(rec, lay, o["uniq"], o["merged"][:, 0], o["merged"][:, 1:], dshow, dclick,
                         dict(accp, grad_scale=float(label.shape[0])))
        got = m.rec.cpu().numpy()
        so = L.stat_off
        assert np.array_equal(got[:, so:so + 2], rec[:, so:so + 2]), "show / click counters"
        assert np.array_equal(got[:, so + 4], rec[:, so + 4]), "feature states"
        assert np.array_equal(got[:, so + 6], rec[:, so + 6]), "unseen_days"
        np.testing.assert_allclose(got[:, so + 5], rec[:, so + 5], rtol=1e-6, err_msg="delta_score")
        # weights / g2sums: the merged gradient differs from the oracle's by float summation order (1e-6 of its
        # scale); relative to the weight scale of the table
        wscale = float(np.abs(rec[:, :D]).max())
        np.testing.assert_allclose(got[:, :D], rec[:, :D], rtol=1e-5, atol=1e-5 * wscale)
        np.testing.assert_allclose(got[:, so + 2:so + 4], rec[:, so + 2:so + 4], rtol=2e-5, atol=1e-12)
        # the MLP's dense Adam moved the weights: mirror it for the next step's oracle forward
        mwn = [m.mlp_w[i].cpu().numpy().copy() for i in range(len(mwn))]
        mb = [m.mlp_b[i].cpu().numpy().copy() for i in range(len(mb))]
    st = rec[:, L.stat_off + 4]
    assert (st == 0).any() and (st == 1).any() and (st == 2).any()    # unborn, embed-only and full features all occur


# ----------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("D,stride,B,S,N", [(9, 16, 130, 21, 4001), (9, 9, 70, 5, 300), (16, 16, 64, 8, 1000),
                                            (4, 4, 200, 3, 50), (1, 1, 33, 2, 20), (8, 8, 65, 17, 500),
                                            (40, 40, 50, 3, 200), (64, 64, 20, 2, 100), (10, 12, 128, 16, 999)])
def test_multislot_sumpool_vs_oracle(engine_lib, D, stride, B, S, N):
    from paddlerec_amd import ops
    rng = np.random.default_rng(D * 1000 + B)
    samples = _random_problem(rng, B, S, N)
    values, lod, base = M.csr_from_samples(samples, S)
    Wfull = rng.standard_normal((N, stride)).astype(np.float32)
    W = Wfull[:, :D]
    tW = T(Wfull)[:, :D]
    mbatch = ops.MultislotBatch(T(values), T(lod), T(base))
    out, counts, seg, rows, status = ops.multislot_sumpool(mbatch, tW, N, 0, 0)
    want, wcnt, wseg, wrows = M.multislot_sumpool(values, lod, base, W, 0, 0)
    assert np.array_equal(counts.cpu().numpy(), wcnt)
    assert np.array_equal(seg.cpu().numpy()[: len(values)], wseg)
    assert np.array_equal(rows.cpu().numpy()[: len(values)], wrows)
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-5, atol=1e-6)
    assert int(status.item()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("max_len,D,stride", [(12, 9, 16), (16, 9, 16), (17, 9, 16), (16, 12, 12), (16, 16, 16),
                                               (15, 5, 5), (16, 20, 20)])
def test_multislot_run_lengths_around_a_substep(engine_lib, max_len, D, stride):
    """Run lengths around the 16 row groups of one gather sub-step and pieces of (slot, 64 samples) around 320 and 512
    ids (the per-wave segment table holds 512), 30 % empty segments, feasign keys, padding ids; widths with 1-5 float4
    or scalar lanes per row."""
    from paddlerec_amd import ops
    rng = np.random.default_rng(max_len * 31 + D)
    B, S, N = 150, 19, 5003
    samples = _random_problem(rng, B, S, N, max_len=max_len, empty_frac=0.3, feasigns=True)
    values, lod, base = M.csr_from_samples(samples, S)
    Wfull = rng.standard_normal((N, stride)).astype(np.float32)
    mbatch = ops.MultislotBatch(T(values), T(lod), T(base))
    out, counts, seg, rows, status = ops.multislot_sumpool(mbatch, T(Wfull)[:, :D], N, 0, 1)
    want, wcnt, wseg, wrows = M.multislot_sumpool(values, lod, base, Wfull[:, :D], 0, 1, N)
    assert np.array_equal(counts.cpu().numpy(), wcnt)
    assert np.array_equal(seg.cpu().numpy()[: len(values)], wseg)
    assert np.array_equal(rows.cpu().numpy()[: len(values)], wrows)
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-5, atol=2e-6)
    assert int(status.item()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("max_len,empty,D,stride,S", [(70, 0.0, 9, 16, 12), (3, 0.6, 9, 16, 33), (33, 0.2, 10, 12, 16),
                                                      (200, 0.5, 8, 8, 9), (18, 0.1, 12, 12, 16), (5, 0.9, 11, 12, 40)])
def test_multislot_lane_per_id_kernel(engine_lib, monkeypatch, max_len, empty, D, stride, S):
    """The lane-per-id kernel (narrow rows) against the oracle AND the row-group kernel (REC_MS_LANE=0): runs that
    cross DPP rows, cover whole rows (the four-pass flush), cross list drains and slots; mostly-empty slots."""
    from paddlerec_amd import ops
    rng = np.random.default_rng(max_len * 7 + S)
    B, N = 203, 6007
    samples = _random_problem(rng, B, S, N, max_len=max_len, empty_frac=empty, feasigns=True)
    values, lod, base = M.csr_from_samples(samples, S)
    Wfull = rng.standard_normal((N, stride)).astype(np.float32)
    mbatch = ops.MultislotBatch(T(values), T(lod), T(base))
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("REC_MS_LANE", flag)
        out, counts, seg, rows, status = ops.multislot_sumpool(mbatch, T(Wfull)[:, :D], N, 0, 1)
        res[flag] = (out.cpu().numpy(), counts.cpu().numpy(), seg.cpu().numpy(), rows.cpu().numpy(), int(status.item()))
    want, wcnt, wseg, wrows = M.multislot_sumpool(values, lod, base, Wfull[:, :D], 0, 1, N)
    for flag in ("1", "0"):
        out, counts, seg, rows, st = res[flag]
        assert np.array_equal(counts, wcnt) and st == 0
        assert np.array_equal(seg[: len(values)], wseg) and np.array_equal(rows[: len(values)], wrows)
        scale = np.abs(Wfull).max() * max(1, max_len) ** 0.5
        np.testing.assert_allclose(out, want, rtol=1e-5, atol=2e-6 * scale)
    # a second launch gives the same bits (fixed summation order)
    monkeypatch.setenv("REC_MS_LANE", "1")
    out2 = ops.multislot_sumpool(mbatch, T(Wfull)[:, :D], N, 0, 1)[0].cpu().numpy()
    assert np.array_equal(out2, res["1"][0])


@pytest.mark.gpu
@pytest.mark.parametrize("N", [2, 1_000_003, 1_250_000_000, 2 ** 32 + 7, 10 ** 10, 2 ** 62 + 1])
def test_feasign_rows_device_matches_oracle_at_configs4_sizes(engine_lib, N):
    """uint64 feasign -> row of the hashed table on the device (multiply-high exact modulo), bit-exact against the
    oracle for divisors below and above 2^32 — 10^10 rows is BASELINE configs[4]'s table, 1.25e9 one GPU's share."""
    from paddlerec_amd import ops
    rng = np.random.default_rng(N % 1000)
    keys = rng.integers(0, 2 ** 64, size=20011, dtype=np.uint64)
    keys[:8] = [0, 1, 2 ** 63, 2 ** 64 - 1, 2 ** 63 + 5, 12345678901234567890, N, N - 1]
    got = ops.feasign_rows(T(keys.view(np.int64)), N).cpu().numpy()
    assert np.array_equal(got, M.feasign_rows(keys, N))
    assert got[0] == 0 and got[1:].min() >= 1 and got.max() < N


@pytest.mark.gpu
def test_multislot_feasign_keys_oob_and_long_segments(engine_lib):
    from paddlerec_amd import _lib, ops
    rng = np.random.default_rng(5)
    B, S, N, D = 100, 6, 777, 9
    samples = _random_problem(rng, B, S, N, max_len=40, feasigns=True)      # segments far longer than a sub-step
    values, lod, base = M.csr_from_samples(samples, S)
    Wfull = rng.standard_normal((N, 16)).astype(np.float32)
    mbatch = ops.MultislotBatch(T(values), T(lod), T(base))
    out, counts, seg, rows, status = ops.multislot_sumpool(mbatch, T(Wfull)[:, :D], N, 0, 1)
    want, wcnt, wseg, wrows = M.multislot_sumpool(values, lod, base, Wfull[:, :D], 0, 1, N)
    assert np.array_equal(counts.cpu().numpy(), wcnt) and np.array_equal(rows.cpu().numpy()[: len(values)], wrows)
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-5, atol=2e-6)
    assert np.array_equal(ops.feasign_rows(T(values), N).cpu().numpy(), M.feasign_rows(values.astype(np.uint64), N))
    assert int(status.item()) == 0
    # key_mode 0 with an id outside the table: defined behaviour = skipped + flagged (Paddle raises [EXT])
    bad = np.array([1, N + 5, 2, -3], np.int64)
    mb2 = ops.MultislotBatch(T(bad), T(np.array([[0, 2, 4]], np.int64)), T(np.array([0, 4], np.int64)))
    out2, c2, _, _, st2 = ops.multislot_sumpool(mb2, T(Wfull)[:, :D], N, 0, 0)
    assert int(st2.item()) & _lib.REC_FLAG_INDEX_OOB
    assert np.array_equal(c2.cpu().numpy(), [[1], [1]])
    np.testing.assert_allclose(out2.cpu().numpy(), Wfull[[1, 2], :D], rtol=1e-6)


@pytest.mark.gpu
def test_reference_demo_file_through_parser_and_one_launch(engine_lib):
    """The reference's own multi-value fixture (first lines of slot_dnn/data/demo_10): host parser -> device CSR ->
    ONE launch for the 300 slots == the per-slot sparse_embedding + sequence_pool of slot_dnn/net.py:63-75."""
    from paddlerec_amd import ops, reader
    data = open(os.path.join(GOLDEN, "slot_dnn_demo_4.txt"), "rb").read()
    N, D = 100003, 9
    values, lod, base, n = reader.parse_feasign_slots(data, 2, 300, 0)     # hash_rows 0: raw uint64 bit patterns
    rng = np.random.default_rng(1)
    Wfull = rng.standard_normal((N, 16)).astype(np.float32)
    Wfull[0] = 0
    mbatch = ops.MultislotBatch(T(values.numpy()), T(lod.numpy()), T(base.numpy()))
    out, counts, seg, rows, status = ops.multislot_sumpool(mbatch, T(Wfull)[:, :D], N, 0, 1)
    want, wcnt, _, wrows = M.multislot_sumpool(values.numpy(), lod.numpy(), base.numpy(), Wfull[:, :D], 0, 1, N)
    assert np.array_equal(counts.cpu().numpy(), wcnt)
    assert np.array_equal(rows.cpu().numpy()[: len(wrows)], wrows)
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-5, atol=2e-6)
    assert int(counts.max().item()) > 4 and int(status.item()) == 0


@pytest.mark.gpu
def test_multislot_full_size_properties(engine_lib):
    """B = 65536 x 408 slots x D 9 (the slot_dnn benchmark shape): size-independent properties — pooled counts sum to
    the number of non-padding ids, every segment equals a torch index_add of the same rows on the same device,
    seg/rows consistent with the CSR."""
    from paddlerec_amd import ops
    B, S, D, N = 65536, 408, 9, 1_000_003
    g = torch.Generator(device=DEV).manual_seed(3)
    lens = torch.randint(1, 3, (S, B), device=DEV, generator=g)                    # 1-2 ids per (slot, sample)
    lod = torch.zeros(S, B + 1, dtype=torch.int64, device=DEV)
    lod[:, 1:] = torch.cumsum(lens, dim=1)
    per_slot = lod[:, -1]
    base = torch.zeros(S + 1, dtype=torch.int64, device=DEV)
    base[1:] = torch.cumsum(per_slot, 0)
    nnz = int(base[-1].item())
    values = torch.randint(1, N, (nnz,), device=DEV, generator=g, dtype=torch.int64)
    values[torch.rand(nnz, device=DEV, generator=g) < 0.2] = 0
    rec = torch.randn(N, 16, device=DEV, generator=g)
    mbatch = ops.MultislotBatch(values, lod, base)
    out, counts, seg, rows, status = ops.multislot_sumpool(mbatch, rec[:, :D], N, 0, 0)
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    assert int(counts.sum().item()) == int((values != 0).sum().item())
    # reference on the same device: segment id of every value from the CSR, index_add of the gathered rows
    sl = torch.repeat_interleave(torch.arange(S, device=DEV), per_slot)
    within = torch.arange(nnz, device=DEV) - base[sl]
    samp = torch.searchsorted(lod, within.view(S, -1) if False else within.unsqueeze(0).expand(1, -1).contiguous()
                              .view(-1, 1).squeeze(1).unsqueeze(0), right=True) if False else None
    # (per-slot searchsorted: rows of lod are the sorted offset tables)
    samp = torch.empty(nnz, dtype=torch.int64, device=DEV)
    for s in range(S):
        a, b = int(base[s].item()), int(base[s + 1].item())
        samp[a:b] = torch.searchsorted(lod[s], torch.arange(b - a, device=DEV), right=True) - 1
    want_seg = (samp * S + sl).to(torch.int32)
    assert torch.equal(seg[:nnz], want_seg)
    assert torch.equal(rows[:nnz], values)
    ref = torch.zeros(B * S, D, device=DEV)
    live = values != 0
    ref.index_add_(0, want_seg[live].long(), rec[values[live], :D])
    torch.testing.assert_close(out.view(B * S, D), ref, rtol=1e-5, atol=1e-5)
    cref = torch.zeros(B * S, dtype=torch.int32, device=DEV)
    cref.index_add_(0, want_seg[live].long(), torch.ones(int(live.sum()), dtype=torch.int32, device=DEV))
    assert torch.equal(counts.view(-1), cref)


@pytest.mark.gpu
def test_layer_gpu_vs_golden_and_oracle(engine_lib):
    from paddlerec_amd import ops
    _check_layer_step(DEV, None, ops.MultislotBatch)


@pytest.mark.gpu
def test_layer_ps_accessor_gpu(engine_lib):
    from paddlerec_amd import ops
    _check_ps_layer(DEV, None, ops.MultislotBatch)


@pytest.mark.gpu
@pytest.mark.parametrize("opt", ["ps", "adam"])
def test_padded_pool_stride_equals_the_dense_layout(engine_lib, monkeypatch, opt):
    """S x D = 88 x 9 = 792 inputs: the layer pools into a [B, 800] buffer (rec_multislot_desc.out_stride), runs layer 0 on
    its weight with 8 zero rows behind it and hands the row updates rec_grad_layout{group S, stride 800} — against the dense
    layout (REC_SLOT_PAD0=0) over three steps: same pooled values and counts bit for bit, predictions / losses at fp32
    rounding of the GEMM's K order, the table and the dense parameters inside the optimizer's bar, shapes unchanged."""
    from helpers import assert_adam_weights_close
    from paddlerec_amd import ops
    from paddlerec_amd.slot_dnn import BenchmarkDNNLayer
    B, S, N, D, lr, steps = 200, 88, 5000, 9, 1e-2, 3
    rng = np.random.default_rng(7)
    batches = []
    for _ in range(steps):
        values, lod, base = M.csr_from_samples(_random_problem(rng, B, S, N, max_len=4), S)
        batches.append((values, lod, base, (rng.random((B, 1)) < 0.3).astype(np.int64)))
    accp = dict(lr=0.05, initial_g2sum=3.0, bounds=(-10.0, 10.0), initial_range=1e-2, embedx_threshold=1.5,
                nonclk_coeff=0.1, click_coeff=1.0, seed=99)
    runs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("REC_SLOT_PAD0", mode)
        torch.manual_seed(6)
        m = BenchmarkDNNLayer(N, D, S, [64, 32], device=DEV, key_mode=0, sparse_optimizer=opt,
                              accessor=accp if opt == "ps" else None)
        assert m.padded == (mode == "1") and m.in0 == 792 and m.ld0 == (800 if m.padded else 792)
        assert tuple(m.state_dict()["linear_0.weight"].shape) == (792, 64)
        outs = []
        for values, lod, base, label in batches:
            mbatch = ops.MultislotBatch(T(values), T(lod), T(base))
            loss, pred = m.train_step(mbatch, T(label), lr=lr)
            outs.append((float(loss), pred.cpu().numpy().copy(), m.last_counts.cpu().numpy().copy()))
        ev = m.forward(mbatch).cpu().numpy()
        if m.padded:
            o = m.dense.offsets["linear_0.weight"]
            for buf in (m.dense.data, m.dense.grad, m.dense.m, m.dense.v):
                assert float(buf[o + 792 * 64: o + 800 * 64].abs().max()) == 0.0
        runs[mode] = (outs, m.rec.cpu().numpy().copy(), {k: v.cpu().numpy().copy() for k, v in m.dense.p.items()}, ev)
        assert int(m.status.item()) == 0
    (oa, ra, da, ea), (ob, rb, db_, eb) = runs["1"], runs["0"]
    for (la, pa, ca), (lb, pb, cb) in zip(oa, ob):
        assert abs(la - lb) <= 2e-6 * max(abs(lb), 1e-3) and np.array_equal(ca, cb)
        np.testing.assert_allclose(pa, pb, rtol=0, atol=5e-6)
    np.testing.assert_allclose(ea, eb, rtol=0, atol=5e-6)
    if opt == "ps":        # the whole record: counters and states exact, weights / g2sums at the merged gradient's rounding
        wscale = float(np.abs(rb[:, :D]).max())
        np.testing.assert_allclose(ra, rb, rtol=2e-5, atol=2e-5 * wscale)
    else:
        assert_adam_weights_close(ra[:, :D], rb[:, :D], lr, steps, err_msg="embedding")
    for k in da:
        assert_adam_weights_close(da[k], db_[k], lr, steps, err_msg=k)


@pytest.mark.gpu
def test_ps_shrink_rows(engine_lib):
    from oracle import ps_ref
    from paddlerec_amd import ops
    rng = np.random.default_rng(11)
    N, D = 500, 9
    tbl = ops.PsTable(N, D, DEV, kind="slot")
    L = tbl.layout
    rec = np.zeros((N, L.row_stride), np.float32)
    born = rng.random(N) < 0.7
    rec[born, :D] = rng.standard_normal((int(born.sum()), D))
    rec[born, L.stat_off] = rng.integers(1, 30, int(born.sum()))
    rec[born, L.stat_off + 1] = np.minimum(rec[born, L.stat_off], rng.integers(0, 3, int(born.sum())))
    rec[born, L.stat_off + 4] = rng.integers(1, 3, int(born.sum()))
    rec[born, L.stat_off + 5] = rng.random(int(born.sum()))                       # delta_score
    rec[born, L.stat_off + 6] = rng.integers(0, 40, int(born.sum()))              # unseen_days: some > 30
    tbl.rec.copy_(T(rec))
    deleted = ops.ps_shrink_rows(tbl, 0.98, 0.8, 30.0)
    lay = dict(embed_off=L.embed_off, embedx_off=L.embedx_off, embedx_dim=L.embedx_dim, stat_off=L.stat_off)
    acc = dict(nonclk_coeff=0.1, click_coeff=1.0)
    want = rec.copy()
    wdel = ps_ref.shrink_rows(want, lay, acc, 0.98, 0.8, 30.0)
    only_score = ps_ref.shrink_rows(rec.copy(), lay, acc, 0.98, 0.8)
    assert deleted == wdel and wdel > only_score > 0            # both deletion causes occur
    np.testing.assert_allclose(tbl.rec.cpu().numpy(), want, rtol=1e-6)
    # Save(param) + UpdateStatAfterSave(param): delta, base, daily — masks and statistics against the oracle
    for param in (1, 2, 3, 0):
        sel = ops.ps_save_select(tbl, param, 1.5, 0.25, 16.0).cpu().numpy()
        wsel = ps_ref.save_select(want, lay, acc, param, 1.5, 0.25, 16.0)
        assert np.array_equal(sel, wsel) and (param != 1 or 0 < wsel.sum() < (want[:, L.stat_off + 4] != 0).sum())
        assert np.array_equal(tbl.rec.cpu().numpy(), want)


@pytest.mark.gpu
def test_sparse_adam_record_equals_two_row_passes(engine_lib):
    """rec_sparse_adam_record == rec_sparse_adam_rows on W + rec_sparse_adam_rows on W1, bit for bit."""
    from paddlerec_amd import ops
    rng = np.random.default_rng(2)
    B, S, N, D = 3000, 26, 5000, 16
    ids = rng.integers(0, N, (B, S), dtype=np.int64)
    ids[:, :3] = rng.integers(1, 20, (B, 3))                      # hot rows: long segments through the partials path
    grad = (rng.standard_normal((B * S, D)) * 0.1).astype(np.float32)
    dz = (rng.standard_normal((B, 1)) * 0.1).astype(np.float32)
    rec0 = rng.standard_normal((N, 32)).astype(np.float32) * 0.1
    rec0[:, D + 2] = np.abs(rec0[:, D + 2])                       # v1 >= 0
    mv0 = np.abs(rng.standard_normal((N, 32)).astype(np.float32)) * 0.01
    ws = ops.Workspace(DEV)
    groups, _ = ops.ids_group(T(ids), N, 0, ws)
    tg, tdz = T(grad), T(dz)
    pp = ops.segment_partials(groups, tg, D)
    pp1 = ops.segment_partials(groups, tdz, 1, grad_div=S)
    a_rec, a_mv = T(rec0), T(mv0)
    ops.sparse_adam_rows(groups, tg, 1, a_rec[:, :D], a_mv[:, :D], a_mv[:, D:2 * D], 3, 1e-3, partials=pp)
    ops.sparse_adam_rows(groups, tdz, S, a_rec[:, D:D + 1], a_rec[:, D + 1:D + 2], a_rec[:, D + 2:D + 3], 3, 1e-3,
                         partials=pp1)
    b_rec, b_mv = T(rec0), T(mv0)
    ops.sparse_adam_record(groups, tg, tdz, S, b_rec, b_mv, D, 3, 1e-3, partials=pp, partials1=pp1)
    assert torch.equal(a_rec, b_rec) and torch.equal(a_mv, b_mv)
    assert not torch.equal(b_rec, T(rec0))


@pytest.mark.gpu
def test_ps_pull_group_push_full_size_properties(engine_lib):
    """configs[4]'s model shape at full size (B 65536 x 408 slots x D 9, ~40 M values, lazily born accessor table):
    pull (one-launch pool on unborn rows) -> grouping with the segment payload -> lane-per-row accessor push, checked
    against torch on the same device: show / click counters bit-exact per row (bincount), merged gradients and the
    AdaGrad step of every touched row (index_add), rows born with exactly the values the pull showed, untouched rows
    still zero memory."""
    from paddlerec_amd import ops
    B, S, D, N = 65536, 408, 9, 3_000_017
    g = torch.Generator(device=DEV).manual_seed(11)
    lens = torch.randint(1, 3, (S, B), device=DEV, generator=g)                    # 1-2 ids per (slot, sample)
    lod = torch.zeros(S, B + 1, dtype=torch.int64, device=DEV)
    lod[:, 1:] = torch.cumsum(lens, dim=1)
    base = torch.zeros(S + 1, dtype=torch.int64, device=DEV)
    base[1:] = torch.cumsum(lod[:, -1], 0)
    nnz = int(base[-1].item())
    values = torch.randint(1, N, (nnz,), device=DEV, generator=g, dtype=torch.int64)
    values[torch.rand(nnz, device=DEV, generator=g) < 0.3] = 0
    # embed_zero_init off: a key that does not exist reads as its embed_w creation value (embedx as 0) without the
    # table being written; the first push stores that embed_w and — threshold 0 — creates embedx at its end
    table = ops.PsTable(N, D, DEV, kind="slot", embedx_threshold=0.0, initial_range=1e-2, embed_zero_init=False)
    mbatch = ops.MultislotBatch(values, lod, base)
    status = ops.new_status(DEV)
    out, counts, seg, rows, _ = ops.multislot_sumpool(mbatch, table.W, N, 0, 0, status, lazy_init=table.lazy_init)
    # creation values of every row, as a pull shows them: N samples x 1 slot, one id each, on an untouched table
    fresh = ops.PsTable(N, D, DEV, kind="slot", embedx_threshold=0.0, initial_range=1e-2, embed_zero_init=False)
    allrows = torch.arange(N, device=DEV)
    one = ops.MultislotBatch(allrows, torch.arange(N + 1, device=DEV).view(1, -1).contiguous(),
                             torch.tensor([0, N], device=DEV))
    init, _, _, _, _ = ops.multislot_sumpool(one, fresh.W, N, None, 0, status, lazy_init=fresh.lazy_init,
                                             want_backward=False, want_counts=False)
    assert float(init.abs().max().item()) <= 1e-2 and bool((init[:, 1:] == 0).all())
    assert float(init[1:, 0].abs().max().item()) > 0
    # Paddle's default (zero_init): the same pull on a table that does not exist yet returns zeros
    ztab = ops.PsTable(N, D, DEV, kind="slot", embedx_threshold=0.0, initial_range=1e-2)
    zout, _, _, _, _ = ops.multislot_sumpool(one, ztab.W, N, None, 0, status, lazy_init=ztab.lazy_init,
                                             want_backward=False, want_counts=False)
    assert not bool(zout.any())
    live = values != 0
    r, sg = values[live], seg[:nnz][live].long()
    # the pooled output IS the sum of creation values of the live ids of a cell
    ref_out = torch.zeros(B * S, D, device=DEV)
    ref_out.index_add_(0, sg, init[r])
    torch.testing.assert_close(out.view(B * S, D), ref_out, rtol=1e-5, atol=1e-7)
    groups = ops.IdGroups(nnz, DEV)
    ops.ids_group(rows[:nnz], N, 0, ops.Workspace(DEV), None, status, groups, payload=seg[:nnz])
    dx = torch.randn(B, S * D, device=DEV, generator=g) * 1e-3
    label = torch.randint(0, 2, (B,), device=DEV, generator=g)
    ops.ps_push_rows(table, groups, dx, S, click=label)
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    rec = table.rec
    L = table.layout
    show = torch.bincount(r, minlength=N)
    click = torch.bincount(r, weights=label[sg // S].double(), minlength=N)
    assert torch.equal(rec[:, L.stat_off].long(), show)
    assert torch.equal(rec[:, L.stat_off + 1].double(), click)
    touched = show > 0
    assert torch.equal(rec[:, L.stat_off + 4], torch.where(touched, 2.0, 0.0).float())      # embedx created / no key
    assert bool((rec[~touched] == 0).all())                                                 # still zero memory
    gsum = torch.zeros(N, D, device=DEV, dtype=torch.float64)
    gsum.index_add_(0, r, dx.view(B * S, D)[sg].double())
    a = table.accessor
    scaled = gsum[:, 0] / show.clamp(min=1).double()                    # the rule divides by the pushed show
    want_w = (init[:, 0].double() - a.lr * scaled).clamp(a.min_bound, a.max_bound)           # g2sum = 0: ratio 1
    torch.testing.assert_close(rec[touched][:, 0].double(), want_w[touched], rtol=1e-6, atol=1e-9)
    torch.testing.assert_close(rec[touched][:, L.stat_off + 2].double(), scaled[touched] ** 2, rtol=1e-5, atol=1e-14)
    # embedx: created at the END of this first push (score (show-click)*0.1 + click >= 0), its gradient dropped:
    # uniform(+-x_initial_range) creation values, embedx_g2sum = 0
    ex = rec[touched][:, 1:D]
    assert float(ex.abs().max().item()) <= 1e-2 and float(ex.abs().mean().item()) > 2e-3
    assert not bool(rec[:, L.stat_off + 3].any())
    want_delta = (show.float() - click.float()) * 0.1 + click.float()
    torch.testing.assert_close(rec[:, L.stat_off + 5], want_delta, rtol=1e-6, atol=0)
    assert not bool(rec[:, L.stat_off + 6].any())
