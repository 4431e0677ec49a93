"""rec_din_train_step (csrc/din_step.hip): the whole DIN train step behind ONE C-ABI call — the per-batch body of
tools/trainer.py:148-152 for models/rank/din — against the Python mirror's step (paddlerec_amd/din.py:_step), which
issues the same entry points one by one: every loss, prediction and parameter bit-identical, at the reference's batch
size (32: the seven row updates in one launch), with one table above the one-launch merge limit (per-table small
launches) and at a batch whose history tables take the sort-based merge."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV)


def _problem(rng, B, Tn, ni, nc):
    lens = rng.integers(1, Tn + 1, B)
    lens[0] = Tn
    hi = np.zeros((B, Tn), np.int64)
    hc = np.zeros((B, Tn), np.int64)
    for b in range(B):
        hi[b, :lens[b]] = rng.integers(1, ni, lens[b])
        hc[b, :lens[b]] = rng.integers(1, nc, lens[b])
    mask = np.where(np.arange(Tn)[None] < lens[:, None], 0, -1000000000).astype(np.int64)
    ti = rng.integers(1, ni, B).astype(np.int64)
    tc = rng.integers(1, nc, B).astype(np.int64)
    label = (rng.random((B, 1)) < 0.5).astype(np.float32)
    tis, tcs = np.repeat(ti[:, None], Tn, 1), np.repeat(tc[:, None], Tn, 1)
    return [T(x) for x in (hi, hc, ti, tc, label, mask, tis, tcs)]


@pytest.mark.parametrize("B,lens,Ei,Ec", [
    (32, (40, 152, 40), 64, 64),          # din/config.yaml: batch 32 — one rec_sparse_sgd_small_multi launch
    (7, (5, 1, 9), 8, 8),                 # a shape whose attention forward saves no layer-1 activations
    (200, (100, 60), 64, 64),             # 20 000 history lookups: sort-based merge for the history tables
    (120, (120,), 32, 96),                # 14 400 lookups: still the one-launch merge, uneven item / category widths
    (16000, (2,), 16, 16)])               # the per-sample tables above the one-launch merge too (their own buffer set)
def test_c_step_equals_the_mirror_bit_for_bit(engine_lib, B, lens, Ei, Ec):
    from paddlerec_amd.din import DINLayer
    rng = np.random.default_rng(B + Ei)
    ni, nc = 900, 60
    a = DINLayer(Ei, Ec, "sigmoid", False, True, ni, nc, device=DEV)
    b = DINLayer(Ei, Ec, "sigmoid", False, True, ni, nc, device=DEV)
    with torch.no_grad():
        a.params["item_b_attr.weight"].copy_(T((rng.standard_normal((ni, 1)) * 0.1).astype(np.float32)))
    b.set_dict({k: v.clone() for k, v in a.state_dict().items()})
    b.set_attention([w.clone() for w in a.attention_w], [x.clone() for x in a.attention_b])
    import os
    os.environ["REC_STEP_PLAN"] = "0"            # the mirror's eager step (the replayed plan is tested against it elsewhere)
    try:
        for step, Tn in enumerate(lens * 2):
            bt = _problem(rng, B, Tn, ni, nc)
            la, pa = a.train_step(*bt, base_lr=0.5)
            lb, pb = b.train_step_c(*bt, base_lr=0.5)
            assert torch.equal(la, lb), (step, float(la), float(lb))
            assert torch.equal(pa, pb), step
    finally:
        os.environ.pop("REC_STEP_PLAN", None)
    assert a.step_count == b.step_count
    for k, v in a.state_dict().items():
        assert torch.equal(v, b.state_dict()[k]), k
    for k in a._gb:
        assert torch.equal(a._gb[k], b._gb[k]), k
    assert int(a.status.item()) == 0 and int(b.status.item()) == 0


def test_two_stream_schedule_changes_no_bit(engine_lib, monkeypatch):
    """Batches on the sort-based merge group their keys on a side stream and update two tables per stream (REC_DIN_SIDE,
    default on): mirror and C entry, one stream against two."""
    from paddlerec_amd.din import DINLayer
    ni, nc, B, Tn = 900, 60, 256, 80
    runs = {}
    for side in ("0", "1"):
        monkeypatch.setenv("REC_DIN_SIDE", side)
        for entry in ("train_step", "train_step_c"):
            torch.manual_seed(11)
            rng = np.random.default_rng(4)
            m = DINLayer(64, 64, "sigmoid", False, True, ni, nc, device=DEV)
            for _ in range(3):
                loss, pred = getattr(m, entry)(*_problem(rng, B, Tn, ni, nc), base_lr=0.5)
            torch.cuda.synchronize()
            runs[side, entry] = (loss.clone(), pred.clone(), {k: v.clone() for k, v in m.state_dict().items()})
            assert (m._side is not None) == (side == "1")
    ref = runs["0", "train_step"]
    for key, (loss, pred, sd) in runs.items():
        assert torch.equal(loss, ref[0]) and torch.equal(pred, ref[1]), key
        for k, v in sd.items():
            assert torch.equal(v, ref[2][k]), (key, k)


def test_c_step_argument_checks(engine_lib):
    import ctypes as C
    from paddlerec_amd import _lib
    from paddlerec_amd.din import DINLayer
    m = DINLayer(8, 8, "sigmoid", False, True, 50, 9, device=DEV)
    net = m.c_net()
    nb = C.c_size_t(0)
    assert _lib.lib().rec_din_train_step_workspace_bytes(C.byref(net), 4, 6, C.byref(nb)) == 0 and nb.value > 0
    assert _lib.lib().rec_din_train_step_workspace_bytes(C.byref(net), 0, 6, C.byref(nb)) != 0          # empty batch
    rng = np.random.default_rng(0)
    bt = _problem(rng, 4, 6, 50, 9)
    loss, pred = torch.empty(1, device=DEV), torch.empty(4, 1, device=DEV)
    small = torch.empty(16, dtype=torch.uint8, device=DEV)
    rc = _lib.lib().rec_din_train_step(C.byref(net), 4, 6, *[C.c_void_p(t.data_ptr()) for t in
                                                             (bt[0], bt[1], bt[2], bt[3], bt[4], bt[5], bt[6], bt[7])],
                                       C.c_float(0.1), C.c_void_p(loss.data_ptr()), C.c_void_p(pred.data_ptr()),
                                       C.c_void_p(m.status.data_ptr()), C.c_void_p(small.data_ptr()), C.c_size_t(16), None,
                                       None)
    assert rc != 0 and b"workspace" in _lib.lib().rec_last_error()
