"""rec_dcn_v2_train_step (csrc/dcn_v2_step.hip): the whole DCN-v2 train step behind ONE C-ABI call — the per-batch body
of tools/trainer.py:148-152 for models/rank/dcn_v2 — against the Python mirror's eager step
(paddlerec_amd/dcn_v2.py:train_step), which issues the same entry points one by one: every loss, prediction, dense
gradient, parameter, Adam moment and AUC bucket bit-identical, for the four structures of dcn_v2/net.py:110-137
(stacked / parallel x CrossNetV2 / CrossNetMix), with and without the train-mode dropout + L2Decay, with and without
global-norm clipping, at batches on both sides of the fused one-logit head's limit (B >= 64)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(stacked, mix, drop, layers, **kw):
    from paddlerec_amd.dcn_v2 import DCN_V2Layer
    return DCN_V2Layer(sparse_feature_number=kw.get("N", 3000), sparse_feature_dim=kw.get("D", 9), dense_feature_dim=13,
                       sparse_num_field=kw.get("S", 26), layer_sizes=layers, cross_num=kw.get("L", 2), is_Stacked=stacked,
                       use_low_rank_mixture=mix, low_rank=8, num_experts=3, device=DEV,
                       dropout_rate=0.5 if drop else 0.0, l2_dnn=1e-4 if drop else 0.0)


def _batch(rng, B, S, N):
    ids = rng.integers(0, N, (B, S)).astype(np.int64)
    ids[rng.random((B, S)) < 0.1] = 0                       # padding rows
    ids[:, 0] = 7                                           # a hot row: one long segment for the tile partials
    dense = rng.random((B, 13)).astype(np.float32)
    label = (rng.random((B, 1)) < 0.4).astype(np.int64)
    return [torch.as_tensor(x).to(DEV) for x in (ids, dense, label)]


@pytest.mark.parametrize("stacked", [True, False])
@pytest.mark.parametrize("mix", [False, True])
@pytest.mark.parametrize("drop", [False, True])
@pytest.mark.parametrize("B,layers,clip", [(512, [64, 32], 10.0), (40, [48, 20, 12], 0.0)])
def test_c_step_equals_the_mirror_bit_for_bit(engine_lib, stacked, mix, drop, B, layers, clip):
    rng = np.random.default_rng(B + 2 * stacked + 4 * mix + 8 * drop)
    a, b = _model(stacked, mix, drop, layers), _model(stacked, mix, drop, layers)
    b.set_dict({k: v.clone() for k, v in a.state_dict().items()})
    stats_a = [torch.zeros(4096, dtype=torch.int64, device=DEV) for _ in range(2)]
    stats_b = [torch.zeros(4096, dtype=torch.int64, device=DEV) for _ in range(2)]
    os.environ["REC_STEP_PLAN"] = "0"            # the mirror's eager step (the replayed plan is tested against it elsewhere)
    try:
        for step in range(4):
            ids, dense, label = _batch(rng, B, 26, 3000)
            la, pa = a.train_step(ids, dense, label, lr=1e-2, clip_norm=clip, auc_stats=stats_a)
            lb, pb = b.train_step_c(ids, dense, label, lr=1e-2, clip_norm=clip, auc_stats=stats_b)
            assert torch.equal(la, lb), (step, float(la), float(lb))
            assert torch.equal(pa, pb), step
            assert torch.equal(a.dense.grad, b.dense.grad), step
    finally:
        os.environ.pop("REC_STEP_PLAN", None)
    assert a.step_count == b.step_count == 4
    for k, v in a.state_dict().items():
        assert torch.equal(v, b.state_dict()[k]), k
    for x, y in ((a.dense.m, b.dense.m), (a.dense.v, b.dense.v), (a.sparse_state["mv"], b.sparse_state["mv"]),
                 (stats_a[0], stats_b[0]), (stats_a[1], stats_b[1])):
        assert torch.equal(x, y)
    assert float(la) == float(la) and int(a.status.item()) == 0 and int(b.status.item()) == 0


def test_c_step_flags_an_out_of_range_id_like_the_mirror(engine_lib):
    m = _model(True, False, False, [32, 16])
    rng = np.random.default_rng(3)
    ids, dense, label = _batch(rng, 64, 26, 3000)
    ids[5, 3] = 3000
    m.train_step_c(ids, dense, label)
    torch.cuda.synchronize()
    assert int(m.status.item()) != 0


def test_c_step_argument_checks(engine_lib):
    import ctypes as C
    from paddlerec_amd import _lib
    m = _model(False, True, True, [32, 16])
    net = m.c_net()
    nb = C.c_size_t(0)
    L = _lib.lib()
    assert L.rec_dcn_v2_train_step_workspace_bytes(C.byref(net), 64, C.byref(nb)) == 0 and nb.value > 0
    assert L.rec_dcn_v2_train_step_workspace_bytes(C.byref(net), 0, C.byref(nb)) != 0          # empty batch
    net.n_dnn = _lib.DCN_MAX_LAYERS + 1
    assert L.rec_dcn_v2_train_step_workspace_bytes(C.byref(net), 64, C.byref(nb)) != 0
    assert b"n_dnn" in L.rec_last_error()
    net = m.c_net()
    rng = np.random.default_rng(0)
    ids, dense, label = _batch(rng, 64, 26, 3000)
    loss, pred = torch.empty(1, device=DEV), torch.empty(64, 1, device=DEV)
    small = torch.empty(16, dtype=torch.uint8, device=DEV)
    h = _lib.AdamHyper(1e-3, 0.9, 0.999, 1e-8, 1)
    args = [C.c_void_p(t.data_ptr()) for t in (ids, dense, label)]
    rc = L.rec_dcn_v2_train_step(C.byref(net), 64, *args, C.byref(h), None, None, 4095, C.c_void_p(loss.data_ptr()),
                                 C.c_void_p(pred.data_ptr()), C.c_void_p(m.status.data_ptr()),
                                 C.c_void_p(small.data_ptr()), C.c_size_t(16), None)
    assert rc != 0 and b"workspace" in L.rec_last_error()
    net.fc_w = None
    rc = L.rec_dcn_v2_train_step(C.byref(net), 64, *args, C.byref(h), None, None, 4095, C.c_void_p(loss.data_ptr()),
                                 C.c_void_p(pred.data_ptr()), C.c_void_p(m.status.data_ptr()),
                                 C.c_void_p(small.data_ptr()), C.c_size_t(16), None)
    assert rc != 0 and b"NULL" in L.rec_last_error()
