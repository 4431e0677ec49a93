"""The pipelined DeepFM step (round 6: the chain fm_bwd -> sparse update -> next lookup on ONE stream, the dense tail with
the next step's weight images on the side stream, joined where the next step's first GEMM needs it) and the one-launch
weight images (rec_gemm_b_images / rec_gemm_epilogue_args.b_image) change the SCHEDULE, not one bit of the results:
every parameter, both Adam moment sets and the losses after several steps are identical to the plain two-stream step and
to the step whose GEMMs split their own weights.  Also: rec_gemm_b_images == the per-call split, byte for byte."""
import os

import numpy as np
import pytest
import torch

from helpers import make_deepfm_problem

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _layer(pr, fc, tables):
    from paddlerec_amd.deepfm import DeepFMLayer
    p = pr["params"]
    so = pr["slot_offsets"] if tables else None
    m = DeepFMLayer(pr["N"], pr["D"], pr["Dn"], pr["S"], fc, device=DEV, slot_offset=so)
    sd = {"fm.embedding.weight": p["W"], "fm.embedding_one.weight": p["W1"], "fm.dense_w": p["dense_w"],
          "fm.dense_w_one": p["dense_w_one"]}
    for i in range(len(fc) + 1):
        sd["dnn.linear_%d.weight" % i], sd["dnn.linear_%d.bias" % i] = p["mlp_w"][i], p["mlp_b"][i]
    m.set_dict(sd)
    return m


def _run(pr, batches, fc, tables, pipelined, images):
    os.environ["REC_GEMM_IMAGES"] = "1" if images else "0"
    try:
        m = _layer(pr, fc, tables)
        m.pipelined = pipelined
        losses = []
        for ids, dense, label in batches:
            loss, _ = m.train_step(ids, dense, label, lr=1e-2)
            losses.append(loss)                       # device tensors: no host sync between the steps
        m.sync()
        torch.cuda.synchronize()
        out = {k: v.detach().clone() for k, v in m.state_dict().items()}
        out["dense.m"], out["dense.v"] = m.dense.m.clone(), m.dense.v.clone()
        out["sparse.mv"] = m.sparse_state["mv"].clone()
        out["losses"] = torch.cat([x.reshape(1) for x in losses])
        assert int(m.status.item()) == 0
        return out
    finally:
        os.environ.pop("REC_GEMM_IMAGES", None)


@pytest.mark.parametrize("tables", [True, False])
def test_pipelined_step_and_weight_images_change_no_bit(engine_lib, tables):
    B, fc = 16384, [400, 400, 400]           # the bench's tower at a batch the bf16 x 3 GEMMs and the sort-based merge take
    pr = make_deepfm_problem(B=B, N=3000, D=16, fc=fc, seed=5, tables=tables)
    rng = np.random.default_rng(9)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(DEV)
    batches = []
    for _ in range(5):
        n_ids = 3000
        ids = rng.integers(1, n_ids, size=(B, 26), dtype=np.int64)
        ids[rng.random((B, 26)) < 0.03] = 0
        batches.append((t(ids), t(rng.random((B, 13), dtype=np.float32)), t((rng.random((B, 1)) < 0.25).astype(np.int64))))
    ref = _run(pr, batches, fc, tables, pipelined=False, images=False)
    for pipelined, images in ((False, True), (True, True), (True, False)):
        got = _run(pr, batches, fc, tables, pipelined, images)
        for k in ref:
            assert torch.equal(got[k], ref[k]), (k, pipelined, images)
    assert float(ref["losses"][-1]) < float(ref["losses"][0])


def test_pipelined_step_mixes_with_other_entry_points(engine_lib):
    """forward(), state_dict() and a plain step after pipelined steps see the finished parameters (they join the side
    stream themselves)."""
    B, fc = 16384, [400, 400, 400]
    pr = make_deepfm_problem(B=B, N=3000, D=16, fc=fc, seed=6, tables=True)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(DEV)
    ids, dense, label = t(pr["ids"]), t(pr["dense"]), t(pr["label"])

    def run(pipelined):
        m = _layer(pr, fc, True)
        m.pipelined = pipelined
        m.train_step(ids, dense, label, lr=1e-2)
        m.train_step(ids, dense, label, lr=1e-2)
        p1 = m.forward(ids, dense).clone()
        m.pipelined = False
        m.train_step(ids, dense, label, lr=1e-2)
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
        return p1, sd
    pa, sa = run(False)
    pb, sb = run(True)
    assert torch.equal(pa, pb)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


def test_weight_images_in_one_launch_equal_the_per_call_split(engine_lib):
    """rec_gemm_b_images (all images of a step, one launch) against the image each rec_gemm_f32 call makes for itself:
    the GEMM results are bit-identical for W and W^T operands of the tower's shapes."""
    from paddlerec_amd import ops
    g = torch.Generator().manual_seed(3)
    ws = ops.Workspace(DEV)
    M = 8192
    shapes = [(432, 400), (400, 400), (768, 512)]
    Ws = [torch.randn(k, n, generator=g).to(DEV) for k, n in shapes]
    imgs = ops.GemmImages([(w, False) for w in Ws] + [(w, True) for w in Ws], DEV)
    imgs.refresh()
    for i, w in enumerate(Ws):
        k, n = w.shape
        a = torch.randn(M, k, generator=g).to(DEV)
        assert imgs.get(i) is not None
        assert torch.equal(ops.gemm(a, w, ws, b_image=imgs.get(i)), ops.gemm(a, w, ws))
        gy = torch.randn(M, n, generator=g).to(DEV)
        assert torch.equal(ops.gemm(gy, w, ws, trans_b=True, b_image=imgs.get(len(Ws) + i)), ops.gemm(gy, w, ws, trans_b=True))
    # a shape without an image form is skipped, and the call runs as before
    odd = ops.GemmImages([(torch.randn(50, 3, generator=g).to(DEV), False)], DEV)
    odd.refresh()
    assert odd.get(0) is None
