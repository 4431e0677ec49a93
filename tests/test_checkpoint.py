"""Checkpoint round trip (paddlerec_amd/checkpoint.py; reference: tools/utils/save_load.py:25-47)."""
import os
import pickle

import numpy as np
import pytest
import torch

from paddlerec_amd import checkpoint as ck


class _Net:
    """Minimal stand-in with the state_dict/set_dict surface of the host mirrors (CPU, no kernels)."""

    def __init__(self, n=10, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.params = {"fm.embedding.weight": torch.randn(n, 4, generator=g),
                       "fm.embedding_one.weight": torch.randn(n, 1, generator=g),
                       "dnn.linear_0.weight": torch.randn(8, 3, generator=g)}
        self.step_count = 7

    def state_dict(self):
        return self.params

    def set_dict(self, sd):
        for k, v in sd.items():
            self.params[k].copy_(torch.as_tensor(v).reshape(self.params[k].shape))


def test_round_trip_and_layout(tmp_path):
    a, b = _Net(seed=1), _Net(seed=2)
    d = ck.save_model(a, None, str(tmp_path), 3)
    assert os.path.exists(os.path.join(d, "rec.pdparams")) and os.path.exists(os.path.join(d, "rec.pdopt"))
    raw = pickle.load(open(os.path.join(d, "rec.pdparams"), "rb"))
    assert set(raw) == set(a.params) and raw["fm.embedding.weight"].dtype == np.float32   # plain name -> ndarray dict
    ck.load_model(d, b)
    for k in a.params:
        assert torch.equal(a.params[k], b.params[k])


def test_sharded_checkpoint_reassembles_global_tables(tmp_path):
    class Comm:
        def __init__(self, r, w):
            self.rank, self.world = r, w
    full = _Net(n=11, seed=5)
    world = 3
    for r in range(world):
        shard = _Net(n=4)
        for k in ("fm.embedding.weight", "fm.embedding_one.weight"):
            rows = full.params[k][r::world]
            shard.params[k].zero_()
            shard.params[k][: rows.shape[0]] = rows
        shard.params["dnn.linear_0.weight"] = full.params["dnn.linear_0.weight"].clone()
        shard.comm, shard.global_rows = Comm(r, world), 11
        ck.save_model(shard, None, str(tmp_path), 0)
    out = _Net(n=11, seed=9)
    ck.load_model(os.path.join(str(tmp_path), "0"), out)
    for k in full.params:
        assert torch.equal(out.params[k], full.params[k]), k


@pytest.mark.gpu
def test_deepfm_resume_continues_identically(engine_lib, tmp_path):
    """train 2 steps, save, train 1 more; a fresh model loaded from the checkpoint takes the same third step."""
    from helpers import make_deepfm_problem
    from paddlerec_amd.deepfm import DeepFMLayer
    dev = "cuda"
    pr = make_deepfm_problem(B=128, N=300, D=16, fc=[16], seed=3)
    t = lambda x: torch.as_tensor(x).to(dev)
    m = DeepFMLayer(300, 16, 13, 26, [16], device=dev)
    for _ in range(2):
        m.train_step(t(pr["ids"]), t(pr["dense"]), t(pr["label"]), lr=1e-2)
    d = ck.save_model(m, None, str(tmp_path), 1)
    loss_a, _ = m.train_step(t(pr["ids"]), t(pr["dense"]), t(pr["label"]), lr=1e-2)
    m2 = DeepFMLayer(300, 16, 13, 26, [16], device=dev)
    ck.load_model(d, m2)
    loss_b, _ = m2.train_step(t(pr["ids"]), t(pr["dense"]), t(pr["label"]), lr=1e-2)
    assert torch.equal(loss_a, loss_b)
    for k, v in m.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k]), k


# --------------------------------------------------------------------------------------- PS / gpubox accessor tables
class _FakeComm:
    def __init__(self, r, w):
        self.rank, self.world = r, w


class _PsNet:
    """A net with a row-sharded ops.PsTable ('deepfm' records) and one dense parameter — the surface checkpoint.py uses
    of ShardedDeepFMLayer(table='ps'), without a process group."""

    def __init__(self, rank, world, n_global=23, D=4):
        from paddlerec_amd.ops import PsTable
        self.comm, self.global_rows = _FakeComm(rank, world), n_global
        self.ps = PsTable((n_global + world - 1) // world, D, "cpu", kind="deepfm", row_mul=world, row_add=rank)
        self.dense_w = torch.zeros(3, 2)
        self.step_count = 0
        self.sparse_state = None

    def state_dict(self):
        D = self.ps.emb_dim
        return {"fm.embedding.weight": self.ps.rec[:, :D], "fm.embedding_one.weight": self.ps.rec[:, D:D + 1],
                "dnn.linear_0.weight": self.dense_w}

    def set_dict(self, sd):
        assert not any(k.startswith("fm.") or k.startswith("ps.") for k in sd), "tables must not come back dense"
        for k, v in sd.items():
            self.dense_w.copy_(torch.as_tensor(v))

    def _ensure_sparse_state(self):
        raise AssertionError("a PS table has no Adam row state: nothing must be allocated for it")


def _fill(nets, seed=0):
    """Some existing values (global rows 1, 4, 5, 9, 22) with recognisable records; the rest stays 'no such key'."""
    world = len(nets)
    g = torch.Generator().manual_seed(seed)
    want = {}
    for row in (1, 4, 5, 9, 22):
        t = nets[row % world].ps
        rec = torch.randn(t.rec.shape[1], generator=g)
        rec[t.state_col] = 2.0 if row % 2 else 1.0
        t.rec[row // world] = rec
        want[row] = rec.clone()
    for n in nets:
        n.dense_w.fill_(3.5)
        n.step_count = 11
    return want


@pytest.mark.parametrize("old_world,new_world", [(2, 2), (2, 3), (3, 1), (1, 2)])
def test_ps_table_checkpoint_keeps_whole_records_and_reshards(tmp_path, old_world, new_world, monkeypatch):
    """a PS table used to be saved through the dense fm.embedding views — show / click / g2sum / state were
    lost and every loaded row came back as 'no such key'.  Now: existing values only, whole records, any world size."""
    from paddlerec_amd.deepfm import DeepFMLayer
    monkeypatch.setattr(DeepFMLayer, "set_dict", lambda self, sd: self.set_dict(sd), raising=False)
    old = [_PsNet(r, old_world) for r in range(old_world)]
    want = _fill(old)
    for n in old:
        d = ck.save_model(n, None, str(tmp_path), 0)
    raw = pickle.load(open(os.path.join(d, "rec.pdparams" if old_world == 1 else "rec.shard0of%d.pdparams" % old_world), "rb"))
    assert "fm.embedding.weight" not in raw and raw["ps.records"].shape[1] == old[0].ps.rec.shape[1]
    assert set(int(x) for x in raw["ps.rows"]) == {r for r in want if r % old_world == 0}
    new = [_PsNet(r, new_world) for r in range(new_world)]
    for n in new:
        n.ps.rec.fill_(9.0)                       # stale content must be gone after the load
        if new_world == 1:
            n.comm = None
        ck.load_model(d, n)
    for n in new:
        assert float(n.dense_w[0, 0]) == 3.5
    for row in range(23):
        rec = new[row % new_world].ps.rec[row // new_world]
        if row in want:
            assert torch.equal(rec, want[row]), row
        else:
            assert not rec.any(), row


def test_ps_shards_into_an_adam_table_net_are_refused(tmp_path, monkeypatch):
    """_load_sharded used to skip every table key when an adam-table net was given PS shards (ps.rows /
    ps.records, no fm.embedding.*) and return with the tables at their initial values.  Both directions raise now."""
    from paddlerec_amd.deepfm import DeepFMLayer
    monkeypatch.setattr(DeepFMLayer, "set_dict", lambda self, sd: self.set_dict(sd), raising=False)
    old = [_PsNet(r, 2) for r in range(2)]
    _fill(old)
    for n in old:
        d = ck.save_model(n, None, str(tmp_path), 0)

    class _AdamNet:                                   # row-sharded Adam tables: no `ps`, dense embedding views
        def __init__(self, rank, world, n_global=23, D=4):
            self.comm, self.global_rows = _FakeComm(rank, world), n_global
            rows = (n_global + world - 1) // world
            self.W, self.W1, self.dense_w = torch.ones(rows, D), torch.ones(rows, 1), torch.zeros(3, 2)
            self.step_count, self.sparse_state = 0, None

        def state_dict(self):
            return {"fm.embedding.weight": self.W, "fm.embedding_one.weight": self.W1, "dnn.linear_0.weight": self.dense_w}

        def set_dict(self, sd):
            for k, v in sd.items():
                self.state_dict()[k].copy_(torch.as_tensor(v))

        def _ensure_sparse_state(self):
            pass

    for world in (2, 3):
        net = _AdamNet(0, world)
        with pytest.raises(ValueError, match="PS-table run"):
            ck.load_model(d, net)
        assert bool((net.W == 1).all())               # nothing half-loaded into the tables


def test_flat_params_reserved_rows_stay_out_of_the_packed_image():
    """_FlatParams(reserve=...) keeps room behind a tensor (the zero rows of a padded layer-0 weight live there, in all four
    buffers): offsets stay 256-byte aligned, the gap is zero, and the gap-free checkpoint image (packed / load_packed) is the
    one of an unreserved layout."""
    import torch
    from paddlerec_amd.deepfm import _FlatParams
    shapes = [("a", (13,)), ("w0", (390, 80)), ("b0", (80,)), ("w1", (80, 1))]
    plain = _FlatParams(shapes, "cpu")
    padded = _FlatParams(shapes, "cpu", reserve={"w0": 400 * 80})
    assert padded.offsets["a"] == plain.offsets["a"] and padded.offsets["w0"] == plain.offsets["w0"]
    assert padded.offsets["b0"] >= padded.offsets["w0"] + 400 * 80 and all(o % 64 == 0 for o in padded.offsets.values())
    g = torch.Generator().manual_seed(0)
    for n in plain.names:
        v = torch.rand(plain.shapes[n], generator=g)
        plain.p[n].copy_(v)
        padded.p[n].copy_(v)
    o = padded.offsets["w0"]
    assert float(padded.data[o + 390 * 80: o + 400 * 80].abs().max()) == 0.0
    view = padded.data[o: o + 400 * 80].view(400, 80)               # what the GEMMs see
    assert torch.equal(view[:390], plain.p["w0"]) and float(view[390:].abs().max()) == 0.0
    assert torch.equal(padded.packed(padded.data), plain.packed(plain.data))
    fresh = _FlatParams(shapes, "cpu", reserve={"w0": 400 * 80})
    fresh.load_packed(fresh.data, plain.packed(plain.data))
    assert torch.equal(fresh.data, padded.data)
