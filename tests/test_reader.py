"""Rows R and H: the engine's host parsers (librecengine.so, no GPU needed) against the oracle restatements
of the reference readers — bit-exact labels / ids / hashes, dense values equal after the float32 cast."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import deepfm_ref as R
from oracle import hash_ref


@pytest.fixture(scope="module")
def reader(engine_lib):
    from paddlerec_amd import reader as rd
    return rd


def _oracle_slot(lines, log1p=False):
    lab, ids, dense = [], [], []
    for ln in lines:
        a, b, c = R.parse_slot_line(ln)
        if log1p:
            c = np.log(c.astype(np.float64) + 1).astype(np.float32)
        lab.append(a); ids.append(b); dense.append(c)
    return np.asarray(lab, np.int64), np.stack(ids), np.stack(dense)


def test_slot_text_reference_sample(reader):
    """First lines of the reference's own sample file (models/rank/deepfm/data/sample_data/train)."""
    data = open(os.path.join(GOLDEN, "criteo_slot_sample.txt"), "rb").read()
    label, ids, dense = reader.parse_slot_text(data)
    lines = data.decode().strip().split("\n")
    ol, oi, od = _oracle_slot(lines)
    assert np.array_equal(label.numpy(), ol) and np.array_equal(ids.numpy(), oi)
    assert np.array_equal(dense.numpy(), od)
    assert ids.shape == (len(lines), 26) and int(ids.min()) > 0


def _synthetic_slot_lines(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        toks = []
        if rng.random() > 0.05:
            toks.append("click:%d" % rng.integers(0, 2))
        if rng.random() > 0.1:
            toks += ["dense_feature:%s" % repr(float(np.round(rng.random() * 10.0 ** int(rng.integers(-6, 3)), 9)))
                     for _ in range(13)]
        for s in range(1, 27):
            if rng.random() > 0.1:                      # missing slots exercise the padding branch
                toks.append("%d:%d" % (s, rng.integers(0, 1000001)))
        toks.append("unknown_slot:7")
        out.append(" ".join(toks))
    return out


@pytest.mark.parametrize("threads,log1p", [(1, False), (7, False), (0, True)])
def test_slot_text_synthetic(reader, threads, log1p):
    lines = _synthetic_slot_lines(2000, 3)
    data = ("\n".join(lines) + "\n").encode()
    label, ids, dense = reader.parse_slot_text(data, log1p_dense=log1p, threads=threads)
    ol, oi, od = _oracle_slot(lines, log1p)
    assert label.shape[0] == len(lines)
    assert np.array_equal(label.numpy(), ol)
    assert np.array_equal(ids.numpy(), oi)
    if log1p:       # np.log vs libm log may differ in the last ulp of the double; after the f32 cast <= 1 ulp
        np.testing.assert_allclose(dense.numpy(), od, rtol=2e-7, atol=0)
    else:
        assert np.array_equal(dense.numpy(), od)


def test_slot_text_edge_cases(reader):
    label, ids, dense = reader.parse_slot_text(b"")
    assert label.shape[0] == 0
    label, ids, dense = reader.parse_slot_text(b"click:1 3:42\n\nclick:0 dense_feature:0.5\n")   # ragged / empty line
    assert label.tolist() == [1, 0, 0]
    assert ids[0].tolist() == [0, 0, 42] + [0] * 23 and int(ids[1].sum()) == 0
    assert dense[2].tolist() == [0.5] + [0.0] * 12


def _synthetic_tsv(n, seed):
    rng = np.random.default_rng(seed)
    lines = []
    for _ in range(n):
        f = [str(rng.integers(0, 2))]
        f += ["" if rng.random() < 0.2 else str(rng.integers(-3, 70000)) for _ in range(13)]
        f += ["" if rng.random() < 0.1 else "%08x" % rng.integers(0, 2 ** 32) for _ in range(26)]
        lines.append("\t".join(f))
    return lines


@pytest.mark.parametrize("threads", [1, 5])
def test_criteo_tsv_hashing(reader, threads):
    lines = _synthetic_tsv(1500, 11)
    data = ("\n".join(lines) + "\n").encode()
    label, ids, dense = reader.parse_criteo_tsv(data, threads=threads)
    assert label.shape[0] == len(lines)
    xxhash = pytest.importorskip("xxhash")
    for i in (0, 1, 7, 700, 1499):
        ol, oi, od = hash_ref.criteo_tsv_line(lines[i])
        assert int(label[i]) == ol
        assert np.array_equal(ids[i].numpy(), oi)                 # integer path: bit-exact
        assert np.array_equal(dense[i].numpy(), od)
        feats = lines[i].split("\t")
        assert int(ids[i, 0]) == xxhash.xxh32("14" + feats[14]).intdigest() % 1000001   # the wheel the reference imports
    assert int(ids.min()) >= 0 and int(ids.max()) <= 1000000


def test_file_batches_drop_last(reader, tmp_path):
    lines = _synthetic_slot_lines(25, 5)
    paths = []
    for k in range(2):
        p = tmp_path / ("part-%d" % k)
        p.write_text("\n".join(lines) + "\n")
        paths.append(str(p))
    batches = list(reader.SlotTextReader(paths, 8, device="cpu"))
    assert len(batches) == 50 // 8                                 # remainder carried across files, last dropped
    lab = np.concatenate([b[0].numpy().reshape(-1) for b in batches])
    ol, oi, _ = _oracle_slot(lines + lines)
    assert np.array_equal(lab, ol[:48])
    assert np.array_equal(np.concatenate([b[1].numpy() for b in batches]), oi[:48])


@pytest.mark.parametrize("bs", [4, 1, 7])
def test_din_reader_matches_reference_restatement(reader, bs):
    """Sorting by history length inside 20*bs groups, padding, mask, repeated target — vs oracle/din_ref.py
    on the first 40 lines of the reference's own sample file."""
    from oracle import din_ref
    path = os.path.join(GOLDEN, "din_sample.txt")
    lines = open(path).read().strip().split("\n")
    want = din_ref.reader_batches(lines, bs)
    got = list(reader.DinReader([path], bs, device="cpu"))
    assert len(got) == len(want) and len(got) == (len(lines) // (bs * 20)) * 20 + (len(lines) % (bs * 20)) // bs
    keys = ("hist_item", "hist_cat", "target_item", "target_cat", "label", "mask", "target_item_seq", "target_cat_seq")
    for g, w in zip(got, want):
        for t, k in zip(g, keys):
            assert np.array_equal(t.numpy(), w[k]), k
            assert t.numpy().dtype == w[k].dtype, k
