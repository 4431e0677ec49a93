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


def test_number_forms_match_python(reader):
    """The parsers' bounded fast path for "[-]digits[.digits]" and their C-library fallback give float(x) / int(x) of the
    reference's readers for every form a field can take: short and 15/16/17-digit decimals, long fractions, exponents,
    signs, leading zeros, '.5' / '5.', inf / nan, 18/19-digit integers — as the last token of an unterminated buffer too."""
    rng = np.random.default_rng(5)
    vals = ["0.0", "0.05", "0.00497512437811", "0.207421875", "94.864944714", "7.884e-06", "4.535E-05", "1e3", "-0", "-0.0",
            ".5", "5.", "-.25", "+3.5", "00012.50", "123456789012345", "1234567890123456", "12345678901234567",
            "0.1234567890123456789", "0.000000000000000000012345", "9007199254740993", "0.3", "2.675", "1e22", "1e23",
            "123456789012345678901234567890", "inf", "-inf", "nan", "1.7976931348623157e308", "4.9e-324", "0.1e-1"]
    for _ in range(400):
        sig = int(rng.integers(1, 17))
        digs = "".join(str(d) for d in rng.integers(0, 10, sig))
        cut = int(rng.integers(0, sig + 1))
        v = ("-" if rng.random() < 0.3 else "") + digs[:cut] + ("." + digs[cut:] if rng.random() < 0.8 or cut == 0 else "")
        if v not in ("-", "", "."):
            vals.append(v if any(c.isdigit() for c in v) else "0")
    ints = ["0", "7", "-5", "+5", "007", "737395", "123456789012345678", "-123456789012345678", "9223372036854775807",
            "-9223372036854775808", "1234567890123456789"]
    lines = []
    for i in range(0, len(vals), 13):
        chunk = vals[i:i + 13]
        lines.append("click:%s " % ints[(i // 13) % len(ints)] + " ".join("dense_feature:" + v for v in chunk) +
                     " 1:" + ints[(i // 13 + 3) % len(ints)])
    for data in (("\n".join(lines) + "\n").encode(), "\n".join(lines).encode()):      # with and without a final newline
        buf = np.frombuffer(data, np.uint8).copy()        # exactly len(data) bytes: nothing readable behind the last token
        label, ids, dense = reader.parse_slot_text(buf, n_sparse=1, n_dense=13, threads=2)
        assert label.shape[0] == len(lines)
        for r, i in enumerate(range(0, len(vals), 13)):
            chunk = vals[i:i + 13]
            with np.errstate(over="ignore"):
                want = np.asarray([float(v) for v in chunk], np.float64).astype(np.float32)
            got = dense[r, :len(chunk)].numpy()
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)) or \
                np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(got[~np.isnan(got)], want[~np.isnan(want)]), \
                (chunk, got, want)
            assert int(label[r]) == int(ints[(i // 13) % len(ints)])
            assert int(ids[r, 0]) == int(ints[(i // 13 + 3) % len(ints)])
    # the TSV parser shares the number code: dense fields through (x - min) / diff in float64, then float32
    from paddlerec_amd.reader import CONT_MIN, CONT_DIFF
    tsv = []
    for i in range(0, len(vals) - 13, 13):
        tsv.append("\t".join(["1"] + vals[i:i + 13] + ["ab%02d" % j for j in range(26)]))
    label, ids, dense = reader.parse_criteo_tsv(np.frombuffer("\n".join(tsv).encode(), np.uint8).copy(), threads=2)
    for r, ln in enumerate(tsv):
        ol, oi, od = hash_ref.criteo_tsv_line(ln)
        assert int(label[r]) == ol and np.array_equal(ids[r].numpy(), oi)
        g, w = dense[r].numpy(), np.asarray(od, np.float32)
        assert np.array_equal(np.isnan(g), np.isnan(w)) and np.array_equal(g[~np.isnan(g)], w[~np.isnan(w)]), ln


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


# ------------------------------------------------------------------------------ multi-value slot lines (row P)
def _oracle_csr(lines, first_slot, num_slots, hash_rows):
    per_line = [R.parse_feasign_line(ln, first_slot, num_slots) for ln in lines]
    values, lod, base = [], np.zeros((num_slots, len(lines) + 1), np.int64), np.zeros(num_slots + 1, np.int64)
    for s in range(num_slots):
        base[s] = len(values)
        for b, pl in enumerate(per_line):
            vals = pl[s]
            if hash_rows:
                vals = [R.feasign_row(f, hash_rows) for f in vals]
            values += vals
            lod[s, b + 1] = lod[s, b] + len(vals)
    base[num_slots] = len(values)
    as_i64 = np.asarray([v - (1 << 64) if v >= (1 << 63) else v for v in values], dtype=np.int64)  # bit pattern
    return as_i64, lod, base


@pytest.mark.parametrize("threads,hash_rows", [(1, 0), (5, 0), (3, 1000003)])
def test_feasign_slots_reference_demo(reader, threads, hash_rows):
    """First 4 lines of the reference's models/rank/slot_dnn/data/demo_10 (uint64 feasigns > 2^63, multi-value
    slots, 271 distinct slots) against the restatement of queuedataset_reader.py line_process."""
    data = open(os.path.join(GOLDEN, "slot_dnn_demo_4.txt"), "rb").read()
    lines = data.decode().strip().split("\n")
    values, lod, base, n = reader.parse_feasign_slots(data, 1, 301, hash_rows, threads)
    ov, ol, ob = _oracle_csr(lines, 1, 301, hash_rows)
    assert n == 4 and np.array_equal(base.numpy(), ob) and np.array_equal(lod.numpy(), ol)
    assert np.array_equal(values.numpy(), ov)
    if not hash_rows:
        assert (values.numpy() < 0).any()                       # feasigns above 2^63 kept as int64 bit patterns
    else:
        assert values.min() >= 0 and values.max() < hash_rows
    # slot "1" is the click label (first token of every line is "<0|1>:1")
    assert [int(values[base[0] + lod[0, b]]) for b in range(4)] == [int(ln.split(":")[0]) for ln in lines]
    # every (line, slot) holds at least one id: absent slots are padded with a single 0
    assert int((lod[:, 1:] - lod[:, :-1]).min()) == 1


def test_feasign_slots_edge_cases(reader):
    import ctypes as C
    from paddlerec_amd._lib import lib
    data = b"7:2 9:2 5:9  18446744073709551615:3 x:3 4: :5 11:77\n\n3:3\r\n8:2"
    values, lod, base, n = reader.parse_feasign_slots(data, 2, 3, 0, 2)          # slots 2, 3, 4
    assert n == 4
    v = values.numpy().tolist()
    seg = lambda s, b: v[base[s] + lod[s, b]: base[s] + lod[s, b + 1]]
    assert seg(0, 0) == [7, 9] and seg(1, 0) == [-1] and seg(2, 0) == [0]        # 2^64-1 -> bit pattern -1; pad
    assert seg(0, 1) == [0] and seg(1, 1) == [0] and seg(2, 1) == [0]            # empty line: every slot padded
    assert seg(1, 2) == [3] and seg(0, 3) == [8]
    assert int(base[-1]) == len(v) == 13
    nl, nv = C.c_int64(0), C.c_int64(0)
    rc = lib().rec_parse_feasign_slots(data, len(data), 2, 3, 0, 4, 5, 1, None, lod.data_ptr(), base.data_ptr(),
                                       C.byref(nl), C.byref(nv))
    assert rc == -3 and nv.value == 13                                            # REC_EWORKSPACE reports the size
    assert lib().rec_parse_feasign_slots(data, len(data), 2, 3, 1, 4, 99, 1, None, None, None, C.byref(nl),
                                         C.byref(nv)) == -1                       # hash_rows = 1 is meaningless


def test_feasign_token_cache_equals_second_tokenize(reader, monkeypatch):
    """rec_parse_feasign_slots keeps pass 1's (slot, feasign) pairs for pass 2 unless the text is huge; with the cache
    switched off (the > 1 GiB path) it tokenizes twice.  Same values / lod / bases, any thread count, hashed or raw."""
    rng = np.random.default_rng(12)
    lines = []
    for i in range(700):
        toks = ["%d:1" % (i & 1)]
        for s in rng.choice(np.arange(2, 40), size=int(rng.integers(0, 20)), replace=False):
            toks += ["%d:%d" % (int(rng.integers(0, 2 ** 63)), s) for _ in range(int(rng.integers(1, 5)))]
        if i % 50 == 7:
            toks += ["junk", "12:", ":3", "5:x", "9:2:1", ""]
        rng.shuffle(toks)
        lines.append(" ".join(toks))
    lines[100] = ""
    data = ("\n".join(lines)).encode()
    for threads in (1, 3):
        for hash_rows in (0, 1000003):
            monkeypatch.delenv("REC_FEASIGN_CACHE", raising=False)
            a = reader.parse_feasign_slots(data, 1, 39, hash_rows, threads)
            monkeypatch.setenv("REC_FEASIGN_CACHE", "0")
            b = reader.parse_feasign_slots(data, 1, 39, hash_rows, threads)
            assert a[3] == b[3] == len(lines)
            for x, y in zip(a[:3], b[:3]):
                assert np.array_equal(x.numpy(), y.numpy())


def test_feasign_slot_reader_batches(reader, tmp_path):
    data = open(os.path.join(GOLDEN, "slot_dnn_demo_4.txt"), "rb").read()
    p = tmp_path / "part-0"
    p.write_bytes(data + data[: data.index(b"\n") + 1])                           # 5 lines -> 2 batches of 2
    got = list(reader.FeasignSlotReader([str(p)], 2, device="cpu", hash_rows=1000003))
    assert len(got) == 2
    lines = data.decode().strip().split("\n")
    for i, (values, lod, base) in enumerate(got):
        ov, ol, ob = _oracle_csr(lines[2 * i: 2 * i + 2], 1, 301, 1000003)
        assert np.array_equal(values.numpy(), ov) and np.array_equal(lod.numpy(), ol) and np.array_equal(base.numpy(), ob)
