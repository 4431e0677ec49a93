"""Row H: xxh32 feature hash — oracle vs published known-answer vectors, vs the `xxhash` wheel the
reference imports (models/rank/dnn/benchmark_reader.py:52), vs the engine's host function."""
import ctypes as C

import numpy as np
import pytest

from oracle import hash_ref

# Official XXH32 known answers (xxHash repository sanity vectors, seed 0 / PRIME 0x9E3779B1)
KAT = [
    (b"", 0, 0x02CC5D05),
    (b"", 0x9E3779B1, 0x36B78AE7),
    (b"a", 0, 0x550D7456),
    (b"abc", 0, 0x32D153FF),
    (b"Nobody inspects the spammish repetition", 0, 0xE2293B2F),
]


@pytest.mark.parametrize("data,seed,want", KAT)
def test_oracle_known_answers(data, seed, want):
    assert hash_ref.xxh32(data, seed) == want


def _random_strings(n, seed=7):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        ln = int(rng.integers(0, 70))
        out.append(bytes(rng.integers(0, 256, ln, dtype=np.uint8)))
    return out


def test_oracle_matches_xxhash_wheel():
    xxhash = pytest.importorskip("xxhash")
    for s in _random_strings(500):
        assert hash_ref.xxh32(s) == xxhash.xxh32(s).intdigest()
    # the reader's exact call form: str(idx) + feature, default seed
    for idx, feat in [(14, "68fd1e64"), (39, ""), (20, "a" * 33)]:
        assert hash_ref.hash_feature(idx, feat) == xxhash.xxh32(str(idx) + feat).intdigest() % 1000001


def test_c_oracle_and_engine_host_function(oracle_lib, engine_lib):
    oracle_lib.oracle_xxh32.restype = C.c_uint32
    oracle_lib.oracle_xxh32.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32]
    for data, seed, want in KAT:
        assert oracle_lib.oracle_xxh32(data, len(data), seed) == want
        assert engine_lib.rec_xxh32(data, len(data), seed) == want
    for s in _random_strings(300, 11):
        want = hash_ref.xxh32(s)
        assert oracle_lib.oracle_xxh32(s, len(s), 0) == want
        assert engine_lib.rec_xxh32(s, len(s), 0) == want


def test_engine_hash_mod_batch(engine_lib):
    from paddlerec_amd import ops
    fields = list(range(14, 40)) * 3
    vals = ["%08x" % (i * 2654435761 % (1 << 32)) for i in range(len(fields))]
    vals[5] = ""
    got = ops.hash_features(fields, vals)
    want = [hash_ref.hash_feature(f, v) for f, v in zip(fields, vals)]
    assert got == want
    assert all(0 <= g <= 1000000 for g in got)
