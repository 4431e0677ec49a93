"""Row H — feature hash (oracle, test infrastructure only).

Reference: models/rank/dnn/benchmark_reader.py:41-54
    sparse_feature = xxhash.xxh32(str(idx) + features[idx]).intdigest() % hash_dim_
The algorithm lives in the third-party `xxhash` wheel (python-xxhash wrapping Cyan4973/xxHash,
XXH32, seed 0).  It is restated here from the published XXH32 specification and pinned in
tests/test_hash.py against the official known-answer vectors (XXH32("",0)=0x02CC5D05, ...)
and against the `xxhash` module installed in this image.
"""
P1, P2, P3, P4, P5 = 2654435761, 2246822519, 3266489917, 668265263, 374761393
M32 = 0xFFFFFFFF


def _rotl(x, r):
    return ((x << r) | (x >> (32 - r))) & M32


def _round(acc, lane):
    acc = (acc + lane * P2) & M32
    return (_rotl(acc, 13) * P1) & M32


def xxh32(data: bytes, seed: int = 0) -> int:
    n = len(data)
    i = 0
    if n >= 16:
        v1 = (seed + P1 + P2) & M32
        v2 = (seed + P2) & M32
        v3 = seed & M32
        v4 = (seed - P1) & M32
        while i + 16 <= n:
            v1 = _round(v1, int.from_bytes(data[i:i + 4], "little"))
            v2 = _round(v2, int.from_bytes(data[i + 4:i + 8], "little"))
            v3 = _round(v3, int.from_bytes(data[i + 8:i + 12], "little"))
            v4 = _round(v4, int.from_bytes(data[i + 12:i + 16], "little"))
            i += 16
        h = (_rotl(v1, 1) + _rotl(v2, 7) + _rotl(v3, 12) + _rotl(v4, 18)) & M32
    else:
        h = (seed + P5) & M32
    h = (h + n) & M32
    while i + 4 <= n:
        h = (h + int.from_bytes(data[i:i + 4], "little") * P3) & M32
        h = (_rotl(h, 17) * P4) & M32
        i += 4
    while i < n:
        h = (h + data[i] * P5) & M32
        h = (_rotl(h, 11) * P1) & M32
        i += 1
    h ^= h >> 15
    h = (h * P2) & M32
    h ^= h >> 13
    h = (h * P3) & M32
    h ^= h >> 16
    return h


def hash_feature(field_idx: int, value: str, hash_dim: int = 1000001) -> int:
    """benchmark_reader.py:52 — xxh32(str(idx)+feat) % hash_dim."""
    return xxh32((str(field_idx) + value).encode("utf-8")) % hash_dim


# benchmark_reader.py:23-26
CONT_MIN = [0, -3, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]
CONT_DIFF = [20, 603, 100, 50, 64000, 500, 100, 50, 500, 10, 10, 10, 50]


def criteo_tsv_line(line: str, hash_dim: int = 1000001):
    """models/rank/dnn/benchmark_reader.py:39-54 line_process -> (label, ids[26], dense[13] as float32)."""
    import numpy as np
    features = line.rstrip("\n").split("\t")
    dense = []
    for idx in range(1, 14):
        if features[idx] == "":
            dense.append(0.0)
        else:
            dense.append((float(features[idx]) - CONT_MIN[idx - 1]) / CONT_DIFF[idx - 1])
    ids = [hash_feature(idx, features[idx], hash_dim) for idx in range(14, 40)]
    return int(features[0]), np.asarray(ids, np.int64), np.asarray(dense, np.float64).astype(np.float32)
