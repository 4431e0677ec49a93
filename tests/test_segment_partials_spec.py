"""Executable statement of the hot-row partial-sum layout (include/recengine.h: rec_segment_partials, REC_SEG_TILE,
REC_SEG_LONG) that the producer kernel and every consumer (`segment_sum` in csrc/sparse_update.hip) follow:
  producer, per tile t of 64 sorted positions [s, e): slot 0 = piece of the LONG segment holding position s,
            slot 1 = piece of a different LONG segment holding position e-1;
  consumer, per long segment [beg, end): tiles beg//64 .. (end-1)//64, slot = 0 if beg <= 64*t else 1.
The GPU test (tests/test_deepfm_gpu.py::test_hot_rows_segment_partials) checks the kernels; this checks that the
two index rules cover every long segment exactly once for arbitrary segmentations (no GPU needed)."""
import re

import numpy as np
import pytest

from conftest import REPO

T, LONG = 64, 128


def test_constants_match_the_header():
    import os
    h = open(os.path.join(REPO, "include", "recengine.h")).read()
    assert int(re.search(r"#define REC_SEG_TILE (\d+)", h).group(1)) == T
    assert int(re.search(r"#define REC_SEG_LONG (\d+)", h).group(1)) == LONG
    assert LONG >= 2 * T        # a long segment spans >= 2 tiles, so no tile holds a third long piece


def _produce(seg, g):
    nvalid = seg[-1]
    part = np.full(((nvalid + T - 1) // T, 2), np.nan)
    find = lambda x: int(np.searchsorted(seg, x, side="right") - 1)
    for t in range(part.shape[0]):
        s, e = t * T, min(t * T + T, nvalid)
        u0 = find(s)
        b0, e0 = seg[u0], seg[u0 + 1]
        if e0 - b0 >= LONG:
            part[t, 0] = g[s:min(e, e0)].sum()
        if e0 < e:
            u1 = find(e - 1)
            if seg[u1 + 1] - seg[u1] >= LONG:
                part[t, 1] = g[seg[u1]:e].sum()
    return part


def _consume(part, beg, end):
    tot = 0.0
    for t in range(beg // T, (end - 1) // T + 1):
        v = part[t, 0 if beg <= t * T else 1]
        assert not np.isnan(v), "consumer reads a slot the producer did not write"
        tot += v
    return tot


@pytest.mark.parametrize("seed", range(4))
def test_every_long_segment_is_covered_exactly_once(seed):
    rng = np.random.default_rng(seed)
    for _ in range(400):
        lens = rng.choice([1, 2, 5, 63, 64, 65, 127, 128, 129, 191, 192, 200, 256, 1000], size=rng.integers(1, 30))
        seg = np.concatenate([[0], np.cumsum(lens)]).astype(int)
        g = rng.standard_normal(seg[-1])
        part = _produce(seg, g)
        for u in range(len(lens)):
            if lens[u] >= LONG:
                assert abs(_consume(part, seg[u], seg[u + 1]) - g[seg[u]:seg[u + 1]].sum()) < 1e-9
