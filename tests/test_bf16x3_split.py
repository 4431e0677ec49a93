"""The arithmetic of csrc/gemm_bf16x3.h restated in NumPy (no GPU): a f32 operand as three bf16 terms, a product as the six
term products with i + j <= 2.  Pins what the GPU tests (tests/test_gemm_gpu.py::test_gemm_bf16x3*) measure on hardware:
the split reproduces x to 2^-24 |x| or better whatever the exponent, and a K-term dot product from the six products, summed
in float64, is within 2^-22 of sum |a||b| of the exact one — below the rounding of ONE f32 accumulation step, so the f32
accumulator of the MFMA, not the split, sets the kernel's error."""
import numpy as np


def bf16_rne(x):
    """float32 -> nearest bfloat16 (ties to even), returned as float32 (what v_cvt_pk_bf16_f32 does)."""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return r.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, np.float32)
    x0 = bf16_rne(x)
    r1 = (x - x0).astype(np.float32)            # exact in f32: x3_split_pair's v_sub / v_pk_add
    x1 = bf16_rne(r1)
    r2 = (r1 - x1).astype(np.float32)
    x2 = bf16_rne(r2)
    return x0, x1, x2, r1, r2


def test_three_terms_reproduce_the_operand():
    rng = np.random.default_rng(5)
    x = rng.standard_normal(200000) * np.exp(20 * rng.standard_normal(200000))
    x = x[(np.abs(x) > 1e-30) & (np.abs(x) < 1e30)].astype(np.float32)     # (denormal remainders / overflow: not this path's data)
    x = np.concatenate([x, np.float32([0.0, -0.0, 1.0, -1.0, 3.0e38, 1.2e-30, 1 + 2 ** -23, 1 - 2 ** -24])])
    x0, x1, x2, r1, r2 = split3(x)
    xd = x.astype(np.float64)
    # the remainders are exact (the kernel relies on it)
    assert np.array_equal(r1.astype(np.float64), xd - x0.astype(np.float64))
    assert np.array_equal(r2.astype(np.float64), xd - x0.astype(np.float64) - x1.astype(np.float64))
    err = np.abs(xd - (x0.astype(np.float64) + x1 + x2))
    assert np.all(err <= 2.0 ** -24 * np.abs(xd)), float((err / np.maximum(np.abs(xd), 1e-300)).max())
    # most operands are reproduced exactly (8 + 8 + 8 significand bits, signed remainders)
    assert np.mean(err == 0) > 0.99
    # every term is a bf16 value (low 16 bits clear)
    for t in (x0, x1, x2):
        assert np.all(t.view(np.uint32) & 0xFFFF == 0)
    assert np.all(np.abs(r1) <= 2.0 ** -8 * np.abs(x)) and np.all(np.abs(r2) <= 2.0 ** -16 * np.abs(x) * 1.01)


def test_six_products_are_f32_grade():
    rng = np.random.default_rng(6)
    K, M, N = 400, 64, 48
    A = rng.uniform(-1, 1, (M, K)).astype(np.float32)
    B = rng.uniform(-1, 1, (K, N)).astype(np.float32)
    a = [t.astype(np.float64) for t in split3(A)[:3]]
    b = [t.astype(np.float64) for t in split3(B)[:3]]
    got = sum(a[i] @ b[j] for i in range(3) for j in range(3) if i + j <= 2)      # exact accumulation of the kept products
    want = A.astype(np.float64) @ B.astype(np.float64)
    mag = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
    assert (np.abs(got - want) / mag).max() < 2.0 ** -22          # the dropped products a1 b2 + a2 b1 + a2 b2
    # three products (a0 b0 + a0 b1 + a1 b0) are not: ~1e-6 of sum |a||b| from the split alone, several times the f32
    # accumulation error of the exact kernels (3.7e-7 at K 400) — why six are kept
    got3 = a[0] @ b[0] + a[0] @ b[1] + a[1] @ b[0]
    assert (np.abs(got3 - want) / mag).max() > 2.0 ** -21
    # plain bf16 (one product) is 2^-9-grade
    assert (np.abs(a[0] @ b[0] - want) / mag).max() > 1e-4
