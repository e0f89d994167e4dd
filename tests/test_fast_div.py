"""The exact magic-number division of k_fast_rows' work-unit decode (hyslam_amd/csrc/kernels_fast.hip: fast_div, fast_sched): floor(a / d) as one high
multiply by m = ceil(2^32 / d) and one correction.  The same integer arithmetic restated in Python and checked against // over the ranges the
kernel uses (work units < 2^31; divisors = items per queue / images per queue / items per image) and far beyond them."""
import numpy as np


def fast_div(a, d, m):
    q = (a * m) >> 32                 # __umulhi((uint32_t)a, m)
    q = np.where(d == 1, a, q)        # ceil(2^32 / 1) does not fit 32 bits: m is unused for d == 1
    q = np.where(q * d > a, q - 1, q)
    return q


def magic(d):
    return np.where(d > 1, ((1 << 32) + d - 1) // np.maximum(d, 1), 0)


def test_magic_division_is_exact():
    rng = np.random.default_rng(5)
    d = np.concatenate([np.arange(1, 5000), rng.integers(1, 1 << 20, 20000), rng.integers(1, (1 << 31) - 1, 5000)]).astype(np.int64)
    m = magic(d)
    assert (m < (1 << 32)).all()
    for a in (np.zeros_like(d), d - 1, d, d + 1, 2 * d - 1, rng.integers(0, (1 << 31) - 1, len(d)), np.full_like(d, (1 << 31) - 1),
              (rng.integers(0, 1 << 12, len(d)) * d + rng.integers(0, 1 << 12, len(d)) % d)):
        a = np.clip(a.astype(np.int64), 0, (1 << 31) - 1)
        assert np.array_equal(fast_div(a, d, m), a // d)
