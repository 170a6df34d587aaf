"""mlst_round_tenths (the rounding the device-side allele choice uses) against Python's own round(): the reference
computes round(float(localScore) / float(nHits), 1) at metamlst.py:149-151 and compares the rounded values at :244."""
import random

import pytest

from metamlst_amd import engine


@pytest.fixture(scope="module")
def lib():
    return engine.load_library()


def _want(p, q):
    return round(float(p) / float(q), 1)


def _check(lib, p, q):
    got = lib.mlst_round_tenths(p, q)
    assert got / 10.0 == _want(p, q) and round(_want(p, q) * 10) == got, (p, q, got, _want(p, q))


def test_rational_ties_every_twentieth(lib):
    # p/q = (2m+1)/20 exactly: the double lands above or below the boundary, or on it (x.25, x.75)
    for q in (20, 40, 60, 100, 4, 8, 12, 2000, 7 * 20):
        for k in range(-2001, 2001, 2):
            p = k * (q // 20) if q % 20 == 0 else None
            if p is not None:
                _check(lib, p, q)
    for q in (4, 8, 12, 16, 28, 1 << 20):
        for k in range(-999, 1000, 2):       # odd quarters: x.25 / x.75 are exact doubles
            _check(lib, k * (q // 4), q)


def test_typical_scores(lib):
    rng = random.Random(7)
    for _ in range(200_000):
        q = rng.randint(1, 5000)
        p = rng.randint(-100 * q, 300 * q)
        _check(lib, p, q)


def test_large_and_small(lib):
    rng = random.Random(11)
    for _ in range(50_000):
        q = rng.randint(1, (1 << 32) - 1)
        p = rng.randint(-(1 << 44), 1 << 44)
        _check(lib, p, q)
    for q in range(1, 400):
        for p in range(-400, 400):
            _check(lib, p, q)
    assert lib.mlst_round_tenths(5, 0) == 0
