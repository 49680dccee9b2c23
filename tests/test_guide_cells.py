"""The search guide of the inverse-CDF draws (radiative3d_amd/csrc/r3d_tables.h GuideCell,
r3d_physics.h sample_cdf_guided) returns the index the reference's bisection returns
(probability.cpp:104-128: smallest k with r <= cdf[k]) -- for short brackets (entries in the cell),
long ones (pivots, then eight at once) and very long ones (pivots, eight at once, bisection), flat
stretches (equal neighbours) and draws at the ends."""
import numpy as np
import pytest

from . import emul_ffi

RNG = np.random.default_rng(7)


def _check(weights, bits, n_draws=20000, extra_u=()):
    cdf = np.cumsum(weights)
    u = np.concatenate([RNG.uniform(0.0, 1.0, n_draws), [np.nextafter(0.0, 1.0), 1.0, 0.5], np.asarray(extra_u, dtype=float)])
    # uniforms sitting exactly on table entries and on guide-cell edges
    on_entries = cdf[RNG.integers(0, cdf.size, 200)] / cdf[-1]
    on_edges = RNG.integers(0, 1 << bits, 200) / float(1 << bits)
    u = np.concatenate([u, on_entries[(on_entries > 0) & (on_entries <= 1)], on_edges[on_edges > 0]])
    a, b, longest = emul_ffi.sample_cdf_both_ways(cdf, u, bits)
    assert np.array_equal(a, b), (np.flatnonzero(a != b)[:5], longest)
    return longest


def test_smooth_table_short_brackets():
    assert _check(RNG.uniform(0.5, 1.5, 4096), bits=10) <= 16


def test_peaked_table_long_brackets():
    x = np.linspace(-1, 1, 50000)
    w = np.exp(-(x / 0.01) ** 2) + 1e-9          # nearly all of the mass in 1 % of the entries
    longest = _check(w, bits=12)
    assert longest > 64                           # pivots + eight at once + bisection all exercised


def test_brackets_of_8_to_64_entries():
    w = np.ones(1 << 15)
    longest = _check(w, bits=10)                  # every bracket 32 entries: pivots + one fetch
    assert 8 <= longest <= 64


def test_flat_stretches_and_zero_weights():
    w = RNG.uniform(0.0, 1.0, 20000)
    w[RNG.integers(0, w.size, 8000)] = 0.0        # runs of equal cumulative values
    w[:50] = 0.0
    _check(w, bits=9)
    _check(w, bits=13)                            # more cells than distinct values


def test_tiny_tables():
    for n in (1, 2, 3, 9, 17):
        _check(RNG.uniform(0.1, 1.0, n), bits=4, n_draws=500)
