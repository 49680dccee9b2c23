"""The traversal kernel's own elementary functions (radiative3d_amd/csrc/r3d_math.h) against
numpy, over the argument ranges the kernel feeds them and across every tier boundary of the
tiered routines.  Host build of the same source (the device build differs only in using the
hardware reciprocal / reciprocal-root seeds with Newton steps)."""
import math

import numpy as np
import pytest

from . import emul_ffi

EXP, LOG, ATANH, ASIN, ANGLE, ROT_S, ROT_C = range(7)
RNG = np.random.default_rng(20261004)


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-300)


def test_exp_and_log():
    for x in np.concatenate([RNG.uniform(-30, 5, 300), -np.logspace(-12, 1, 60), [0.0, -0.0]]):
        assert rel(emul_ffi.math_fn(EXP, x), math.exp(x)) < 4e-16
    for x in np.concatenate([RNG.uniform(1e-9, 1, 300), np.logspace(-300, 3, 80), 1 + np.logspace(-15, -1, 30)]):
        assert abs(emul_ffi.math_fn(LOG, x) - math.log(x)) <= 4e-16 * max(1.0, abs(math.log(x)))


def test_atanh_every_tier():
    edges = [1 / 16, 0.25, 0.268, 0.5]
    xs = np.concatenate([np.logspace(-14, math.log10(0.98), 400), RNG.uniform(0, 0.99, 400)] +
                        [[e * (1 - 1e-12), e, e * (1 + 1e-12)] for e in edges])
    for x in xs:
        for sgn in (1.0, -1.0):
            got, want = emul_ffi.math_fn(ATANH, sgn * x), math.atanh(sgn * x)
            assert rel(got, want) < 6e-16, (x, got, want)
    assert emul_ffi.math_fn(ATANH, 0.0) == 0.0
    assert math.isnan(emul_ffi.math_fn(ATANH, float("nan")))


def test_asin_small_range():
    for x in np.concatenate([np.logspace(-14, math.log10(0.5), 200), RNG.uniform(-0.5, 0.5, 300), [1 / 16, 0.5, -0.5]]):
        assert rel(emul_ffi.math_fn(ASIN, x), math.asin(x)) < 4e-16 or x == 0


def test_angle_from_sine_and_cosine_full_circle():
    sector_edges = np.deg2rad([0, 30, 60, 90, 120, 150, 180])
    angles = np.concatenate([RNG.uniform(-math.pi, math.pi, 2000), np.logspace(-13, 0, 100), -np.logspace(-13, 0, 100)] +
                            [[e - 1e-9, e, e + 1e-9, -e - 1e-9, -e, -e + 1e-9] for e in sector_edges])
    for a in angles:
        if abs(a) > math.pi:
            continue
        got = emul_ffi.math_fn(ANGLE, math.sin(a), math.cos(a))
        want = math.atan2(math.sin(a), math.cos(a))
        assert abs(got - want) < 5e-16 * max(1.0, abs(want)) + 2e-16, (a, got, want)
    # the conventions of atan2 at the ends
    assert emul_ffi.math_fn(ANGLE, 0.0, 1.0) == 0.0
    assert emul_ffi.math_fn(ANGLE, 0.0, -1.0) == pytest.approx(math.pi, abs=1e-15)
    assert emul_ffi.math_fn(ANGLE, -0.0, -1.0) == pytest.approx(-math.pi, abs=1e-15)
    assert math.isnan(emul_ffi.math_fn(ANGLE, float("nan"), 0.5))


def test_rotation_up_to_pi_and_beyond():
    xs = np.concatenate([RNG.uniform(-math.pi, math.pi, 1500), np.logspace(-12, 0.49, 100), [math.pi / 4, math.pi / 2, math.pi],
                         RNG.uniform(-40, 40, 100)])
    for x in xs:
        # (each doubling of the half / quarter angle roughly doubles the absolute error)
        tol = 5e-16 if abs(x) <= math.pi / 4 or abs(x) > math.pi else 2e-15
        assert abs(emul_ffi.math_fn(ROT_S, x) - math.sin(x)) < tol, x
        assert abs(emul_ffi.math_fn(ROT_C, x) - math.cos(x)) < tol, x
