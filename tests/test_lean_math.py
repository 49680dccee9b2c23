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


# ---- the same functions on the device (wave-level tiers: 64 consecutive elements share a vote) ----
def _device(which, x, y=None):
    import ctypes as C
    from radiative3d_amd import _ffi
    L = _ffi.hip_lib()
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    dp = C.POINTER(C.c_double)
    yp = None
    if y is not None:
        y = np.ascontiguousarray(y, dtype=np.float64)
        yp = y.ctypes.data_as(dp)
    rc = L.r3d_selftest_math(0, which, x.ctypes.data_as(dp), yp, out.ctypes.data_as(dp), x.size)
    assert rc == 0, L.r3d_last_error().decode()
    return out


def _waves(small, large, n_waves=24):
    """Waves of 64: all-small, all-large, and mixed (one large lane among small ones, half and half)."""
    rows = []
    for w in range(n_waves):
        kind = w % 4
        s, l = RNG.choice(small, 64), RNG.choice(large, 64)
        if kind == 0:
            rows.append(s)
        elif kind == 1:
            rows.append(l)
        elif kind == 2:
            r = s.copy(); r[RNG.integers(64)] = l[0]; rows.append(r)
        else:
            r = s.copy(); r[::2] = l[::2]; rows.append(r)
    return np.concatenate(rows)


@pytest.mark.gpu
def test_device_atanh_asin_rotation_across_wave_tiers():
    sgn = lambda n: RNG.choice([-1.0, 1.0], n)
    # atanh: tiers at 1/16, 1/4 (halving), 1/2 (logarithm)
    small, large = np.logspace(-12, np.log10(1 / 16), 500), np.concatenate([RNG.uniform(1 / 16, 0.5, 400), RNG.uniform(0.5, 0.97, 100)])
    x = _waves(small, large); x *= sgn(x.size)
    got = _device(ATANH, x)
    assert np.max(np.abs(got - np.arctanh(x)) / np.abs(np.arctanh(x))) < 8e-16
    # asin_small: |x| <= 1/16 short series by vote, else the rational form
    x = _waves(np.logspace(-12, np.log10(1 / 16), 500), RNG.uniform(1 / 16, 0.5, 500)); x *= sgn(x.size)
    assert np.max(np.abs(_device(ASIN, x) - np.arcsin(x)) / np.abs(np.arcsin(x))) < 6e-16
    # rotation: pi/4, pi/2, pi tiers, and beyond
    x = _waves(RNG.uniform(0, math.pi / 4, 500), np.concatenate([RNG.uniform(math.pi / 4, math.pi, 450), RNG.uniform(math.pi, 30, 50)]))
    x *= sgn(x.size)
    assert np.max(np.abs(_device(ROT_S, x) - np.sin(x))) < 2e-15
    assert np.max(np.abs(_device(ROT_C, x) - np.cos(x))) < 2e-15


@pytest.mark.gpu
def test_device_angle_exp_log_and_the_newton_reciprocals():
    a = _waves(RNG.uniform(-0.5, 0.5, 500), RNG.uniform(-math.pi, math.pi, 500))
    got = _device(ANGLE, np.sin(a), np.cos(a))
    want = np.arctan2(np.sin(a), np.cos(a))
    assert np.max(np.abs(got - want)) < 1.2e-15
    x = np.concatenate([RNG.uniform(-30, 5, 1000), -np.logspace(-12, 1, 280)])
    assert np.max(np.abs(_device(EXP, x) / np.exp(x) - 1)) < 5e-16
    x = np.concatenate([RNG.uniform(1e-9, 1, 1000), np.logspace(-300, 3, 280)])
    assert np.max(np.abs(_device(LOG, x) - np.log(x)) / np.maximum(1.0, np.abs(np.log(x)))) < 5e-16
    x = np.concatenate([RNG.uniform(1e-3, 1e4, 1000), np.logspace(-30, 30, 280)]) * RNG.choice([-1.0, 1.0], 1280)
    assert np.max(np.abs(_device(7, x) * x - 1)) < 5e-16                 # frcp
    x = np.abs(x)
    assert np.max(np.abs(_device(8, x) * np.sqrt(x) - 1)) < 5e-16        # frsqrt
    assert np.max(np.abs(_device(9, x) / np.sqrt(x) - 1)) < 5e-16        # fsqrt
    assert _device(9, np.zeros(64))[0] == 0.0
