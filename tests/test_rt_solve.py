"""The reflection / transmission event: the engine's slowness-form solve beside the oracle's RTCoef (round 6).

`rt_choose` / `rt_apply` (csrc/r3d_physics.h) work from vertical slownesses sqrt(1 / v^2 - p^2) and one set of
formulas in (own type, other type) velocities; the oracle (`oracle/r3d_oracle.cpp` rt_coefs_psv / rt_coefs_sh /
rt_event_core) restates the reference's RTCoef with std::complex cosines, divisions by the determinant and the
six-entry chooser (rtcoef.cpp:107-198, :207-278, :289-393, :406-588).  Same interface, same phonon, same two uniforms
for both, case by case (tests/emul/emul.cpp `r3d_emul_rt_events` makes the cases):

* the outcome (ray type and side) is the oracle's wherever the outcome draw is further than 1e-9 of the weights' total
  from every partial sum (the two formulations round differently in the last digits, so a draw ON a boundary may fall
  either way -- none is observed);
* the outgoing direction and the polarisation agree to 1e-9 -- plus, at grazing and at near-normal incidence, what the
  REFERENCE's formulation loses there (cos i taken as sqrt(1 - sin^2 i): 1e-16 / cos i; its unit axis normal to the plane
  of incidence: 1e-16 / sin i): the engine has cos i = n.d itself and never normalises that axis, and is the better
  conditioned side of the two.  tests/emul/emul.cpp states the allowances where they are applied.
"""
import ctypes as C

import pytest

import emul_ffi as E
from oracle import oracle_ffi as O

MODES = {0: "solid on solid", 1: "free surface", 2: "within 1e-12 .. 1e-2 of a critical angle",
         3: "grazing incidence", 4: "nearly identical media",
         5: "a fluid on one side (the reference's default outcome)", 6: "contrasts up to 1e3",
         7: "near-normal incidence", 8: "along the normal exactly (the substitute axis)",
         9: "horizontal faces through the layered models' flat-face form"}
TOL, MARGIN = 1e-9, 1e-9


def run(mode, n, seed):
    L = E.lib()
    L.r3d_emul_rt_events.restype = None
    L.r3d_emul_rt_events.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_double, C.c_double, C.POINTER(C.c_uint64),
                                     C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p]
    fn = C.cast(O.lib().r3d_oracle_rt_event, C.c_void_p)
    out, dev, first = (C.c_uint64 * 8)(), (C.c_double * 2)(), (C.c_double * 16)()
    L.r3d_emul_rt_events(mode, n, seed, TOL, MARGIN, out, dev, first, fn)
    return dict(cases=out[0], outcome_differs=out[1], outcome_differs_outside_margin=out[2], direction_off=out[3],
                polarisation_off=out[4], transmitted=out[5], s_out=out[6], sh_in=out[7], dev_dir=dev[0], dev_pol=dev[1],
                first=list(first))


@pytest.mark.parametrize("mode,n", [(0, 6_000_000), (1, 3_000_000), (2, 3_000_000), (3, 2_000_000), (4, 2_000_000),
                                    (5, 1_000_000), (6, 3_000_000), (7, 2_000_000), (8, 500_000), (9, 4_000_000)])
def test_event_is_the_oracles_event(mode, n):
    r = run(mode, n, seed=20261005 + mode)
    assert r["cases"] == n
    assert r["outcome_differs_outside_margin"] == 0, (MODES[mode], r)
    assert r["outcome_differs"] <= 2, (MODES[mode], r)          # (expected: n x 1e-15; observed: 0)
    assert r["direction_off"] == 0 and r["polarisation_off"] == 0, (MODES[mode], r)
    # every kind of outcome takes part in the comparison
    if mode == 1:
        assert r["transmitted"] == 0
    elif mode == 5:
        assert r["transmitted"] == 0      # rtcoef.cpp:70-75: indeterminate weights -> reflected, same type
    elif mode != 3:                       # (a grazing ray is reflected, nearly always)
        assert r["transmitted"] > 0.05 * n
    assert r["s_out"] > 0.15 * n
    if mode != 5:
        assert r["sh_in"] > 0.05 * n


def test_weights_are_the_oracles_probabilities_times_the_determinant():
    """`rt_weights` (what `./main --rtcoef-test` prints from) against `r3d_oracle_rt_probs` on the reference's own test
    interface (main.cpp:82-84) and its mirror image, 2000 sines each."""
    import numpy as np
    E.lib().r3d_emul_rt_weights.restype = None
    E.lib().r3d_emul_rt_weights.argtypes = [C.POINTER(C.c_double), C.c_double, C.c_int, C.POINTER(C.c_double)]
    for media in [(10, 8, 4, 8, 4, 2), (8, 4, 2, 10, 8, 4), (3.3, 8.1, 4.5, 2.7, 6.4, 3.6)]:
        for intype in (0, 1, 2):
            for s in np.linspace(0.0, 0.99999, 2000):
                w = (C.c_double * 7)()
                E.lib().r3d_emul_rt_weights((C.c_double * 6)(*media), float(s), intype, w)
                probs = O.rt_probs(*media, float(s), intype)
                ours = np.array(w[:6]) / w[6]
                assert np.allclose(ours, probs, rtol=1e-9, atol=1e-12 * max(probs)), (media, intype, s, ours, probs)


BEND_MODES = {0: "random faces", 1: "velocity steps of 1e-5 .. 1e-3", 2: "within 1e-12 .. 1e-2 of total reflection",
              3: "grazing and near-normal incidence", 4: "horizontal faces through the flat-face form"}


@pytest.mark.parametrize("mode,n", [(0, 4_000_000), (1, 2_000_000), (2, 2_000_000), (3, 2_000_000), (4, 4_000_000)])
def test_bend_is_the_oracles_bend(mode, n):
    """`bend` (csrc/r3d_physics.h: Snell's law on the tangential part of the direction, the particle motion carried in
    un-normalised axes; on a horizontal face of a layered model the polarisation angle simply carries over) against the
    oracle's restatement of Phonon::Refraction_Bend (phonons.cpp:311-405: unit axes, SH / SV components, atan2), face by
    face: crossed or totally reflected alike unless the outgoing sine is within 1e-9 of 1, directions and polarisations to
    1e-9 (plus what the reference's own cos i and unit axis lose at grazing and near-normal incidence)."""
    L = E.lib()
    L.r3d_emul_bend_events.restype = None
    L.r3d_emul_bend_events.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_double, C.c_double, C.POINTER(C.c_uint64),
                                       C.POINTER(C.c_double), C.c_void_p]
    out, dev = (C.c_uint64 * 8)(), (C.c_double * 2)()
    L.r3d_emul_bend_events(mode, n, 20261006 + mode, TOL, MARGIN, out, dev, C.cast(O.lib().r3d_oracle_bend_event, C.c_void_p))
    r = dict(cases=out[0], side_differs=out[1], side_differs_outside_margin=out[2], direction_off=out[3],
             polarisation_off=out[4], crossed=out[5], s_rays=out[6], dev_dir=dev[0], dev_pol=dev[1])
    assert r["cases"] == n
    assert r["side_differs_outside_margin"] == 0 and r["side_differs"] <= 2, (BEND_MODES[mode], r)
    assert r["direction_off"] == 0 and r["polarisation_off"] == 0, (BEND_MODES[mode], r)
    assert r["s_rays"] > 0.4 * n and 0.05 * n < r["crossed"] <= n, (BEND_MODES[mode], r)


@pytest.mark.parametrize("mode,n", [(0, 4_000_000), (1, 1_000_000), (2, 1_000_000)])
def test_scatter_rotation_is_the_oracles_transform(mode, n):
    """`scatter_transform` (csrc/r3d_physics.h: the deflection's unit vector and S1 axis expressed in the phonon's
    (S1, S2, direction) frame by algebra) against the oracle's Phonon::Transform (phonons.cpp:116-170, OrthoAxes
    geom_r3.cpp:212-340: three frames built from angles, acos / atan2 back), case by case: random phonons and
    deflections, deflections all but forward or backward, phonons all but along a pole."""
    L = E.lib()
    L.r3d_emul_transforms.restype = None
    L.r3d_emul_transforms.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_double, C.POINTER(C.c_uint64), C.POINTER(C.c_double),
                                      C.c_void_p]
    out, dev = (C.c_uint64 * 3)(), (C.c_double * 2)()
    L.r3d_emul_transforms(mode, n, 20261007 + mode, TOL, out, dev, C.cast(O.lib().r3d_oracle_transform, C.c_void_p))
    assert out[0] == n and out[1] == 0 and out[2] == 0, dict(cases=out[0], direction_off=out[1], polarisation_off=out[2],
                                                               dev_dir=dev[0], dev_pol=dev[1])
