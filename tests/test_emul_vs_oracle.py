"""The engine's per-work-item code (radiative3d_amd/csrc/r3d_step.h, the exact
functions the HIP kernel runs per lane), compiled for the host by
tests/emul, against the oracle: same Philox draws, so the comparison is
history by history.  This is the CPU-side rehearsal of the GPU parity tests."""
import numpy as np
import pytest

import emul_ffi as E
from conftest import finals_differ
from oracle import oracle_ffi as O
from radiative3d_amd import Model
from tests.configs import halfspace


def compare(model, n, first_id=0, seed=0x5EED, allow=0):
    ro, fo = O.run(model, n, first_id, seed, trace=True)
    re, fe = E.run(model, n, first_id, seed, trace=True)
    bad = sum(finals_differ(a, b) for a, b in zip(fe, fo))
    assert bad <= allow, f"{bad} of {n} histories differ"
    assert (ro.n_lost, ro.n_timeout, ro.n_invalid) == (re.n_lost, re.n_timeout, re.n_invalid)
    if bad == 0:
        assert ro.events == re.events
        assert (ro.counts == re.counts).all()
        assert np.allclose(ro.energy, re.energy, rtol=1e-9, atol=1e-13)
    return ro, re


@pytest.mark.parametrize("name,n", [("halfspace", 20000), ("crustpinch", 4000), ("lopnor", 4000),
                                    ("sphere", 400), ("toysphere_vids", 500), ("lopnor_vids", 300), ("upthrust", 4000)])
def test_kernel_code_matches_oracle_history_by_history(models, name, n):
    compare(models(name), n)


def test_other_seeds_and_id_offsets(models):
    compare(models("crustpinch"), 1000, first_id=2**33 + 17, seed=0xABCDEF0123)
    compare(models("halfspace"), 3000, first_id=12345, seed=1)


def test_no_deflect_with_mfp_override(models):
    """--overridemfp + --nodeflect (video runs, do-crustpinch-vids.sh:48-50): one dummy
    scatterer for all cells, undeflected 'scatter' check-points (scatterers.cpp:48-52, :325-329)."""
    m = models("crustpinch", 4, ["--overridemfp=25,50", "--nodeflect", "--timetolive=350"])
    assert m.n_scatterers == 1
    ro, _ = compare(m, 1500)
    assert ro.events["scatter"] / 1500 > 3


def test_single_receiver_and_no_receiver():
    """BASELINE config 1 uses one receiver; a model without receivers must still run."""
    m1 = Model(halfspace(4, one_receiver=True))
    assert m1.n_seismometers == 1
    compare(m1, 5000)
    m0 = Model([a for a in halfspace(4) if not a.startswith("--seis")])
    assert m0.n_seismometers == 0
    ro, re = compare(m0, 2000)
    assert ro.events["catch"] == 0 and ro.events["collect"] > 0


def test_empty_and_ragged_batches(models):
    m = models("halfspace")
    ro, re = compare(m, 0)
    assert ro.events["generated"] == 0
    for n in (1, 63, 65, 257):
        compare(m, n, first_id=1000)


def test_strong_contrast_interface_and_liquid_layer(models):
    """Layer contrast -> full R/T at the interface (two attribute sets on the node,
    user_Halfspace_inc.cpp:147); S waves entering the sphere model's liquid outer core
    (Vs = 1e-5) stall and time out."""
    args = [a.replace("6.40,3.63,2.83,-60,6.40,3.63,2.83,-400", "5.0,2.9,2.5,-20,8.0,4.5,3.3,-400")
            for a in halfspace(4)]
    m = Model(args)
    assert m.desc.cells[0].faces[1].flags & 8
    ro, _ = compare(m, 5000)
    assert ro.events["rtsolve"] > ro.events["collect"]
    compare(models("sphere", 4, ["--source-loc=0,0,-3000"]), 150)


def test_deep_source_in_the_spherical_earth():
    """BASELINE config 4: do-spherical.sh with the source 600 km deep."""
    from radiative3d_amd import Model
    from tests.configs import sphere
    ro, re = compare(Model(sphere(4, source_depth=-600)), 300)
    assert ro.n_timeout == 300
