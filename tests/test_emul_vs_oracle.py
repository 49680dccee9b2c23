"""The engine's per-work-item code (radiative3d_amd/csrc/r3d_step.h, the exact
functions the HIP kernel runs per lane), compiled for the host by
tests/emul, against the oracle: same Philox draws, so the comparison is
history by history.  This is the CPU-side rehearsal of the GPU parity tests."""
import numpy as np
import pytest

import emul_ffi as E
from oracle import oracle_ffi as O
from oracle.check import assert_aggregates_equal, assert_aggregates_equal_without, forked_ids
from radiative3d_amd import Model
from tests.configs import halfspace


def compare(model, n, first_id=0, seed=0x5EED, allow=0):
    ro, fo = O.run(model, n, first_id, seed, trace=True)
    re, fe = E.run(model, n, first_id, seed, trace=True)
    forked = forked_ids(fe, fo, first_id)
    assert len(forked) <= allow, f"{len(forked)} of {n} histories differ: ids {forked[:20]}"
    # (the aggregates on every path: forked histories are taken out of both sides, oracle/check.py)
    assert_aggregates_equal_without(re, ro, forked, lambda k, i: E.run(model, k, i, seed),
                                    lambda k, i: O.run(model, k, i, seed))
    return ro, re


def test_the_checker_still_compares_aggregates_when_a_history_forks(models):
    """oracle/check.py: a forked history is taken out of both sides and everything else is still held
    bin for bin -- a run with a fork AND a wrong bin elsewhere fails; a run with only the fork passes."""
    import copy
    model, n, seed = models("crustpinch"), 600, 0x5EED
    want, wf = O.run(model, n, 0, seed, trace=True)
    victim = next(i for i, f in enumerate(wf) if f.n_catch > 0)      # a history that was caught somewhere
    stand_in = O.run(model, 1, 10**6, seed)                           # what the "engine" made of it instead
    own = O.run(model, 1, victim, seed)
    got = copy.deepcopy(want)
    got.energy += stand_in.energy - own.energy
    got.counts += stand_in.counts
    got.counts -= own.counts
    for k in got.events:
        got.events[k] += stand_in.events[k] - own.events[k]
    got.n_lost += stand_in.n_lost - own.n_lost
    got.n_timeout += stand_in.n_timeout - own.n_timeout
    with pytest.raises(AssertionError):
        assert_aggregates_equal(got, want)                            # the fork shows in the totals ...
    run_engine = lambda k, i: stand_in if i == victim else O.run(model, k, i, seed)   # noqa: E731
    run_oracle = lambda k, i: O.run(model, k, i, seed)                                  # noqa: E731
    assert_aggregates_equal_without(got, want, [victim], run_engine, run_oracle)       # ... and is accounted for
    s, b = np.argwhere(want.counts.sum(-1) > 0)[-1]
    got.counts[s, b, 0] += 1                                           # a wrong bin that no fork explains
    with pytest.raises(AssertionError):
        assert_aggregates_equal_without(got, want, [victim], run_engine, run_oracle)


@pytest.mark.parametrize("name,n", [("halfspace", 20000), ("crustpinch", 4000), ("lopnor", 4000),
                                    ("sphere", 400), ("toysphere_vids", 500), ("lopnor_vids", 300), ("upthrust", 4000),
                                    ("lopnor_moho", 1500), ("scat_params_study", 4000)])
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
