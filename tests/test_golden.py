"""Committed fixtures under tests/golden/:
  reference_recorded.json   numbers the survey measured on the unmodified reference
  philox4x32_10_kat.json    the generator's published known-answer vectors
  oracle_vectors.npz        regression vectors of this repository's oracle (make_golden.py)
CPU: the oracle (and the host builder) against them.  GPU: the engine against the
regression vectors, without running the oracle."""
import json
import os

import numpy as np
import pytest

from tests.configs import CONFIGS

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF = json.load(open(os.path.join(HERE, "reference_recorded.json")))
VEC = np.load(os.path.join(HERE, "oracle_vectors.npz"))
NAMES = sorted(k[:-7] for k in VEC.files if k.endswith("_finals"))


def same_finals(got, want):
    """Integer fields exact, fp64 fields 1e-9 relative (1e-9 absolute on the unit direction)."""
    bad = 0
    for g, w in zip(got, want):
        ok = (g.fate, g.type, g.moves, g.n_catch) == (w["fate"], w["type"], w["moves"], w["n_catch"])
        for a, b in ((g.time, w["time"]), (g.path, w["path"]), (g.amp, w["amp"])):
            ok &= abs(a - b) <= 1e-9 * max(1.0, abs(b))
        ok &= bool(np.allclose(list(g.loc), w["loc"], rtol=1e-9, atol=1e-7))
        ok &= bool(np.allclose(list(g.dir), w["dir"], rtol=0, atol=1e-9))
        bad += not ok
    return bad


def test_philox_known_answers_from_fixture():
    from oracle import oracle_ffi as O
    kat = json.load(open(os.path.join(HERE, "philox4x32_10_kat.json")))["vectors"]
    assert len(kat) == 3
    for v in kat:
        h = lambda xs: tuple(int(x, 16) for x in xs)
        assert tuple(O.philox(h(v["counter"]), h(v["key"]))) == h(v["output"])


@pytest.mark.parametrize("name", ["halfspace", "crustpinch", "lopnor", "sphere"])
def test_reference_recorded_sizes_and_event_mix(models, name):
    from oracle import oracle_ffi as O
    m = models(name, 5)
    want = REF["model_sizes"][name]
    assert (m.n_cells, m.n_scatterers, m.n_seismometers, m.n_bins) == (
        want["cells"], want["scatterers"], want["seismometers"], want["bins"])
    n = {"halfspace": 30000, "crustpinch": 15000, "lopnor": 15000, "sphere": 1200}[name]
    res = O.run(m, n, first_id=7 << 32)
    for k, v in REF["events_per_history"][name].items():
        if k.startswith("_"):
            continue
        assert res.events[k] / n == pytest.approx(v, rel=0.12 if v > 1 else 0.22), (k, res.events[k] / n, v)


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_its_regression_vectors(name):
    from oracle import oracle_ffi as O
    from radiative3d_amd import Model
    want = VEC[name + "_finals"]
    m = Model(CONFIGS[name](3))
    res, fin = O.run(m, len(want), trace=True)
    assert same_finals(fin, want) == 0
    assert np.array_equal(res.scalars(), VEC[name + "_scalars"])
    assert np.array_equal(res.counts.sum(axis=1), VEC[name + "_counts"])
    assert np.allclose(res.energy.sum(axis=1), VEC[name + "_energy"], rtol=1e-12, atol=0)


def test_rt_table_regression_and_flux_conservation():
    from oracle import oracle_ffi as O
    t = VEC["rt_table"]
    assert t.shape == (3, 100, 7)
    for row, intype in ((0, 0), (1, 2), (2, 1)):       # stored P, SV, SH; oracle codes 0 P, 1 SH, 2 SV
        for i in (0, 17, 50, 99):
            assert np.allclose(O.rt_probs(10, 8, 4, 8, 4, 2, t[row, i, 0], intype), t[row, i, 1:], rtol=1e-12, atol=1e-15)
    # each row's six outcome weights add up to the incident energy flux rho1 * v_in * cos(i)
    # (rtcoef.cpp:150-186): for P incidence rho1 alpha1 cos i = 80 cos i
    s = t[0, :, 0]
    assert np.allclose(t[0, :, 1:].sum(1), 80.0 * np.sqrt(1 - s * s), rtol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_engine_reproduces_the_regression_vectors(name):
    from radiative3d_amd import Engine, Model
    want = VEC[name + "_finals"]
    m = Model(CONFIGS[name](3))
    res, fin = Engine(m).run(len(want), trace=True)
    assert same_finals(fin, want) == 0
    assert np.array_equal(res.scalars(), VEC[name + "_scalars"])
    assert np.array_equal(res.counts.sum(axis=1), VEC[name + "_counts"])
    # (an axis component that is a tiny fraction of its catch's energy is ill-conditioned
    #  relative to itself: measure against the largest total)
    want_e = VEC[name + "_energy"]
    assert np.allclose(res.energy.sum(axis=1), want_e, rtol=1e-9, atol=1e-12 * want_e.max())
