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


def oracle_batches(model, n_batches, per_batch, first_id, threads=8):
    """`n_batches` independent oracle runs of `per_batch` histories each (ctypes releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle_ffi as O
    with ThreadPoolExecutor(threads) as pool:
        return list(pool.map(lambda b: O.run(model, per_batch, first_id=first_id + b * per_batch), range(n_batches)))


@pytest.mark.parametrize("name", ["halfspace", "crustpinch", "lopnor", "sphere"])
def test_reference_recorded_sizes_and_event_mix(models, name):
    """The survey's per-history event mix of the unmodified reference, held to its own Monte-Carlo
    error: 3 sigma of a 5000-history mean (from our batch means at that size) + the rounding of the
    printed figure (tests/refstats.py)."""
    from refstats import tolerance
    m = models(name, 5)
    want = REF["model_sizes"][name]
    assert (m.n_cells, m.n_scatterers, m.n_seismometers, m.n_bins) == (
        want["cells"], want["scatterers"], want["seismometers"], want["bins"])
    n_ref = REF["events_per_history"]["_n_histories"]
    batches = oracle_batches(m, 8 if name == "sphere" else 24, n_ref, first_id=7 << 32)
    for k, v in REF["events_per_history"][name].items():
        if k.startswith("_"):
            continue
        means = [b.events[k] / n_ref for b in batches]
        ours = sum(means) / len(means)
        tol = tolerance(means, v)
        print(f"{name:10s} {k:10s} reference {v:8.3f}  ours {ours:8.3f}  allowed +-{tol:.3f} ({100 * tol / v:.1f} %)")
        assert abs(ours - v) <= tol, (name, k, ours, v, tol)


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
