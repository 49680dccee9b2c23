"""Checker helpers (test infrastructure, like everything under oracle/): comparison of an engine run
with the oracle's run of the same ids when a few histories are allowed to fork.

Device libm and glibc differ in the last ulp, so a history whose branch decision sits on such a bit
may legitimately take another path on the GPU (none has been observed).  A per-history comparison
may therefore allow a handful of forked histories -- but the AGGREGATES (bins, counters, event
tallies) are then still held against the oracle: each forked history is run again on its own, by
the engine and by the oracle, and its own contribution is taken out of the respective total before
the two are compared.  There is no path on which a run passes without an aggregate comparison."""
import numpy as np

from radiative3d_amd import _ffi


def finals_differ(a, b, rtol=1e-9):
    return ((a.fate, a.moves, a.type, a.n_catch) != (b.fate, b.moves, b.type, b.n_catch)
            or abs(a.time - b.time) > rtol * max(1.0, abs(b.time))
            or abs(a.path - b.path) > rtol * max(1.0, abs(b.path))
            or abs(a.amp - b.amp) > rtol)


def production_finals_differ(a, b, rtol=1e-9):
    """The production kernels' records (r3d_engine_set_production_finals) carry no catch count (0xFFFF): every
    other field as finals_differ holds it, and position and direction as well -- those to 100 rtol: a direction
    is the end of a chain of rotations (SphereEarth: 90 scatterings and 170 arcs per history), through which the
    last bits of each travel where a time or a path length only adds them up."""
    import copy
    a2 = copy.copy(a)
    a2.n_catch = b.n_catch
    return (finals_differ(a2, b, rtol)
            or any(abs(x - y) > 100 * rtol * max(1.0, abs(y)) for x, y in zip(a.loc, b.loc))
            or any(abs(x - y) > 100 * rtol for x, y in zip(a.dir, b.dir)))


def forked_ids(finals_engine, finals_oracle, first_id, rtol=1e-9):
    """Ids of the histories whose final records differ between engine and oracle."""
    return [first_id + i for i, (a, b) in enumerate(zip(finals_engine, finals_oracle)) if finals_differ(a, b, rtol)]


def subtract_(res, part):
    """res -= part, every field of a Result (energies: the float difference; integers exactly)."""
    res.energy -= part.energy
    assert (res.counts >= part.counts).all(), "a forked history's own run caught where the batch did not"
    res.counts -= part.counts
    res.n_lost -= part.n_lost
    res.n_timeout -= part.n_timeout
    res.n_invalid -= part.n_invalid
    res.invalid_reasons -= part.invalid_reasons
    for k in res.events:
        res.events[k] -= part.events[k]
    return res


def assert_aggregates_equal(got, want, what="", rtol=1e-9, atol=1e-13):
    """Every integer output bit-exact (counts per seismometer, bin and type; lost / timeout / invalid; the
    eight event counters), the invalid reasons as the per-history suite holds them (their sum: which of
    "negative time" / "stuck" / "slow" a trapped phonon is filed under sits on the sign of rounding
    noise), energies to `rtol` relative."""
    assert (got.n_lost, got.n_timeout, got.n_invalid) == (want.n_lost, want.n_timeout, want.n_invalid), what
    assert got.events == want.events, what
    assert int(got.invalid_reasons.sum()) == int(want.invalid_reasons.sum()) == got.n_invalid, what
    assert (got.invalid_reasons[[0, 1, 2, 6]] == want.invalid_reasons[[0, 1, 2, 6]]).all(), what
    assert (got.counts == want.counts).all(), what
    assert np.allclose(got.energy, want.energy, rtol=rtol, atol=atol), what


def assert_aggregates_equal_without(got, want, forked, run_engine, run_oracle, what="", rtol=1e-9, atol=1e-13):
    """The same with the histories in `forked` (ids) taken out of both sides first: run_engine(n, first_id)
    and run_oracle(n, first_id) return the Result of that id range alone.  `got` and `want` are left
    as they were.  With no forked history this is assert_aggregates_equal."""
    if forked:
        import copy
        got, want = copy.deepcopy(got), copy.deepcopy(want)
        for hid in forked:
            subtract_(got, run_engine(1, hid))
            subtract_(want, run_oracle(1, hid))
        what = f"{what} (forked histories taken out of both sides: ids {forked})"
        # (the energies of a bin that lost a forked history's catch are differences now: absolute
        #  tolerance at the size of what was subtracted)
        atol = max(atol, rtol * float(np.abs(want.energy).max(initial=0.0)))
    assert_aggregates_equal(got, want, what, rtol, atol)
