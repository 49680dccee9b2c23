"""Parity tests proper: the HIP engine, called through the C-ABI
(libr3d_hip.so), against the oracle on the same seeded histories -- history by
history at sizes the oracle finishes in seconds, and through size-independent
properties at the full BASELINE size."""
import os

import numpy as np
import pytest
import torch

from conftest import finals_differ
from oracle import oracle_ffi as O
from oracle.check import assert_aggregates_equal, assert_aggregates_equal_without, forked_ids, production_finals_differ
from radiative3d_amd import Engine, Model, _ffi
from radiative3d_amd.parallel import DeviceResult, shard_range
from tests.configs import crustpinch, halfspace, lopnor, sphere_deep


def energies_agree(a, b, tol=1e-11):
    """Two ENGINE runs of the same histories, batched differently (another partition, a chain, other streams, shards):
    integer outputs are identical; a history's numbers are defined to rounding -- which wave serves it, and whether its
    interface solve runs in the R/T phase or inside a thin batch's MOVE phase (two inlinings of the same code, fused
    into multiply-adds differently), depends on the batching -- and a component that is tiny against its bin's energy
    (particle motion all but normal to that axis) moves in its leading digits.  So: every component to `tol` of its
    BIN's energy by type (observed: 6e-13, tools/partition_deviation.py; the reproducible build is bit-identical)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64).reshape(np.shape(a))
    scale = a[..., 3:].sum(-1, keepdims=True)
    return bool(np.all(np.abs(a - b) <= tol * scale + 1e-300))


pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engines(models):
    cache = {}

    def get(name, deg=4, extra=()):
        key = (name, deg, tuple(extra))
        if key not in cache:
            cache[key] = Engine(models(name, deg, extra))
        return cache[key]
    return get


def check_against_oracle(engine, n, first_id=0, seed=0x5EED, allow_frac=0.0):
    """History by history against the oracle, then the aggregates.  allow_frac: share of histories
    that may fork (device libm vs glibc in the last ulp under a branch decision; none has been
    seen, and only the TOA-degree-9 samples ask for any): their ids are printed, and the aggregates
    are compared all the same, with those histories' own contributions taken out of both sides
    (oracle/check.py) -- no path passes without an aggregate comparison."""
    model = engine.model
    rg, fg = engine.run(n, first_id, seed, trace=True)
    ro, fo = O.run(model, n, first_id, seed, trace=True)
    forked = forked_ids(fg, fo, first_id)
    if forked:
        print(f"forked histories (engine vs oracle): ids {forked}")
    assert len(forked) <= int(allow_frac * n), f"{len(forked)} of {n} histories differ from the oracle: ids {forked[:20]}"
    assert rg.n_lost + rg.n_timeout + rg.n_invalid == n
    assert_aggregates_equal_without(rg, ro, forked, lambda k, i: engine.run(k, i, seed),
                                    lambda k, i: O.run(model, k, i, seed), "diagnostic kernel")
    return rg, ro


def assert_result_equals_oracle(got, want, what):
    """Aggregate comparison for runs without per-history records (oracle/check.py)."""
    assert_aggregates_equal(got, want, what)


def check_production_against_oracle(engine, n, first_id=0, seed=0x5EED, pieces=4):
    """The code objects that ship and are timed -- pool_kernel<..., TRACE = false> (r3d_run,
    r3d_run_device[_carry]) and pool_drain_kernel (a chain's flush) -- against the oracle, not
    against the diagnostic kernel or themselves (reference phonons.cpp:540-682, dataout.cpp:103-216)."""
    model = engine.model
    want, want_finals = O.run(model, n, first_id, seed, trace=True)
    # (the final records these kernels themselves leave, history by history: r3d_engine_set_production_finals --
    #  the self-contained kernel writes all of them, the drain kernel those that end in a chain's flush)
    engine.set_production_finals(first_id, n)
    got = engine.run(n, first_id, seed)                       # one self-contained production launch
    assert_result_equals_oracle(got, want, "r3d_run")
    mine = engine.production_finals(0, n) if n else []
    assert all(f.fate != 255 and f.n_catch == 0xFFFF for f in mine), "a history left no final record"
    differ = [first_id + i for i, (a, b) in enumerate(zip(mine, want_finals)) if production_finals_differ(a, b)]
    assert not differ, f"self-contained production kernel: {len(differ)} of {n} final records differ from the oracle: ids {differ[:20]}"
    engine.set_production_finals(first_id, n)                 # (a fresh buffer for the chain)
    # the same ids as a carry chain of `pieces` launches + a flush-only launch (n = 0: the drain kernel)
    total = DeviceResult(model, "cuda:0")
    per = n // pieces
    for k in range(pieces):
        count = per if k < pieces - 1 else n - per * (pieces - 1)
        engine.run_device(count, first_id + k * per, seed, *total.pointers(), carry="carry")
    torch.cuda.synchronize()
    assert engine.carry_pending or n == 0
    engine.run_device(0, 0, seed, *total.pointers(), carry="final")
    torch.cuda.synchronize()
    assert not engine.carry_pending
    assert_result_equals_oracle(total.to_result(), want, "carry chain + drain")
    mine = engine.production_finals(0, n) if n else []
    written = [i for i, f in enumerate(mine) if f.fate != 255]   # histories that ended in the flush (the step kernel writes none)
    assert n < 1000 or len(written) > n // 4, (len(written), n)
    differ = [first_id + i for i in written if production_finals_differ(mine[i], want_finals[i])]
    assert not differ, f"drain kernel: {len(differ)} of {len(written)} final records differ from the oracle: ids {differ[:20]}"
    engine.set_production_finals(0, 0)
    return got, want


# (cell kind, table residency) of every compiled traversal kernel and a model that runs it; residency 1 / 2
# on models whose tables would fit in LDS is asked for through r3d_engine_create_ex (Engine(residency=...)).  tests/test_kernel_coverage.py (CPU)
# holds this list against the symbols in libr3d_hip.so.
KERNEL_CASES = [(0, 0, "lopnor", 6000), (0, 1, "lopnor", 6000), (0, 2, "halfspace", 20000),
                (1, 1, "crustpinch", 8000), (1, 2, "upthrust", 8000),
                (2, 0, "sphere_deep", 1500), (2, 1, "sphere_deep", 1500), (2, 2, "toysphere_vids", 1500)]


@pytest.mark.parametrize("kind,res,name,n", KERNEL_CASES)
def test_every_compiled_kernel_matches_oracle(models, monkeypatch, kind, res, name, n):
    """pool_kernel<kind, res, TRACE>, pool_kernel<kind, res, production> and pool_drain_kernel<kind, res>
    each against the oracle; the engine says which variant it launches."""
    e = Engine(models(name), residency=res)
    assert e.variant == (kind, res)
    check_against_oracle(e, n, first_id=17)                   # the diagnostic kernel (final records)
    check_production_against_oracle(e, n, first_id=17)        # the production kernel and the drain kernel
    e.close()


@pytest.mark.parametrize("name,n", [("halfspace", 50000), ("crustpinch", 20000), ("lopnor", 20000),
                                    ("sphere_deep", 3000), ("upthrust", 20000), ("crustpinch_vids", 5000)])
def test_production_kernel_matches_oracle(engines, name, n):
    check_production_against_oracle(engines(name), n)


def test_production_kernel_small_pool_and_many_receivers(models, monkeypatch):
    """The production kernels under the queue stress of test_small_pool_many_short_launches (the
    smallest pool, no bin accumulators) and with 4500 receivers (every catch through the hash)."""
    for name, n in (("crustpinch", 20000), ("lopnor", 20000), ("sphere_deep", 2000)):
        e = Engine(models(name), pool_slots=64, accumulator_bits=0)   # (64: clamped up to the workgroup size)
        assert e.pool_slots == 768 and e.accumulators == 0
        check_production_against_oracle(e, n, first_id=3, pieces=7)
        e.close()
    args = [a for a in halfspace(4) if not a.startswith("--seis-p2p")] + [
        "--seis-p2p=0,0,0,183.85,183.85,0,2.737,0.105,10.0,1500",
        "--seis-p2p=0,0,0,260,0,0,2.737,0.105,10.0,1500",
        "--seis-p2p=0,0,0,240.21,-99.5,0,2.737,0.105,10.0,1500"]
    e = Engine(Model(args))
    got, _ = check_production_against_oracle(e, 20000)
    assert got.events["catch"] > 100
    e.close()


@pytest.mark.parametrize("name,n", [("halfspace", 50000), ("crustpinch", 20000), ("lopnor", 20000),
                                    ("sphere", 3000), ("sphere_deep", 3000), ("toysphere_vids", 2000),
                                    ("lopnor_vids", 2000), ("upthrust", 20000), ("crustpinch_vids", 5000),
                                    ("lopnor_moho", 10000), ("lopnor_moho_sel3", 3000), ("scat_params_study", 20000)])
def test_engine_matches_oracle_history_by_history(engines, name, n):
    """Every model of the reference's dispatcher (user.cpp:69-125) through the engine, history by history against the
    oracle: the four benchmark models, the video runs, the upthrust grid, the Lop Nor model with its Moho layers
    (selectors 1-4 given 20 arguments) and the scattering-parameter study (128)."""
    check_against_oracle(engines(name), n)


def test_other_seed_and_high_ids(engines):
    check_against_oracle(engines("crustpinch"), 5000, first_id=2**40 + 3, seed=0x1234567890ABCDEF)


def test_edge_batches(engines):
    e = engines("halfspace")
    r = e.run(0)
    assert r.events["generated"] == 0 and r.counts.sum() == 0
    for n in (1, 63, 64, 65, 255, 257):
        check_against_oracle(e, n, first_id=777)


def test_no_deflect_override(engines):
    e = engines("crustpinch", 4, ("--overridemfp=25,50", "--nodeflect", "--timetolive=350"))
    rg, _ = check_against_oracle(e, 4000)
    assert rg.events["scatter"] / 4000 > 3


def test_zero_and_single_receiver():
    m0 = Model([a for a in halfspace(4) if not a.startswith("--seis")])
    check_against_oracle(Engine(m0), 3000)
    m1 = Model(halfspace(4, one_receiver=True))
    check_against_oracle(Engine(m1), 20000)


def test_strong_contrast_interface():
    args = [a.replace("6.40,3.63,2.83,-60,6.40,3.63,2.83,-400", "5.0,2.9,2.5,-20,8.0,4.5,3.3,-400")
            for a in halfspace(4)]
    check_against_oracle(Engine(Model(args)), 20000)


def test_shards_add_up_exactly(engines):
    """Histories are keyed by id: any partition of an id range gives the same
    counts bit for bit and the same energies up to fp64 summation order."""
    e = engines("crustpinch")
    n = 30000
    whole = e.run(n)
    parts = e.model.new_result()
    for r in range(3):
        lo, hi = shard_range(n, r, 3)
        e.run(hi - lo, first_id=lo, result=parts)
    assert (whole.counts == parts.counts).all()
    assert whole.events == parts.events
    assert (whole.n_lost, whole.n_timeout) == (parts.n_lost, parts.n_timeout)
    assert energies_agree(whole.energy, parts.energy)


def test_device_resident_accumulation(engines):
    """r3d_run_device into caller-owned HBM buffers (the multi-GPU path) equals r3d_run."""
    e = engines("lopnor")
    host = e.run(8000, first_id=5)
    dev = DeviceResult(e.model, "cuda:0")
    e.run_device(5000, 5, 0x5EED, *dev.pointers())
    e.run_device(3000, 5005, 0x5EED, *dev.pointers())
    torch.cuda.synchronize()
    assert e.last_kernel_ms() > 0
    got = dev.to_result()
    assert (got.counts == host.counts).all() and got.events == host.events
    assert energies_agree(got.energy, host.energy)


def test_full_size_crustpinch_properties():
    """BASELINE config 2 at full size (TOA degree 9, 1e7 histories): properties
    that do not need the oracle to finish."""
    m = Model(crustpinch(9))
    assert m.n_toa == 20 * 4 ** 9
    e = Engine(m)
    n = 10_000_000
    a = e.run(n, first_id=0)
    assert a.n_lost + a.n_timeout + a.n_invalid == n and a.events["generated"] == n
    assert a.n_invalid == 0
    assert int(a.counts.sum()) == a.events["catch"]
    assert np.allclose(a.energy[:, :, :3].sum(-1), a.energy[:, :, 3:].sum(-1), rtol=1e-10, atol=1e-300)
    # The survey's figures of the unmodified reference (tests/golden/reference_recorded.json), each
    # held to its own Monte-Carlo error (tests/refstats.py): the 1 M-history loss counters to their
    # counting error, the 5000-history event mix to 3 sigma of a 5000-history mean + print rounding,
    # sigma from batch means of that size on the engine.
    import json
    import os
    from refstats import poisson_fraction_tolerance, tolerance
    ref = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_recorded.json")))
    loss = ref["crustpinch_1M_loss_counters"]
    f_ref = loss["timeout"] / loss["_n_histories"]
    assert abs(a.n_timeout / n - f_ref) <= poisson_fraction_tolerance(f_ref, loss["_n_histories"], a.n_timeout / n, n)
    mix = ref["events_per_history"]
    n_ref = mix["_n_histories"]
    small = [e.run(n_ref, first_id=(3 << 40) + k * n_ref) for k in range(64)]
    for k, v in mix["crustpinch"].items():
        means = [r.events[k] / n_ref for r in small]
        tol = tolerance(means, v)
        print(f"crustpinch deg 9 {k:10s} reference {v:8.3f}  engine(1e7) {a.events[k] / n:8.4f}  allowed +-{tol:.3f}")
        assert abs(a.events[k] / n - v) <= tol, (k, a.events[k] / n, v, tol)
    big = [e.run(loss["_n_histories"], first_id=(5 << 40) + k * loss["_n_histories"]) for k in range(8)]
    catches = [r.events["catch"] / loss["_n_histories"] for r in big]
    tol = tolerance(catches, loss["catches_per_history"])
    print(f"crustpinch deg 9 catches/history reference {loss['catches_per_history']}  engine {a.events['catch'] / n:.4f}"
          f"  allowed +-{tol:.4f}")
    assert abs(a.events["catch"] / n - loss["catches_per_history"]) <= tol
    # a disjoint id range is an independent sample: per-bin arrival counts are (compound)
    # Poisson -- a reverberating phonon can be caught more than once per bin -- so their
    # normalised differences look normal with a spread a little above 1; total energy agrees to ~1 %.
    # (A few reverberating histories move a sample's TOTAL catches by +-0.5 %, all bins together: the
    #  second sample is scaled to the first one's total before the differences are formed, so that
    #  this common swing cancels and the mean is held as a per-bin bias would show in it.)
    b = e.run(n, first_id=n)
    z = bin_z(a, b, min_bins=5000)
    bound = bias_bound(e, n)
    print(f"crustpinch deg 9 independent halves: z mean {z.mean():+.4f} (allowed +-{bound:.3f}) std {z.std():.3f} over {z.size} bins")
    assert abs(z.mean()) < bound and 0.9 < z.std() < 1.5, (z.mean(), z.std())
    assert a.energy.sum() == pytest.approx(b.energy.sum(), rel=0.02)
    # small-sample oracle comparison on the big tables too: diagnostic and production kernels
    check_against_oracle(e, 3000, first_id=123456789, allow_frac=0.0005)
    check_production_against_oracle(e, 3000, first_id=123456789)


def bias_bound(e, n, pairs=8):
    """How far the mean of bin_z may sit from zero for two samples of the same code: bins are NOT
    independent -- one long-reverberating history feeds many of them -- so the mean of z over
    thousands of bins does not shrink like 1 / sqrt(bins) (LopNor: -0.11 over 8909 bins between two
    1e7-history samples, eight of the independent-bin standard errors).  A relative swing delta of a
    group of bins with c counts each shifts their z by delta sqrt(c / 2), and delta ~ 1 / sqrt(n),
    c ~ n: the shift does not depend on the sample size.  So the spread of the statistic is MEASURED,
    on `pairs` independent pairs of samples an eighth the size, and the full samples' mean is held to
    four of those standard deviations (+ 0.02): a per-bin bias beyond the sample-to-sample variation
    of the statistic itself fails."""
    m = n // 8
    means = []
    for k in range(pairs):
        a = e.run(m, first_id=(9 << 40) + 2 * k * m)
        b = e.run(m, first_id=(9 << 40) + (2 * k + 1) * m)
        means.append(bin_z(a, b, min_bins=200).mean())
    return 4.0 * float(np.std(means, ddof=1)) + 0.02


def bin_z(a, b, min_bins):
    """Normalised per-bin differences of the arrival counts of two independent samples, the second
    scaled to the first one's total catches: z = (na - s nb) / sqrt(na + s^2 nb), s = Na / Nb, over
    the bins with na + nb >= 50.  Counts are (compound) Poisson -- a reverberating phonon can be caught
    more than once per bin -- so z looks normal with a spread a little above 1; with the common swing
    of the totals scaled out its mean shows a per-bin bias (a kernel change that shifted arrivals
    between bins or lost a share of them in some)."""
    na, nb = a.counts.astype(float), b.counts.astype(float)
    s = na.sum() / nb.sum()
    sel = (na + nb) >= 50
    assert sel.sum() > min_bins
    return (na[sel] - s * nb[sel]) / np.sqrt(na[sel] + s * s * nb[sel])


def independent_halves_agree(e, n, min_bins):
    """Two disjoint id ranges are independent samples (bin_z above)."""
    a, b = e.run(n, first_id=0), e.run(n, first_id=n)
    for r in (a, b):
        assert r.n_lost + r.n_timeout + r.n_invalid == n and r.events["generated"] == n
        assert int(r.counts.sum()) == r.events["catch"]
        assert np.allclose(r.energy[:, :, :3].sum(-1), r.energy[:, :, 3:].sum(-1), rtol=1e-10, atol=1e-300)
    z = bin_z(a, b, min_bins)
    bound = bias_bound(e, n)
    print(f"independent halves: z mean {z.mean():+.4f} (allowed +-{bound:.3f}) std {z.std():.3f} over {z.size} bins")
    assert abs(z.mean()) < bound and 0.9 < z.std() < 1.6, (z.mean(), z.std())
    return a, b


def reference_event_mix(e, name, whole, n):
    """The survey's 5000-history event mix of the reference for this model, at its own error."""
    import json
    import os
    from refstats import tolerance
    ref = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_recorded.json")))
    mix = ref["events_per_history"]
    n_ref = mix["_n_histories"]
    small = [e.run(n_ref, first_id=(3 << 40) + k * n_ref) for k in range(32)]
    for k, v in mix[name].items():
        means = [r.events[k] / n_ref for r in small]
        tol = tolerance(means, v)
        print(f"{name} deg 9 {k:10s} reference {v:8.3f}  engine {whole.events[k] / n:8.4f}  allowed +-{tol:.3f}")
        assert abs(whole.events[k] / n - v) <= tol, (k, whole.events[k] / n, v, tol)


def test_full_size_halfspace_one_receiver():
    """BASELINE config 1 at full size: do-halfspace.sh's arguments (reference do-halfspace.sh:41-101) at TOA
    degree 9 with ONE receiver, as BASELINE.json names it.  The literal job -- 1e5 histories, one self-contained
    launch -- and a 1e7-history run for the size-independent properties (every history ends, the receiver's
    books close, X + Y + Z == P + S), the survey's event mix of the unmodified reference at its own
    Monte-Carlo error, and 3 000 histories of the degree-9 engine against the oracle, history by history on
    the diagnostic kernel and bin by bin on the production, chain-step and drain kernels."""
    m = Model(halfspace(9, one_receiver=True))
    assert m.n_toa == 20 * 4 ** 9 and m.n_seismometers == 1
    e = Engine(m)
    job = e.run(100_000, first_id=(7 << 40))                      # the job as stated
    assert job.n_lost + job.n_timeout + job.n_invalid == 100_000 and job.events["generated"] == 100_000
    assert job.n_invalid == 0 and int(job.counts.sum()) == job.events["catch"]
    n = 10_000_000
    a = e.run(n)
    assert a.n_lost + a.n_timeout + a.n_invalid == n and a.events["generated"] == n and a.n_invalid == 0
    assert int(a.counts.sum()) == a.events["catch"] > 0
    assert np.allclose(a.energy[:, :, :3].sum(-1), a.energy[:, :, 3:].sum(-1), rtol=1e-10, atol=1e-300)
    # the literal job is a sample of the same population: its event rates sit within 5 sigma of the big run's
    for k in ("iterations", "transfer", "reflect", "scatter", "collect"):
        rate, small = a.events[k] / n, job.events[k] / 100_000
        assert abs(small - rate) < 5 * np.sqrt(max(rate, 1e-9) * 4 / 100_000) + 1e-3, (k, small, rate)
    reference_event_mix(e, "halfspace", a, n)
    # the one receiver catches a few histories in a million: an independent half catches a compatible number
    b = e.run(n, first_id=n)
    ca, cb = a.events["catch"], b.events["catch"]
    assert abs(ca - cb) < 6 * np.sqrt(ca + cb) + 10, (ca, cb)
    check_against_oracle(e, 3000, first_id=31415926, allow_frac=0.0005)
    check_production_against_oracle(e, 3000, first_id=31415926)
    e.close()


def test_full_size_lopnor_properties():
    """BASELINE config 3 at its full table size (TOA degree 9: 21 scatterers, 4.4 GB of tables;
    explosion source), 1e7 histories: size-independent properties, the survey's event mix, and a
    small-sample oracle comparison on the big tables."""
    m = Model(lopnor(9))
    assert m.n_toa == 20 * 4 ** 9 and (m.n_cells, m.n_scatterers, m.n_seismometers) == (21, 21, 320)
    e = Engine(m)
    n = 10_000_000
    a, b = independent_halves_agree(e, n, 3000)
    assert a.n_invalid == 0 and a.energy.sum() == pytest.approx(b.energy.sum(), rel=0.03)
    reference_event_mix(e, "lopnor", a, n)
    check_against_oracle(e, 3000, first_id=987654321, allow_frac=0.0005)
    check_production_against_oracle(e, 3000, first_id=987654321)
    e.close()


def test_full_size_sphere_deep_source_properties():
    """BASELINE config 4 at its full table size (TOA degree 9: 15 scatterers, 3.1 GB of tables;
    double-couple source 600 km deep), 2e6 histories.  Every history ends at the time limit (a
    whole Earth has no loss surface).  The survey's event mix was taken with the script's 10 km
    source, so it is checked on that source's engine."""
    m = Model(sphere_deep(9))
    assert m.n_toa == 20 * 4 ** 9 and (m.n_cells, m.n_scatterers, m.n_seismometers) == (15, 15, 480)
    e = Engine(m)
    n = 2_000_000
    a, b = independent_halves_agree(e, n, 3000)
    assert a.n_timeout == n and a.n_invalid == 0
    assert a.events["scatter"] / n > 50 and a.events["rtsolve"] / n > 20
    check_against_oracle(e, 1500, first_id=24680, allow_frac=0.0005)
    check_production_against_oracle(e, 1500, first_id=24680)
    e.close()
    from tests.configs import sphere
    e = Engine(Model(sphere(9)))
    whole = e.run(1_000_000)
    reference_event_mix(e, "sphere", whole, 1_000_000)
    e.close()


def _two_rank_worker(rank, world, port, n, out_path):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = Model(lopnor(4) + ["--device-tables"])      # the multi-rank bench builds its tables on the device
    e = Engine(model, device=0)
    dev = DeviceResult(model, "cuda:0")
    lo, hi = shard_range(n, rank, world)
    e.run_device(hi - lo, lo, 0x5EED, *dev.pointers())
    torch.cuda.synchronize()
    host = DeviceResult(model, "cpu")                   # gloo reduces host tensors; RCCL needs a GPU per rank
    host.energy.copy_(dev.energy), host._ints.copy_(dev._ints)
    host.allreduce_()
    if rank == 0:
        r = host.to_result()
        np.savez(out_path, energy=r.energy, counts=r.counts, scalars=r.scalars())
    dist.barrier()
    e.close()
    dist.destroy_process_group()


def test_two_ranks_on_the_gpu_sum_to_one_engine_run(engines, tmp_path):
    """The multi-rank path end to end with real engines: two rank processes (both on this box's one
    GPU), device-built tables, id shards, r3d_run_device into DeviceResult buffers, one all-reduce
    (gloo here -- RCCL wants a GPU per rank) -- equal to a single engine's run of the union range."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    n = 200_001
    out = str(tmp_path / "two.npz")
    mp.spawn(_two_rank_worker, args=(2, port, n, out), nprocs=2, join=True)
    got = np.load(out)
    want = engines("lopnor", 4, ("--device-tables",)).run(n)
    assert (got["counts"] == want.counts).all() and (got["scalars"] == want.scalars()).all()
    assert energies_agree(got["energy"], want.energy)


def test_engine_leaves_the_callers_device_and_refuses_to_drop_carried_histories(engines):
    e = engines("halfspace")
    before = torch.cuda.current_device()
    e.run(1000)
    assert torch.cuda.current_device() == before
    buf = DeviceResult(e.model, "cuda:0")
    e.run_device(200000, 0, 9, *buf.pointers(), carry="carry")
    torch.cuda.synchronize()
    assert e.carry_pending
    with pytest.raises(RuntimeError, match="carried"):
        e.close()
    e.run_device(0, 0, 9, *buf.pointers(), carry="final")
    torch.cuda.synchronize()
    assert not e.carry_pending
    r = buf.to_result()
    assert r.n_lost + r.n_timeout + r.n_invalid == 200000
    # launches are timed one by one
    k = e.launch_count()
    assert e.kernel_ms(k) > 0 and e.kernel_ms(k - 1) > 0 and e.kernel_ms(k) != e.kernel_ms(k - 1)
    assert e.kernel_ms(k + 1) == -1.0 and e.kernel_ms(0) == -1.0


def test_one_call_seam_equals_engine_run(engines):
    """r3d_run_model (SURVEY.md 8(b)'s one-call form) on one device == Engine.run; asking for
    a device that is not there fails with a message instead of running on the host."""
    from radiative3d_amd import run_model
    e = engines("lopnor")
    want = e.run(7000, first_id=11, seed=99)
    got = run_model(e.model, 7000, first_id=11, seed=99, n_gpus=1)
    assert (got.counts == want.counts).all() and got.events == want.events
    assert (got.n_lost, got.n_timeout, got.n_invalid) == (want.n_lost, want.n_timeout, want.n_invalid)
    assert energies_agree(got.energy, want.energy)
    import torch
    with pytest.raises(RuntimeError, match="device index out of range"):
        run_model(e.model, 100, n_gpus=torch.cuda.device_count() + 1)
    with pytest.raises(RuntimeError, match="no CPU path"):
        run_model(e.model, 100, n_gpus=0)


def test_thousands_of_receivers_tables_in_hbm():
    """4500 receivers: the scan table (252 KB) cannot live in a CU's LDS, so the engine runs the
    kernel variant that reads the scatterer / receiver tables from HBM; results as ever."""
    args = [a for a in halfspace(4) if not a.startswith("--seis-p2p")] + [
        "--seis-p2p=0,0,0,183.85,183.85,0,2.737,0.105,10.0,1500",
        "--seis-p2p=0,0,0,260,0,0,2.737,0.105,10.0,1500",
        "--seis-p2p=0,0,0,240.21,-99.5,0,2.737,0.105,10.0,1500"]
    m = Model(args)
    assert m.n_seismometers == 4500
    rg, ro = check_against_oracle(Engine(m), 20000)
    assert rg.events["catch"] > 100


def test_invalid_phonons_are_reported_like_the_oracle(engines):
    """The sheared crust-upthrust model has degenerate cells: ~4 % of the histories end as
    INVALID, trapped where the travel time of a leg is zero to rounding.  Same histories, same
    move counts, same report lines; WHICH of "negative time" / "stuck" / "slow" such a phonon
    is filed under depends on the sign of that rounding noise (recent time -1e-17, 0 or
    +1e-17), so only the three reasons' sum is compared."""
    e = engines("upthrust")
    rg, ro = check_against_oracle(e, 20000)
    assert rg.n_invalid == ro.n_invalid > 300
    assert int(rg.invalid_reasons[3:6].sum()) == int(ro.invalid_reasons[3:6].sum()) == rg.n_invalid
    assert rg.diag_invalid != 0 and ro.diag_invalid != 0
    e.set_event_log(mask=128, capacity=1 << 16)          # INV lines only
    e.run(20000)
    ev = e.read_event_log()
    e.set_event_log(mask=0, capacity=0)
    assert len(ev) == rg.n_invalid and set(ev["tag"]) == {7}


@pytest.mark.parametrize("name,n,parts", [("crustpinch", 400000, 4), ("lopnor", 300000, 3), ("sphere", 20000, 5)])
def test_carry_chain_equals_one_run(engines, name, n, parts):
    """r3d_run_device_carry: a chain of launches that hand their unfinished histories to the
    next launch, then a flush, sums to exactly what one self-contained run of the same ids
    gives; the pieces themselves differ (a launch holds what it executed)."""
    e = engines(name)
    want = e.run(n, first_id=5, seed=77)
    total, step = DeviceResult(e.model, "cuda:0"), DeviceResult(e.model, "cuda:0")
    per = n // parts
    first_piece = None
    for k in range(parts):
        step.zero_()
        e.run_device(per, 5 + k * per, 77, *step.pointers(), carry="carry")
        torch.cuda.synchronize()
        if k == 0:
            first_piece = step.to_result()
        total.add_(step)
    step.zero_()
    e.run_device(n - per * parts, 5 + per * parts, 77, *step.pointers(), carry="final")   # flush (n may be 0)
    torch.cuda.synchronize()
    total.add_(step)
    got = total.to_result()
    assert (got.counts == want.counts).all() and got.events == want.events
    assert (got.n_lost, got.n_timeout, got.n_invalid) == (want.n_lost, want.n_timeout, want.n_invalid)
    assert energies_agree(got.energy, want.energy)
    # the first launch started `per` histories but left some unfinished for the second
    assert first_piece.events["generated"] == per
    assert first_piece.n_lost + first_piece.n_timeout + first_piece.n_invalid < per
    # a chain under another seed is refused while histories are pending
    e.run_device(1000, 10**9, 77, *step.pointers(), carry="carry")
    with pytest.raises(RuntimeError, match="another seed"):
        e.run_device(1000, 10**9 + 1000, 78, *step.pointers(), carry="carry")
    e.run_device(0, 0, 77, *step.pointers(), carry="final")
    torch.cuda.synchronize()


def test_launches_in_flight_on_two_streams(engines):
    """Self-contained r3d_run_device launches of one engine on different streams, into
    different buffers, may overlap (each takes its own work counter): same sums as one run."""
    e = engines("crustpinch")
    n, parts = 240000, 6
    want = e.run(n, first_id=0, seed=5)
    streams = [torch.cuda.Stream("cuda:0") for _ in range(2)]
    bufs = [DeviceResult(e.model, "cuda:0") for _ in range(2)]
    per = n // parts
    for k in range(parts):
        with torch.cuda.stream(streams[k % 2]):
            e.run_device(per, k * per, 5, *bufs[k % 2].pointers(), stream=streams[k % 2].cuda_stream)
    torch.cuda.synchronize()
    got = bufs[0].add_(bufs[1]).to_result()
    assert (got.counts == want.counts).all() and got.events == want.events
    assert energies_agree(got.energy, want.energy)


def test_report_stream_over_a_carry_chain(engines):
    """Histories that span two launches of a chain keep their event order in the report buffer."""
    e = engines("lopnor")
    n = 6000
    _, ev_o, total = O.run_with_events(e.model, n, capacity=1 << 20)
    e.set_event_log(capacity=1 << 20)
    buf = DeviceResult(e.model, "cuda:0")
    for k in range(3):
        e.run_device(n // 3, k * (n // 3), 0x5EED, *buf.pointers(), carry="carry")
    e.run_device(0, 0, 0x5EED, *buf.pointers(), carry="final")
    torch.cuda.synchronize()
    ev_g = e.read_event_log(reset=True)
    e.set_event_log(mask=0, capacity=0)
    assert len(ev_g) == total

    def by_history(ev):
        order = np.argsort(ev["id"], kind="stable")
        return ev[order]
    a, b = by_history(ev_o), by_history(ev_g)
    assert np.array_equal(a["id"], b["id"]) and np.array_equal(a["tag"], b["tag"])
    assert np.array_equal(a["moves"], b["moves"]) and np.allclose(a["time"], b["time"], rtol=1e-9, atol=1e-9)


def test_small_pool_many_short_launches(models, monkeypatch):
    """The pool kernel's queues under stress: the smallest pool the engine accepts (a slot per lane of
    the workgroup, so the waves compete for every slot and the rings wrap constantly), no bin
    accumulators, and a carry chain of many launches far smaller than the pool (most workgroups
    find the id counter exhausted at once and park an almost empty pool) -- against the oracle
    and against one self-contained run."""
    for name, n in (("crustpinch", 20000), ("lopnor", 20000), ("sphere_deep", 2000)):
        e = Engine(models(name), pool_slots=64, accumulator_bits=0)      # (64: clamped up to the workgroup size)
        check_against_oracle(e, n, first_id=3)
        want = e.run(n, first_id=10**6, seed=21)
        total, step = DeviceResult(e.model, "cuda:0"), DeviceResult(e.model, "cuda:0")
        per = 777
        first = 10**6
        while first < 10**6 + n:
            m = min(per, 10**6 + n - first)
            step.zero_()
            e.run_device(m, first, 21, *step.pointers(), carry="carry")
            total.add_(step)
            first += m
        step.zero_()
        e.run_device(0, 0, 21, *step.pointers(), carry="final")
        torch.cuda.synchronize()
        total.add_(step)
        got = total.to_result()
        assert (got.counts == want.counts).all() and got.events == want.events
        assert energies_agree(got.energy, want.energy)
        e.close()


def test_run_model_on_three_engines_on_one_device(engines):
    """r3d_run_model_on (SURVEY.md 8(b)'s n_gpus seam with the devices named): three engines, three
    host threads, all on this box's one GPU, host sum == one engine's run of the union (counts
    exact, energies to summation order) == the oracle; a failing shard fails the call, names
    itself and leaves *out untouched.  Reference: model.cpp:602-633, combine.m:26-33."""
    from radiative3d_amd import run_model
    e = engines("lopnor")
    n = 9001
    want = e.run(n, first_id=11, seed=99)
    got = run_model(e.model, n, first_id=11, seed=99, devices=[0, 0, 0])
    assert (got.counts == want.counts).all() and got.events == want.events
    assert (got.n_lost, got.n_timeout, got.n_invalid) == (want.n_lost, want.n_timeout, want.n_invalid)
    assert energies_agree(got.energy, want.energy)
    assert_result_equals_oracle(got, O.run(e.model, n, 11, 99), "r3d_run_model_on")
    with pytest.raises(RuntimeError, match=r"shard 1 \(device 99\): device index out of range"):
        run_model(e.model, 100, devices=[0, 99, 0])
    with pytest.raises(RuntimeError, match="at least one device"):
        run_model(e.model, 100, devices=[])


def test_node_sums_the_shards_and_the_library_communicator_reduces_a_block(engines):
    """r3d_node_* and r3d_comm_*: the product's own reduction (north_star: "an RCCL reduce over xGMI of the per-receiver
    energy-envelope histograms at the end"; semantics combine.m:26-33).  Both go through ONE function
    (csrc/r3d_rccl.h reduce_block).  On the box's one GPU: a node of one shard has nothing to reduce (its block is the
    job's: "host", with the reason), a node whose shards share the device takes the host sum, both equal Engine.run
    and the oracle (counts exactly, energies to summation order); the library's communicator is formed at one rank
    (ncclGetUniqueId / ncclCommInitRank through the lazily bound librccl), reports its size from ncclCommCount and the
    device's UUID, and its reduce -- all-reduce and reduce to a root -- leaves a one-rank block as it was; a node is
    reusable -- a second run adds into the caller's block without rebuilding tables; a failing run leaves *out untouched."""
    import ctypes as C
    import torch
    from radiative3d_amd import Node, _ffi
    from radiative3d_amd.parallel import Comm, DeviceResult
    e = engines("crustpinch")
    n = 6001
    want = e.run(n, first_id=5, seed=77)
    one = Node(e.model, [0])
    assert one.reduction == "host" and "one shard" in one.reduction_note and len(one) == 1
    got = one.run(n, first_id=5, seed=77)
    assert (got.counts == want.counts).all() and got.events == want.events
    assert (got.n_lost, got.n_timeout, got.n_invalid) == (want.n_lost, want.n_timeout, want.n_invalid)
    assert energies_agree(got.energy, want.energy)
    assert_result_equals_oracle(got, O.run(e.model, n, 5, 77), "r3d_node_run")
    # the communicator of a one-rank job, without any process group around it
    L = _ffi.hip_lib()
    ident = C.create_string_buffer(_ffi.R3D_COMM_ID_BYTES)
    assert L.r3d_comm_unique_id(ident) == 0, L.r3d_last_error()
    handle = L.r3d_comm_create(ident, 0, 1, 0)
    assert handle, L.r3d_last_error()
    comm = Comm(handle, L)
    info = comm.describe()
    assert info["n_ranks"] == 1 and info["rank"] == 0 and info["device"] == 0 and info["rccl_version"] > 20000
    assert len(info["device_uuid"]) == 32 and "librccl" in info["library"]
    block = DeviceResult(e.model, "cuda:0", comm)
    e.run_device(n, 5, 77, *block.pointers())
    torch.cuda.synchronize()
    before = block.to_result()
    block.allreduce_()                                                     # ncclAllReduce x 3, grouped
    comm.reduce_(block.energy, block.counts, block.scalars, root=0)        # ncclReduce x 3, grouped
    torch.cuda.synchronize()
    after = block.to_result()
    assert (after.counts == before.counts).all() and after.events == before.events and (after.energy == before.energy).all()
    assert (after.counts == want.counts).all() and after.events == want.events
    assert L.r3d_comm_reduce(handle, None, 0, None, 0, None, 0, 5, None) != 0 and b"no such root" in L.r3d_last_error()
    comm.close()
    three = Node(e.model, [0, 0, 0])
    assert three.reduction == "host" and len(three) == 3
    host = three.run(n, first_id=5, seed=77)
    assert (host.counts == got.counts).all() and host.events == got.events
    assert energies_agree(host.energy, got.energy)
    # the same node again: its result is ADDED to what the caller's block holds
    again = one.run(n, first_id=5, seed=77, result=got)
    assert again is got and (got.counts == 2 * want.counts).all() and got.events["generated"] == 2 * n
    assert energies_agree(got.energy, 2 * want.energy)
    one.close(), three.close()
    with pytest.raises(RuntimeError, match=r"shard 1 \(device 99\): device index out of range"):
        Node(e.model, [0, 99])


@pytest.mark.parametrize("name,per_launch", [("crustpinch", 150_000), ("lopnor", 150_000), ("sphere_deep", 120_000)])
def test_histories_that_lived_in_the_chain_step_kernel_end_with_the_oracles_records(engines, name, per_launch):
    """The chain-step kernel -- the code object the headline is measured on -- carries no final-record code (it cost that
    kernel 3.5 %).  Its per-history work is witnessed through the histories it hands on: four step launches that together
    bring more histories than the pool has slots (262 144), so that the later launches run for real -- histories move,
    scatter, reflect and END inside step kernels to make room --, and the quarter of a million still in flight at the
    end are finished by the drain kernel, which writes their records.  Every one of those records -- positions, times,
    path lengths, move counts of histories that spent most of their moves in step kernels -- must be the oracle's."""
    e = engines(name)
    n = 4 * per_launch
    first, seed = 50_000_000, 0xBEEF
    # Times and path lengths: 1e-9 relative, as everywhere -- except the whole-Earth shells, 1e-8.  The engine's shell move
    # is the local form (good to rounding); the oracle's is the reference's construction, whose arc lengths are good to
    # ~1e-11 of the arc's RADIUS (tests/test_face_filter.py `dev_over_R`), and a steep ray's radius reaches 1e6 km: a
    # history that bounces up and down 170 times ends 4e-9 of its 50 000 km from the oracle's record (2 of these 60 000
    # pass 1e-9; none passes 5e-9).  Integer fields -- fate, type, moves -- are exact: no history forks.
    rtol = 1e-8 if name == "sphere_deep" else 1e-9
    e.set_production_finals(first, n)
    total = DeviceResult(e.model, "cuda:0")
    for k in range(4):
        e.run_device(per_launch, first + k * per_launch, seed, *total.pointers(), carry="carry")
    torch.cuda.synchronize()
    assert e.carry_pending
    e.run_device(0, 0, seed, *total.pointers(), carry="final")
    torch.cuda.synchronize()
    got = total.to_result()
    assert got.events["generated"] == n and got.n_lost + got.n_timeout + got.n_invalid == n
    mine = e.production_finals(0, n)
    e.set_production_finals(0, 0)
    written = [i for i in range(n) if mine[i].fate != 255]
    assert 100_000 < len(written) < n - 100_000, len(written)        # the drain finished many; step kernels finished many too
    # (the oracle on the written histories only, in runs of consecutive ids)
    differ, checked = [], 0
    # group consecutive indices
    runs, start, prev = [], written[0], written[0]
    for i in written[1:]:
        if i != prev + 1:
            runs.append((start, prev + 1))
            start = i
        prev = i
    runs.append((start, prev + 1))
    # the oracle is a scalar program (~3e4 histories/s): a sample of the runs, 60 000 histories at most
    budget = 60_000
    for lo, hi in runs:
        if budget <= 0:
            break
        hi = min(hi, lo + budget)
        _, want = O.run(e.model, hi - lo, first + lo, seed, trace=True)
        differ += [first + lo + j for j in range(hi - lo) if production_finals_differ(mine[lo + j], want[j], rtol)]
        checked += hi - lo
        budget -= hi - lo
    assert checked >= 20_000, checked
    assert not differ, f"{name}: {len(differ)} of {checked} records of histories carried through step launches differ from the oracle: ids {differ[:20]}"


def finals_bytes(f):
    """A final record (or a few) as its bytes."""
    import ctypes
    if isinstance(f, (list, tuple)):
        return b"".join(finals_bytes(x) for x in f)
    return bytes(ctypes.string_at(ctypes.addressof(f), ctypes.sizeof(f)))


STEP_FINALS_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "radiative3d_amd", "lib",
                               "variant_STEPFINALS.so")


@pytest.mark.parametrize("name,per_launch", [("crustpinch", 150_000), ("lopnor", 150_000), ("sphere_deep", 120_000)])
def test_chain_step_kernel_with_its_records_compiled_in_matches_the_oracle_per_history(name, per_launch):
    """A direct witness of the chain-step kernel's code, with a caveat.  The shipped step kernel
    (pool_kernel<kind, ., ., false>, the code object the headline is timed on) is compiled WITHOUT the final-record
    stores (3.5 % of its launch); `make variant NAME=STEPFINALS DEFS=-DR3D_STEP_FINALS=1` (built by
    __graft_entry__.build()) compiles the SAME sources with the SAME flags and those stores in -- a sibling
    compilation: same code, its own register allocation and schedule, not the shipped binary.  Its step launches write
    a record for every history that ENDS in them; those records are read BEFORE the chain's flush (so the drain kernel
    has written none of them) and held against the oracle history by history, for the three cell kinds.  The shipped
    binary itself stays held through its bins, counters and the drain's records of the histories it hands on (the test
    above).  Reference loop: phonons.cpp:540-682."""
    from radiative3d_amd.configs import CONFIGS
    if not os.path.exists(STEP_FINALS_LIB):   # (__graft_entry__.build() makes it; a tree that was built by `make` alone: here)
        import subprocess
        repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        subprocess.run(["make", "-C", repo, "variant", "NAME=STEPFINALS", "DEFS=-DR3D_STEP_FINALS=1"], capture_output=True, timeout=900)
    assert os.path.exists(STEP_FINALS_LIB), "variant_STEPFINALS.so is not built (__graft_entry__.build() makes it)"
    e = Engine(Model(CONFIGS[name](4)), lib=STEP_FINALS_LIB)
    n = 4 * per_launch
    first, seed = 70_000_000, 0xFEED
    rtol = 1e-8 if name == "sphere_deep" else 1e-9      # (as in the test above)
    e.set_production_finals(first, n)
    total = DeviceResult(e.model, "cuda:0")
    for k in range(4):
        e.run_device(per_launch, first + k * per_launch, seed, *total.pointers(), carry="carry")
    torch.cuda.synchronize()
    assert e.carry_pending
    step_written = e.production_finals(0, n)             # before the flush: only step launches have run
    ended = [i for i in range(n) if step_written[i].fate != 255]
    assert len(ended) > 100_000, len(ended)              # histories do end inside step kernels, and leave their record
    e.run_device(0, 0, seed, *total.pointers(), carry="final")
    torch.cuda.synchronize()
    got = total.to_result()
    assert got.events["generated"] == n and got.n_lost + got.n_timeout + got.n_invalid == n
    after = e.production_finals(0, n)
    e.set_production_finals(0, 0)
    assert all(f.fate != 255 for f in after)             # the drain wrote the rest: every history left a record
    assert all(finals_bytes([after[i]]) == finals_bytes([step_written[i]]) for i in ended[:5000])   # (written once)
    # the oracle on a sample of the histories that ended in step kernels, in runs of consecutive ids
    differ, checked, budget = [], 0, 30_000
    lo = 0
    while lo < len(ended) and budget > 0:
        hi = lo
        while hi + 1 < len(ended) and ended[hi + 1] == ended[hi] + 1 and hi - lo < 2000:
            hi += 1
        a, b = ended[lo], ended[hi] + 1
        _, want = O.run(e.model, b - a, first + a, seed, trace=True)
        differ += [first + a + j for j in range(b - a) if production_finals_differ(step_written[a + j], want[j], rtol)]
        checked += b - a
        budget -= b - a
        lo = hi + 1
    assert checked >= 10_000, checked
    assert not differ, f"{name}: {len(differ)} of {checked} records written by step launches differ from the oracle: ids {differ[:20]}"
    e.close()


def test_production_finals_buffer_rules(engines):
    """r3d_engine_set_production_finals: while a buffer is attached a launch whose ids it does not cover is refused
    (the kernel indexes the buffer by the history id itself), it cannot be attached or detached while histories are
    carried over, records never written say so (fate 255), and detaching restores the plain run."""
    e = engines("halfspace")
    buf = DeviceResult(e.model, "cuda:0")
    e.set_production_finals(1000, 500)
    with pytest.raises(RuntimeError, match="does not cover this launch's ids"):
        e.run(10, first_id=990)
    with pytest.raises(RuntimeError, match="does not cover this launch's ids"):
        e.run(10, first_id=1495)
    e.run(100, first_id=1200)
    fin = e.production_finals(0, 500)
    assert all(f.fate == 255 for f in fin[:200]) and all(f.fate in (1, 2, 3) for f in fin[200:300]) and all(f.fate == 255 for f in fin[300:])
    with pytest.raises(RuntimeError, match="range beyond the buffer"):
        e.production_finals(400, 200)
    e.run_device(50, 1300, 5, *buf.pointers(), carry="carry")
    torch.cuda.synchronize()
    if e.carry_pending:
        with pytest.raises(RuntimeError, match="carried over"):
            e.set_production_finals(0, 0)
    e.run_device(0, 0, 5, *buf.pointers(), carry="final")
    torch.cuda.synchronize()
    e.set_production_finals(0, 0)
    with pytest.raises(RuntimeError, match="no production finals attached"):
        e.production_finals(0, 1)
    e.run(10, first_id=990)                                            # any ids again


def test_run_device_rejects_unknown_carry_words_and_keeps_launch_ids_in_step(engines):
    e = engines("halfspace")
    buf = DeviceResult(e.model, "cuda:0")
    for bad in ("Final", True, "flush", 1):
        with pytest.raises(ValueError, match="carry must be"):
            e.run_device(10, 0, 1, *buf.pointers(), carry=bad)
    # a rejected launch takes no launch id and no timing slot: ids and times stay in step
    before = e.launch_count()
    e.run_device(50000, 0, 5, *buf.pointers(), carry="carry")
    with pytest.raises(RuntimeError, match="another seed"):
        e.run_device(1000, 50000, 6, *buf.pointers(), carry="carry")
    assert e.launch_count() == before + 1 and e.carry_pending
    e.run_device(3_000_000, 50000, 5, *buf.pointers(), carry="carry")
    e.run_device(0, 0, 5, *buf.pointers(), carry="final")
    torch.cuda.synchronize()
    assert e.launch_count() == before + 3
    small, big = e.kernel_ms(before + 1), e.kernel_ms(before + 2)
    assert 0 < small < big, (small, big)       # launch before+2 is the 3e6-history one


def test_reproducible_build_defines_every_history_to_the_bit(models, monkeypatch):
    """libr3d_hip_repro.so (-DR3D_REPRODUCIBLE: no wave-voted series choice, csrc/r3d_math.h): a history's
    final record is bit-identical whatever shares its wave -- another pool size, a chain of launches
    against one launch -- on all three cell kinds; and it still matches the oracle."""
    for name, n in (("crustpinch", 30000), ("lopnor", 20000), ("sphere_deep", 3000)):
        m = models(name)
        big = Engine(m, reproducible=True)
        small = Engine(m, reproducible=True, pool_slots=768)
        assert big.pool_slots > small.pool_slots == 768
        ra, fa = big.run(n, first_id=5, seed=3, trace=True)
        rb, fb = small.run(n, first_id=5, seed=3, trace=True)
        assert finals_bytes(fa) == finals_bytes(fb), name
        assert (ra.counts == rb.counts).all() and ra.events == rb.events
        # chained against self-contained: the production kernels; energies are sums of identical terms in
        # another order, the integer outputs must be identical
        total = DeviceResult(m, "cuda:0")
        for k in range(3):
            small.run_device(n // 3, 5 + k * (n // 3), 3, *total.pointers(), carry="carry")
        small.run_device(n - 3 * (n // 3), 5 + 3 * (n // 3), 3, *total.pointers(), carry="final")
        torch.cuda.synchronize()
        rc = total.to_result()
        assert (rc.counts == ra.counts).all() and rc.events == ra.events
        # (the production kernels against the diagnostic one: another compilation of the same code, so a
        #  component that is tiny against its bin's energy -- particle motion all but normal to that
        #  axis, its projection a difference of nearly equal products -- moves in its leading digits;
        #  held to 1e-12 of the bin's energy by type, and the sums to the order of the atomics)
        scale = ra.energy[:, :, 3:].sum(-1, keepdims=True)
        assert np.all(np.abs(rc.energy - ra.energy) <= 1e-12 * scale + 1e-300)
        check_against_oracle(big, min(n, 5000), first_id=5, seed=3)
        big.close(), small.close()


def _bench_child(extra_args, launched, port=None):
    """bench.py as a fresh child process; `launched` gives it the launcher's environment of a
    one-rank job, i.e. the path the driver's N > 1 runs take: an nccl (= RCCL) process group, the
    result block all-reduced, barrier, destroy."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    if launched:
        env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "1"] + extra_args, env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("config,volume_reduce", [("crustpinch", None), ("crustpinch_volume", "dense"),
                                                  ("crustpinch_volume", "sparse"), ("crustpinch_volume", "allreduce")])
def test_bench_under_a_launcher_runs_rccl_and_equals_the_direct_run(config, volume_reduce):
    """The launched path of bench.py at world size 1: RCCL initialised, DeviceResult.allreduce_, barrier,
    destroy -- same totals as the direct run.  For config 5 the 10 GB event grid goes through the group
    in each of its forms: by frame with one RCCL reduce per owner and wave type ("dense": the MAX pre-pass
    and the chunked SUM on the int32 storage -- a job of one rank takes the same path through the
    library), by frame as compacted pairs ("sparse": r3d_volume_compact on the full grid), and all-reduced.
    (Reference semantics: scripts/do-parallel.sh:23-29, vis/seisplot/combine.m:26-33,
    vis/scattervid/scattervid_above.m:111.)"""
    from radiative3d_amd.launch import free_port
    args = ["--config", config, "--steps", "2", "--warmup", "1", "--histories", "200000", "--toa-degree", "5",
            "--no-cpu-baseline"] + (["--volume-reduce", volume_reduce] if volume_reduce else [])
    direct = _bench_child(args, launched=False)
    rccl = _bench_child(args, launched=True, port=free_port())
    assert direct["n_gpus"] == rccl["n_gpus"] == 1
    # the bins went through the library's own communicator (the code `./main --devices` reduces with), which says what
    # RCCL itself reports about the job
    c = rccl["collective"]
    assert c["control_plane"] == "nccl" and c["process_group"] and c["r3d_comm_error"] is None
    assert c["bins"].startswith("r3d_comm_reduce") and c["rccl_ranks"] == 1 and c["rccl_version"] > 20000
    assert [r["rank"] for r in c["ranks"]] == [0] and len(c["ranks"][0]["device_uuid"]) == 32
    assert direct["collective"]["bins"] is None and direct["collective"]["control_plane"] is None
    assert rccl["roofline"]["events_per_history"] == direct["roofline"]["events_per_history"]
    assert rccl["value"] > 0 and len(rccl["collective"]["per_rank_kernel_ms"]) == 1
    if config == "crustpinch_volume":
        v = rccl["volume"]
        assert v["events_binned"] == direct["volume"]["events_binned"] > 0
        assert v["reduced_as"] == {"dense": "dense int32", "sparse": "sparse pairs", "allreduce": "int32 in place"}[volume_reduce]
        assert direct["volume"]["reduced_as"] is None            # (no process group: nothing to reduce over)
        assert v["reduce_over_ranks_s"] > 0 and v["saturated_cells"] == 0
        assert rccl["collective"]["per_rank_kernel_ms"][0]["volume_reduce_s"] > 0
        if volume_reduce != "allreduce":
            assert v["frames_held_by_rank_0"] == [0, 300]        # world 1: rank 0 owns every frame
        if volume_reduce == "dense":
            assert v["phases_rank_0"]["bytes_sent"] == 0          # (N - 1) / N of the grid = nothing at N = 1


def test_bench_cpu_baseline_and_envelope_under_a_launcher():
    """At any world size the line carries cpu_baseline and the envelope figure (rank 0 times the
    oracle on host-built tables; the GPU batches run sharded and all-reduced over the group)."""
    from radiative3d_amd.launch import free_port
    line = _bench_child(["--config", "halfspace", "--steps", "2", "--warmup", "1", "--histories", "500000",
                         "--toa-degree", "5"], launched=True, port=free_port())
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["kind"] == "port"
    assert line["envelope"]["gpu_side"].startswith("device-built tables, each batch sharded over 1 rank")


def test_a_model_with_more_layers_keeps_its_cells_in_lds_beside_a_smaller_accumulator_table(monkeypatch):
    """The LDS carve-up for layered / spherical models whose cell records do not fit beside 256 bin accumulators
    (r3d_engine.hip): 128 accumulators, cells still staged (variant residency 0).  Reached on LopNor by reserving
    LDS as if its tables were 2.5 KB larger; the bins must not notice."""
    m = Model(lopnor(3))
    base = Engine(m)
    assert base.variant == (0, 0) and base.accumulators == 256
    e = Engine(m, lds_reserve=2500)
    assert e.variant == (0, 0) and e.accumulators == 128
    check_production_against_oracle(e, 20000)
    # and further down: with 7000 bytes gone neither the cell records nor the scatterer heads fit -- both from L2
    e2 = Engine(m, lds_reserve=7000)
    assert e2.variant == (0, 2) and e2.accumulators == 256
    check_production_against_oracle(e2, 5000)
