"""Volumetric scatter-event grid (BASELINE.json config 5):
count[type][frame][z][y][x] of SCT + REF events, the histogram the reference's
video pipeline builds from its per-event text stream (dataout.cpp:570-577,
vis/scattervid/preprocess.sh:17-29, scattervid_above.m:111).  Video runs use
--overridemfp=25,50 --nodeflect (do-crustpinch-vids.sh:48-50)."""
import numpy as np
import pytest

import emul_ffi as E
from oracle import oracle_ffi as O
from radiative3d_amd.model import volume_desc

VIDEO = ("--overridemfp=25,50", "--nodeflect", "--timetolive=350")
GRID = dict(origin=(-200.0, -600.0, -130.0), cell_size=(20.0, 20.0, 10.0), dims=(64, 60, 14),
            n_frames=35, frame_dt=10.0)


def test_oracle_histogram_accounts_for_every_event_in_range(models):
    m = models("crustpinch", 4, VIDEO)
    n = 3000
    big = volume_desc((-5000, -5000, -1000), (10000, 10000, 2000), (1, 1, 1), 1, 1e9)  # one all-embracing cell
    res, vol = O.run_with_volume(m, n, big)
    assert int(vol.sum()) == res.events["scatter"] + res.events["reflect"]
    # 7.2 SCT + 2.8 REF ~ 10 grid increments per history with these overrides
    assert vol.sum() / n == pytest.approx(10.0, rel=0.15)


def test_kernel_code_fills_the_same_histogram_as_the_oracle(models):
    m = models("crustpinch", 4, VIDEO)
    v = volume_desc(**GRID)
    ro, vo = O.run_with_volume(m, 2500, v)
    re, ve = E.run_with_volume(m, 2500, v)
    assert ro.events == re.events
    assert vo.sum() > 10000 and (vo == ve).all()
    assert vo[0].sum() > 0 and vo[1].sum() > 0            # both wave types present
    # the wavefront expands: the first frames occupy more and more cells
    occupied = [(vo[:, f] > 0).sum() for f in range(35)]
    assert occupied[0] < occupied[1] < occupied[2]


@pytest.mark.gpu
def test_engine_histogram_matches_oracle_and_accumulates(models):
    from radiative3d_amd import Engine
    m = models("crustpinch", 4, VIDEO)
    v = volume_desc(**GRID)
    e = Engine(m)
    e.set_volume(**GRID)
    n = 20000
    rg = e.run(n)
    ro, vo = O.run_with_volume(m, n, v)
    vg = e.read_volume()
    assert rg.events == ro.events
    assert (vg == vo).all()                               # integer work: bit-exact
    e.run(n, first_id=n)                                  # a second shard accumulates on top
    _, vo2 = O.run_with_volume(m, n, v, first_id=n)
    assert (e.read_volume(reset=True) == vo + vo2).all()
    assert e.read_volume().sum() == 0


@pytest.mark.gpu
def test_dense_volume_grid_at_scale(models):
    """Config 5 flavour: a dense grid (2 x 300 x 64 x 256 x 256 uint32 = 10 GB) under 1e7 histories."""
    from radiative3d_amd import Engine
    m = models("crustpinch", 6, VIDEO)
    e = Engine(m)
    e.set_volume(origin=(-300.0, -900.0, -400.0), cell_size=(6.0, 7.5, 6.5), dims=(256, 256, 64),
                 n_frames=300, frame_dt=350.0 / 300)
    n = 10_000_000
    r = e.run(n)
    vol = e.read_volume()
    assert r.n_lost + r.n_timeout + r.n_invalid == n
    total = int(vol.sum(dtype=np.uint64))
    assert 0.8 * (r.events["scatter"] + r.events["reflect"]) < total <= r.events["scatter"] + r.events["reflect"]
