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
    assert int(vol.sum()) == res.events["scatter"] + res.events["reflect"] and res.events["volume_out"] == 0
    # 7.2 SCT + 2.8 REF ~ 10 grid increments per history with these overrides
    assert vol.sum() / n == pytest.approx(10.0, rel=0.15)
    # a grid that does not hold everything: what falls outside is counted, so the books close exactly
    res, vol = O.run_with_volume(m, n, volume_desc(**GRID))
    assert res.events["volume_out"] > 0
    assert int(vol.sum()) == res.events["scatter"] + res.events["reflect"] - res.events["volume_out"]
    assert O.run(m, n).events["volume_out"] == 0          # no grid attached: nothing to fall outside of


def test_kernel_code_fills_the_same_histogram_as_the_oracle(models):
    m = models("crustpinch", 4, VIDEO)
    v = volume_desc(**GRID)
    ro, vo = O.run_with_volume(m, 2500, v)
    re, ve = E.run_with_volume(m, 2500, v)
    assert ro.events == re.events and re.events["volume_out"] > 0
    assert int(ve.sum()) == re.events["scatter"] + re.events["reflect"] - re.events["volume_out"]
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
    assert int(vg.sum()) == rg.events["scatter"] + rg.events["reflect"] - rg.events["volume_out"] and rg.events["volume_out"] > 0
    e.run(n, first_id=n)                                  # a second shard accumulates on top
    _, vo2 = O.run_with_volume(m, n, v, first_id=n)
    assert (e.read_volume(reset=True) == vo + vo2).all()
    assert e.read_volume().sum() == 0


@pytest.mark.gpu
def test_full_size_crustpinch_video_run_with_the_10_gb_grid():
    """BASELINE config 5 at full size: do-crustpinch-vids.sh's arguments at TOA degree 9 (reference
    do-crustpinch-vids.sh:22-72), the bench's dense grid (2 x 300 x 64 x 256 x 256 uint32 = 10 GB,
    radiative3d_amd/configs.py CRUSTPINCH_VOLUME), 1e7 histories through the production kernel: every history
    ends, the receivers' books close, and the GRID's books close exactly -- its total is the SCT + REF events
    minus the ones counted as falling outside it (r3d_result.events[R3D_EV_VOLUME_OUT]).  Then, on the same
    degree-9 engine and the same grid, 3 000 histories against the oracle: the grid cell for cell, the bins, the
    counters, and the histories one by one (diagnostic and production kernels)."""
    from radiative3d_amd import Engine, Model
    from radiative3d_amd.configs import CRUSTPINCH_VOLUME, crustpinch_vids
    from test_gpu_parity import check_against_oracle, check_production_against_oracle
    m = Model(crustpinch_vids(9))
    assert m.n_toa == 20 * 4 ** 9
    e = Engine(m)
    e.set_volume(**CRUSTPINCH_VOLUME)
    n = 10_000_000
    r = e.run(n)
    assert r.n_lost + r.n_timeout + r.n_invalid == n and r.events["generated"] == n and r.n_invalid == 0
    assert int(r.counts.sum()) == r.events["catch"]
    assert np.allclose(r.energy[:, :, :3].sum(-1), r.energy[:, :, 3:].sum(-1), rtol=1e-10, atol=1e-300)
    vol = e.read_volume(reset=True)
    assert vol.nbytes == 2 * 300 * 64 * 256 * 256 * 4 > 10e9
    total = int(vol.sum(dtype=np.uint64))
    assert total == r.events["scatter"] + r.events["reflect"] - r.events["volume_out"], (total, r.events)
    assert r.events["scatter"] / n > 5 and total > 0.9 * (r.events["scatter"] + r.events["reflect"])
    del vol
    # 3 000 histories against the oracle, grid included (the oracle's 10 GB array is touched in ~30 000 places)
    k, first = 3000, 777_000_000
    rg = e.run(k, first_id=first)
    ro, vo = O.run_with_volume(m, k, volume_desc(**CRUSTPINCH_VOLUME), first_id=first)
    assert rg.events == ro.events and (rg.counts == ro.counts).all()
    vg = e.read_volume(reset=True).reshape(-1)
    vo = vo.reshape(-1)
    ig, io = np.flatnonzero(vg), np.flatnonzero(vo)
    assert ig.size == io.size > 15000 and (ig == io).all() and (vg[ig] == vo[io]).all()
    assert int(vg[ig].sum()) == rg.events["scatter"] + rg.events["reflect"] - rg.events["volume_out"]
    del vg, vo
    e.detach_volume()        # (the per-history comparisons below run the oracle without a grid)
    check_against_oracle(e, 3000, first_id=first, allow_frac=0.0005)
    check_production_against_oracle(e, 3000, first_id=first)
    e.close()


@pytest.mark.gpu
def test_pairs_of_a_grid_and_their_add_rebuild_the_grid(models):
    """r3d_volume_compact / r3d_volume_scatter_add (the grid between ranks, include/r3d.h) on an engine-
    filled grid, against the counters themselves: the pairs are exactly the non-zero cells (ragged
    ranges, ranges that do not start on a 16-byte boundary, a buffer too small), and adding them into
    an empty grid -- twice -- gives the grid and its double; the ceiling holds."""
    import ctypes as C

    import torch
    from radiative3d_amd import Engine, _ffi
    from radiative3d_amd.parallel import DeviceVolume
    m = models("crustpinch", 4, VIDEO)
    e = Engine(m)
    vol = DeviceVolume(e, device="cuda:0", **GRID)
    e.run(30000)
    torch.cuda.synchronize()
    lib = _ffi.hip_lib()
    grid = vol.counters
    host = grid.cpu().numpy().view(np.uint32)
    n_cells = host.size
    cap = n_cells // 2
    pairs = torch.empty((cap, 2), dtype=torch.int32, device="cuda:0")
    n_dev = torch.zeros(1, dtype=torch.int64, device="cuda:0")

    def compact(b, e_, capacity=cap):
        n_dev.zero_()
        assert lib.r3d_volume_compact(0, grid.data_ptr(), b, e_, pairs.data_ptr(), capacity, n_dev.data_ptr(), None) == 0, \
            lib.r3d_last_error()
        torch.cuda.synchronize()
        n = int(n_dev.item())
        p = pairs[:min(n, capacity)].cpu().numpy().view(np.uint32)
        return n, p[np.argsort(p[:, 0])]

    for b, e_ in ((0, n_cells), (4096 * 3 + 1, n_cells - 7), (5, 9), (12, 12), (n_cells - 4099, n_cells)):
        n, p = compact(b, e_)
        want = np.flatnonzero(host[b:e_]) + b
        assert n == want.size and (p[:, 0] == want).all() and (p[:, 1] == host[want]).all(), (b, e_)
    assert np.count_nonzero(host) > 20000
    n_small, p_small = compact(0, n_cells, capacity=1000)            # too small a buffer: counted, not written past it
    assert n_small == np.count_nonzero(host) and p_small.shape[0] == 1000 and (host[p_small[:, 0]] == p_small[:, 1]).all()
    n, _ = compact(0, n_cells)
    rebuilt = torch.zeros_like(grid)
    flags = torch.zeros(2, dtype=torch.int64, device="cuda:0")
    for _ in range(2):
        assert lib.r3d_volume_scatter_add(0, rebuilt.data_ptr(), n_cells, pairs.data_ptr(), n, flags.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert (rebuilt.cpu().numpy().view(np.uint32) == 2 * host).all() and flags.tolist() == [0, 0]
    # the ceiling: 2^32 - 3 in a cell, then the pairs on top -> pinned, counted once; an index beyond the grid is refused
    hot = int(np.flatnonzero(host)[0])
    rebuilt.zero_()
    rebuilt[hot] = -3
    bad = torch.tensor([[n_cells, 1]], dtype=torch.int32, device="cuda:0")
    assert lib.r3d_volume_scatter_add(0, rebuilt.data_ptr(), n_cells, pairs.data_ptr(), n, flags.data_ptr(), None) == 0
    assert lib.r3d_volume_scatter_add(0, rebuilt.data_ptr(), n_cells, bad.data_ptr(), 1, flags.data_ptr(), None) == 0
    torch.cuda.synchronize()
    got = rebuilt.cpu().numpy().view(np.uint32)
    want = host.copy()
    want[hot] = 0xFFFFFFFF if host[hot] >= 3 else host[hot] + 0xFFFFFFFD
    assert (got == want).all() and flags.tolist() == [1 if host[hot] >= 3 else 0, 1]
    assert lib.r3d_volume_compact(0, grid.data_ptr(), 8, 4, pairs.data_ptr(), cap, n_dev.data_ptr(), None) != 0
    vol.detach()
    e.close()
    del C


@pytest.mark.gpu
def test_grid_by_frame_between_two_shards_on_one_gpu(models):
    """The sparse reduction by frame (parallel.DeviceVolume.reduce_scatter_frames_) with the HIP kernels doing
    the work and the exchange done by hand: two shards of one job fill two grids on the one GPU of the box (the
    box has no second one for a process group), each compacts the other's frames (r3d_volume_compact) and adds
    what it is handed (r3d_volume_scatter_add) -- every owner's frames must equal the oracle's grid of the whole
    job.  (World sizes 2 and 3 through a real process group: tests/test_multi_rank_gloo.py, host tensors.)"""
    import torch
    from radiative3d_amd import Engine
    from radiative3d_amd.parallel import DeviceVolume, shard_range
    m = models("crustpinch", 4, VIDEO)
    n, world = 24000, 3
    _, want = O.run_with_volume(m, n, volume_desc(**GRID))
    vols = []
    for r in range(world):
        e = Engine(m)
        v = DeviceVolume(e, device="cuda:0", **GRID)
        lo, hi = shard_range(n, r, world)
        e.run(hi - lo, first_id=lo)
        torch.cuda.synchronize()
        v.detach()
        e.close()
        vols.append(v)
    cap = vols[0].counters.numel() // 4
    sent = []
    for r, v in enumerate(vols):                       # every rank's pairs, owner by owner
        pairs = torch.empty((cap, 2), dtype=torch.int32, device="cuda:0")
        n_dev = torch.zeros(1, dtype=torch.int64, device="cuda:0")
        ends = []
        for owner in range(world):
            if owner != r:
                for b, e_ in v._segments(*v.frame_range(owner, world)):
                    v._compact(b, e_, pairs, n_dev, cap)
            ends.append(int(n_dev.item()))
        assert ends[-1] <= cap
        sent.append((pairs, ends))
    for owner, v in enumerate(vols):                   # the hand-over, then the owner's add
        for r, (pairs, ends) in enumerate(sent):
            if r != owner:
                v._scatter_add(pairs[(ends[owner - 1] if owner else 0):ends[owner]])
        v.owned = v.frame_range(owner, world)
        lo, hi, got = v.frames_numpy()
        assert (lo, hi) == v.frame_range(owner, world) and (got == want[:, lo:hi]).all(), owner
        assert v.saturated == 0
    assert sum(v.total() for v in vols) == int(want.sum(dtype=np.uint64))


@pytest.mark.gpu
def test_grids_of_one_process_reduced_by_frame_through_the_c_abi(models):
    """r3d_volume_reduce_by_frame (include/r3d.h): what a host that drives its GPUs from one process calls after
    r3d_run_model_on-style shards -- three engines on the box's one GPU, each with its own grid; afterwards
    engine g holds the oracle's counts of the WHOLE job for its frames (35 frames over 3 owners: 12 + 12 + 11).
    Reference semantics: vis/seisplot/combine.m:26-33 (replicas add), vis/scattervid/scattervid_above.m:111."""
    from radiative3d_amd import Engine
    from radiative3d_amd.model import reduce_volumes_by_frame
    from radiative3d_amd.parallel import shard_range
    m = models("crustpinch", 4, VIDEO)
    n, world = 24000, 3
    _, want = O.run_with_volume(m, n, volume_desc(**GRID))
    engines = []
    for r in range(world):
        e = Engine(m)
        e.set_volume(**GRID)
        lo, hi = shard_range(n, r, world)
        e.run(hi - lo, first_id=lo)
        engines.append(e)
    own = [e.read_volume() for e in engines]
    frames, sat = reduce_volumes_by_frame(engines)
    assert frames == [0, 12, 24, 35] and sat == 0
    for g, e in enumerate(engines):
        got = e.read_volume()
        lo, hi = frames[g], frames[g + 1]
        assert (got[:, lo:hi] == want[:, lo:hi]).all(), g
        rest = np.ones(35, dtype=bool)
        rest[lo:hi] = False
        assert (got[:, rest] == own[g][:, rest]).all()            # the other frames keep the engine's own counts
    # one engine: nothing to add, the frames are all its own
    assert reduce_volumes_by_frame(engines[:1]) == ([0, 35], 0)
    # refusals: an engine without a grid, grids of different shapes, one grid twice
    bare = Engine(m)
    with pytest.raises(RuntimeError, match="without a grid"):
        reduce_volumes_by_frame([engines[0], bare])
    other = dict(GRID, n_frames=7)
    bare.set_volume(**other)
    with pytest.raises(RuntimeError, match="differ in shape"):
        reduce_volumes_by_frame([engines[0], bare])
    with pytest.raises(RuntimeError, match="share one grid"):
        reduce_volumes_by_frame([engines[0], engines[0]])
    for e in engines + [bare]:
        e.close()


@pytest.mark.gpu
def test_a_grid_too_full_for_its_pairs_fails_before_anything_is_added(models):
    """r3d_volume_reduce_by_frame counts first and modifies second: when one engine's non-zero cells in the other owners'
    frames exceed its pair buffer (a sixteenth of the grid), the call fails with EVERY grid as it was -- including the
    owners that an earlier, fitting source would already have added its pairs to (the round-4 form did add them)."""
    import torch
    from radiative3d_amd import Engine
    from radiative3d_amd.model import reduce_volumes_by_frame
    m = models("crustpinch", 4, VIDEO)
    small = dict(origin=(-200.0, -600.0, -130.0), cell_size=(80.0, 80.0, 35.0), dims=(16, 15, 4), n_frames=12, frame_dt=30.0)
    cells = 2 * 12 * 4 * 15 * 16
    engines, bufs = [], []
    for r in range(3):
        e = Engine(m)
        buf = torch.zeros(cells, dtype=torch.int32, device="cuda:0")
        e.set_volume_buffer(counters=buf, **small)
        engines.append(e), bufs.append(buf)
    bufs[0][::97] = 3                    # source 0: a few pairs, they fit
    bufs[1][:] = 1                       # source 1: every cell -- two thirds of the grid against room for a sixteenth
    bufs[2][::29] = 2
    before = [b.clone() for b in bufs]
    with pytest.raises(RuntimeError, match="engine 1 is too full for the pair buffer.*no grid has been modified"):
        reduce_volumes_by_frame(engines)
    for b, want in zip(bufs, before):
        assert torch.equal(b, want)
    bufs[1].zero_()
    bufs[1][1::89] = 7                   # now everything fits: the reduction goes through and adds up
    want = (before[0].long() + bufs[1].long() + before[2].long())
    frames, sat = reduce_volumes_by_frame(engines)
    fc = 4 * 15 * 16
    for g in range(3):
        for t in range(2):
            lo, hi = (t * 12 + frames[g]) * fc, (t * 12 + frames[g + 1]) * fc
            assert torch.equal(bufs[g][lo:hi].long(), want[lo:hi]), (g, t)
    assert sat == 0
    for e in engines:
        e.detach_volume()
        e.close()
