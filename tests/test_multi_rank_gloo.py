"""The N>1 path on CPU: two ranks over gloo shard an id range, each computes
its shard, one all-reduce sums the result blocks -- the same code path
bench.py takes over RCCL.  The per-rank compute is stood in for by the
test-only host build of the kernel code (there is no product CPU path)."""
import os
import socket

import pytest
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import emul_ffi as E
from radiative3d_amd import Model, _ffi
from radiative3d_amd.parallel import DeviceResult, allreduce_result_, shard_range
from tests.configs import lopnor


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = Model(lopnor(3))
    lo, hi = shard_range(n, rank, world)
    res = E.run(model, hi - lo, first_id=lo)
    energy = torch.from_numpy(res.energy.reshape(-1).copy())
    counts = torch.from_numpy(res.counts.astype(np.int64).reshape(-1))
    scalars = torch.from_numpy(res.scalars().astype(np.int64))
    dist.barrier()
    # bench.py's object: result block in (here: host) tensors, per-step buffer reduced and added
    total, step = DeviceResult(model, "cpu"), DeviceResult(model, "cpu")
    step.energy.copy_(energy), step.counts.copy_(counts), step.scalars.copy_(scalars)
    total.add_(step.allreduce_())
    allreduce_result_(energy, counts, scalars)
    assert torch.equal(total.counts, counts) and torch.equal(total.scalars, scalars)
    assert torch.allclose(total.energy, energy, rtol=1e-14, atol=0)
    if rank == 0:
        back = total.to_result()
        assert back.n_lost + back.n_timeout + back.n_invalid == n
        np.savez(out_path, energy=energy.numpy(), counts=counts.numpy(), scalars=scalars.numpy())
    dist.destroy_process_group()


def _reduce_once_worker(rank, world, port, n, steps, out_path):
    """bench.py's two reduction schedules on the same shards: every launch adds into the rank's own
    block and the blocks are all-reduced ONCE at the end (the default; the reference's replicas +
    combine, vis/seisplot/combine.m:26-33) against one all-reduce per step."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = Model(lopnor(3))
    once, per_step, step = DeviceResult(model, "cpu"), DeviceResult(model, "cpu"), DeviceResult(model, "cpu")
    for i in range(steps):
        res = E.run(model, n, first_id=(i * world + rank) * n)        # bench.py's id layout
        step.energy.copy_(torch.from_numpy(res.energy.reshape(-1).copy()))
        step.counts.copy_(torch.from_numpy(res.counts.astype(np.int64).reshape(-1)))
        step.scalars.copy_(torch.from_numpy(res.scalars().astype(np.int64)))
        once.add_(step)                       # what r3d_run_device does in place on the GPU
        per_step.add_(step.allreduce_())
    once.allreduce_()
    assert torch.equal(once._ints, per_step._ints)
    assert torch.allclose(once.energy, per_step.energy, rtol=1e-13, atol=0)
    if rank == 0:
        r = once.to_result()
        np.savez(out_path, counts=r.counts, scalars=r.scalars(), energy=r.energy)
    dist.destroy_process_group()


def test_one_reduction_at_the_end_equals_one_per_step(tmp_path):
    n, steps, world = 1500, 3, 2
    out = str(tmp_path / "once.npz")
    mp.spawn(_reduce_once_worker, args=(world, _free_port(), n, steps, out), nprocs=world, join=True)
    got = np.load(out)
    want = E.run(Model(lopnor(3)), n * steps * world)    # the ids of all steps and ranks are one contiguous range
    assert (got["counts"] == want.counts).all() and (got["scalars"] == want.scalars()).all()
    assert np.allclose(got["energy"], want.energy, rtol=1e-12, atol=1e-300)


def test_shard_range_partitions():
    for n in (0, 1, 7, 64, 1000003):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_ranks_sum_to_the_single_process_result(tmp_path):
    n = 6001
    out = str(tmp_path / "r.npz")
    mp.spawn(_worker, args=(2, _free_port(), n, out), nprocs=2, join=True)
    got = np.load(out)
    model = Model(lopnor(3))
    want = E.run(model, n)
    assert (got["counts"].reshape(want.counts.shape) == want.counts.astype(np.int64)).all()
    assert (got["scalars"] == want.scalars().astype(np.int64)).all()
    assert np.allclose(got["energy"].reshape(want.energy.shape), want.energy, rtol=1e-12, atol=1e-300)
    assert int(got["scalars"][3 + _ffi.R3D_INV_NUM]) == n      # events[generated]


# ---- the volumetric scatter-event grid over ranks (BASELINE config 5) ----
VIDEO = ["--overridemfp=25,50", "--nodeflect", "--timetolive=350"]
GRID = dict(origin=(-200.0, -600.0, -130.0), cell_size=(20.0, 20.0, 10.0), dims=(64, 60, 14),
            n_frames=35, frame_dt=10.0)


def _volume_worker(rank, world, port, n, out_path):
    from radiative3d_amd.model import volume_desc
    from radiative3d_amd.parallel import DeviceVolume
    from tests.configs import crustpinch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = Model(crustpinch(3) + VIDEO)
    lo, hi = shard_range(n, rank, world)
    _, vol = E.run_with_volume(model, hi - lo, volume_desc(**GRID), first_id=lo)
    dv = DeviceVolume(None, device="cpu", **GRID)          # engine-less: the test fills the counters itself
    dv.counters.copy_(torch.from_numpy(vol.reshape(-1).view(np.int32)))
    mine = dv.total()
    # chunks smaller than the grid, not dividing it: the chunk loop is what is being tested
    dv.allreduce_(chunk_elems=100_003)
    # (no cell can reach 2^31 here: the int32 storage itself went over the wire, 4 bytes per cell)
    assert dv.widened is False and dv.saturated == 0 and dv.total() >= mine
    # reduce_: the destination holds the sum afterwards
    dr = DeviceVolume(None, device="cpu", **GRID)
    dr.counters.copy_(torch.from_numpy(vol.reshape(-1).view(np.int32)))
    dr.reduce_(dst=0, chunk_elems=250_000)
    if rank == 0:
        assert torch.equal(dr.counters, dv.counters)
        np.save(out_path, dv.to_numpy())
    # the widened path on the same data (forced by one counter at 2^31 - 1 on one rank): same sums elsewhere
    dw = DeviceVolume(None, device="cpu", **GRID)
    dw.counters.copy_(torch.from_numpy(vol.reshape(-1).view(np.int32)))
    keep = int(dw.counters[0].item())
    if rank == 1:
        dw.counters[0] = 0x7FFFFFFF
    dw.allreduce_(chunk_elems=100_003)
    assert dw.widened is True and dw.saturated == 0
    assert torch.equal(dw.counters[1:], dv.counters[1:])
    assert int(dw.to_numpy().reshape(-1)[0]) == (0x7FFFFFFF + keep if rank == 1 else 0x7FFFFFFF + int(dv.counters[0]) - keep)
    # the headroom test is about the SUM: 2^30 on both ranks must take the widened path too
    dh = DeviceVolume(None, device="cpu", origin=(0, 0, 0), cell_size=(1, 1, 1), dims=(2, 1, 1), n_frames=1,
                      frame_dt=1.0)
    dh.counters.copy_(torch.tensor([1 << 30, 1, 2, 3], dtype=torch.int32))
    dh.allreduce_()
    assert dh.widened is True and dh.to_numpy().reshape(-1).tolist() == [1 << 31, 2, 4, 6]
    # saturation instead of wrap-around: two ranks each holding 2^32 - 5 in one cell, 7 in another
    ds = DeviceVolume(None, device="cpu", origin=(0, 0, 0), cell_size=(1, 1, 1), dims=(2, 1, 1), n_frames=1,
                      frame_dt=1.0)
    ds.counters.copy_(torch.from_numpy(np.array([0xFFFFFFFB, 7, 0, 3], dtype=np.uint32).view(np.int32)))
    ds.allreduce_()
    assert ds.to_numpy().reshape(-1).tolist() == [0xFFFFFFFF, 14, 0, 6] and ds.saturated == 1 and ds.widened is True
    dist.destroy_process_group()


def test_two_ranks_volume_grid_sums_to_the_single_process_histogram(tmp_path):
    from radiative3d_amd.model import volume_desc
    from tests.configs import crustpinch
    n = 3001
    out = str(tmp_path / "vol.npy")
    mp.spawn(_volume_worker, args=(2, _free_port(), n, out), nprocs=2, join=True)
    got = np.load(out)
    model = Model(crustpinch(3) + VIDEO)
    _, want = E.run_with_volume(model, n, volume_desc(**GRID))
    assert got.dtype == np.uint32 and got.shape == want.shape
    assert int(want.sum()) > 10000 and (got == want).all()


# ---- the job's reduction of the grid: by frame (SURVEY.md 8(e) "keep sharded by frame") -------------
def _frames_worker(rank, world, port, n, out_dir):
    """reduce_scatter_frames_ in its three forms against the all-reduced grid, on real engine output
    (the host emulation's grid of this rank's id shard).  World sizes 2 and 3: 35 frames do not divide by 3."""
    from radiative3d_amd.model import volume_desc
    from radiative3d_amd.parallel import DeviceVolume
    from tests.configs import crustpinch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = Model(crustpinch(3) + VIDEO)
    lo, hi = shard_range(n, rank, world)
    _, vol = E.run_with_volume(model, hi - lo, volume_desc(**GRID), first_id=lo)
    own = torch.from_numpy(vol.reshape(-1).view(np.int32)).clone()

    def fresh():
        d = DeviceVolume(None, device="cpu", **GRID)
        d.counters.copy_(own)
        return d

    whole = fresh().allreduce_()                       # the reference point: every cell's job total
    want = whole.to_numpy()
    f_lo, f_hi = whole.frame_range(rank, world)
    assert sum(whole.frame_range(r, world)[1] - whole.frame_range(r, world)[0] for r in range(world)) == 35
    for mode, kwargs in (("sparse", {}), ("dense", dict(chunk_elems=100_003)), ("auto", {}),
                         ("auto", dict(pair_capacity=16))):   # (16 pairs cannot hold a rank's cells: auto goes dense)
        d = fresh().reduce_scatter_frames_(mode=mode, **kwargs)
        assert d.owned == (f_lo, f_hi)
        a, b, mine = d.frames_numpy()
        assert (a, b) == (f_lo, f_hi) and (mine == want[:, f_lo:f_hi]).all(), (mode, kwargs)
        said = d.timing["mode"]
        assert said == ("dense int32" if mode == "dense" or kwargs.get("pair_capacity") else "sparse pairs"), said
        if said == "sparse pairs":       # what travelled: this rank's non-zero cells in the OTHER ranks' frames
            others = own.view(d.shape).clone()
            others[:, f_lo:f_hi] = 0
            assert d.timing["pairs_sent"] == int((others != 0).sum()) and d.timing["bytes_sent"] == 8 * d.timing["pairs_sent"]
        else:
            assert d.timing["bytes_sent"] == 4 * (d.counters.numel() - 2 * (f_hi - f_lo) * d.frame_cells)
        assert d.total() == int(want[:, f_lo:f_hi].sum()) and d.job_total() == int(want.sum())
        with pytest.raises(RuntimeError, match="only frames"):
            d.to_numpy()                               # the other frames are stale: not readable as a whole grid
        d.allgather_frames_()
        assert d.owned is None and (d.to_numpy() == want).all()
    with pytest.raises(RuntimeError, match="refused"):       # asked for by name, the sparse form does not fall back
        fresh().reduce_scatter_frames_(mode="sparse", pair_capacity=16)
    # a reduce_ leaves the other ranks with scratch that refuses to be read
    r = fresh().reduce_(dst=world - 1)
    if rank == world - 1:
        assert (r.to_numpy() == want).all()
    else:
        with pytest.raises(RuntimeError, match="scratch"):
            r.total()
    # ... and allgather_frames_ after it is an error on EVERY rank (the destination holds a whole grid, the others
    # scratch: entering the broadcasts from some ranks only would hang the job), not a deadlock
    with pytest.raises(RuntimeError, match="needs every rank's grid reduced by frame"):
        r.allgather_frames_()
    assert fresh().allgather_frames_().owned is None          # nothing reduced by frame anywhere: a no-op, everywhere
    # saturation and the 2^31 switch-over, by frame: one frame per rank at world 2, cells chosen per owner
    tiny = dict(origin=(0, 0, 0), cell_size=(1, 1, 1), dims=(2, 1, 1), n_frames=world, frame_dt=1.0)
    for mode in ("sparse", "dense"):
        t = DeviceVolume(None, device="cpu", **tiny)
        vals = np.zeros((2, world, 2), dtype=np.uint32)
        vals[0, :, 0] = 0xFFFFFFFB          # every rank holds 2^32 - 5 in one cell of every frame: sums saturate
        vals[0, :, 1] = 1 << 30             # and 2^30 in another: the sum of two is 2^31, beyond int32
        vals[1, :, 0] = 7 + rank
        t.counters.copy_(torch.from_numpy(vals.reshape(-1).view(np.int32)))
        t.reduce_scatter_frames_(mode=mode)
        a, b, got = t.frames_numpy()
        assert b - a == 1
        assert got[0, 0].reshape(-1).tolist() == [0xFFFFFFFF, min(world << 30, 0xFFFFFFFF)], (mode, got)
        assert got[1, 0].reshape(-1).tolist() == [7 * world + sum(range(world)), 0]
        assert t.saturated == 1 and (t.widened is True if mode == "dense" else t.widened is None)
    if rank == 0:
        np.save(os.path.join(out_dir, f"frames_{world}.npy"), want)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_grid_reduced_by_frame_equals_the_all_reduced_grid(tmp_path, world):
    from radiative3d_amd.model import volume_desc
    from tests.configs import crustpinch
    n = 2400
    mp.spawn(_frames_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    got = np.load(str(tmp_path / f"frames_{world}.npy"))
    _, want = E.run_with_volume(Model(crustpinch(3) + VIDEO), n, volume_desc(**GRID))
    assert int(want.sum()) > 5000 and (got == want).all()


def _comm_form_worker(rank, world, port, out_path):
    """Comm.form without a GPU: the library's collective communicator cannot be formed here, and what matters is HOW
    that goes -- every rank gets the same RuntimeError (nobody is left waiting inside ncclCommInitRank for a rank that
    gave up), and the process group is usable afterwards (bench.py then reduces the bins through torch.distributed)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from radiative3d_amd.parallel import Comm
    said = ""
    try:
        Comm.form("cuda:%d" % rank)
    except RuntimeError as exc:
        said = str(exc)
    t = torch.tensor([1 if said else 0], dtype=torch.int64)
    dist.all_reduce(t)                     # (still in step with each other)
    if rank == 0:
        with open(out_path, "w") as f:
            f.write("%d\n%s\n" % (int(t.item()), said))
    dist.destroy_process_group()


def test_a_communicator_that_cannot_form_fails_on_every_rank_alike(tmp_path):
    world = 2
    out = tmp_path / "said.txt"
    mp.spawn(_comm_form_worker, args=(world, _free_port(), str(out)), nprocs=world, join=True)
    lines = out.read_text().splitlines()
    assert int(lines[0]) == world, lines            # every rank raised
    assert "librccl" in lines[1] or "r3d_comm" in lines[1], lines
