"""The N>1 path on CPU: two ranks over gloo shard an id range, each computes
its shard, one all-reduce sums the result blocks -- the same code path
bench.py takes over RCCL.  The per-rank compute is stood in for by the
test-only host build of the kernel code (there is no product CPU path)."""
import os
import socket

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
