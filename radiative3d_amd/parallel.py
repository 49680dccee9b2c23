"""Sharding of a history range over ranks and the one reduction at the end.

Histories are independent and keyed by id (Philox counter = id), the model is
read-only and every output is a sum, so the job shards with no data-path
exchange: rank r runs a contiguous id range, and the per-receiver bins and
counters are summed once at the end -- the same semantics as the reference's
process-level replicas + `combine` (scripts/do-parallel.sh:23-29,
vis/seisplot/combine.m:26-33).  The reduction is one all-reduce(SUM) per
buffer through torch.distributed (backend "nccl" = RCCL over xGMI on the GPU
node, "gloo" in CPU tests).
"""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous, balanced [lo, hi) of range(n) for `rank` of `world`."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def allreduce_result_(energy, counts, scalars):
    """In-place SUM over ranks of the result block.

    energy: float64 tensor, counts/scalars: int64 tensors (uint64 values are
    below 2^63 by construction).  No-op when no process group is initialised."""
    if dist.is_available() and dist.is_initialized():
        for t in (energy, counts, scalars):
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return energy, counts, scalars


class DeviceResult:
    """Result block resident in HBM (torch tensors are only the allocator and
    the handle RCCL reduces; the engine writes through raw pointers)."""

    def __init__(self, model, device):
        from . import _ffi
        n = model.n_seismometers * model.n_bins
        self.model = model
        self.energy = torch.zeros(max(1, n) * _ffi.R3D_N_ENERGY, dtype=torch.float64, device=device)
        # the integer outputs share one buffer, so the reduction is two collectives, not three
        n_counts = max(1, n) * _ffi.R3D_N_COUNT
        self._ints = torch.zeros(n_counts + _ffi.R3D_N_SCALARS, dtype=torch.int64, device=device)
        self.counts = self._ints[:n_counts]
        self.scalars = self._ints[n_counts:]

    def zero_(self):
        self.energy.zero_(), self._ints.zero_()

    def add_(self, other):
        self.energy.add_(other.energy), self._ints.add_(other._ints)
        return self

    def pointers(self):
        return self.energy.data_ptr(), self.counts.data_ptr(), self.scalars.data_ptr()

    def allreduce_(self):
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(self.energy, op=dist.ReduceOp.SUM)
            dist.all_reduce(self._ints, op=dist.ReduceOp.SUM)
        return self

    def to_result(self):
        import numpy as np
        res = self.model.new_result()
        m = self.model
        res.energy[...] = self.energy.cpu().numpy()[:res.energy.size].reshape(res.energy.shape)
        res.counts[...] = self.counts.cpu().numpy()[:res.counts.size].astype(np.uint64).reshape(res.counts.shape)
        res.set_scalars(self.scalars.cpu().numpy())
        return res
