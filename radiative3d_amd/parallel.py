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


U32_MAX = 0xFFFFFFFF


def _as_unsigned(chunk_i32):
    """int32 storage holding uint32 counters -> their values as int64."""
    return chunk_i32.to(torch.int64) & U32_MAX


def _to_storage(values_i64):
    """int64 values in [0, 2^32) -> the int32 bit pattern of the same uint32."""
    return torch.where(values_i64 > 0x7FFFFFFF, values_i64 - (1 << 32), values_i64).to(torch.int32)


class DeviceVolume:
    """The volumetric scatter-event grid count[type][frame][z][y][x] (uint32, include/r3d.h
    r3d_volume_desc) in caller-owned device memory, and its sum over ranks.

    The reference has no such grid: its video pipeline writes one text line per SCT / REF
    event (dataout.cpp:570-577) and the Octave scripts bin them per frame
    (vis/scattervid/scattervid_above.m:111); replicas are combined by adding
    (vis/seisplot/combine.m:26-33).  Here every rank's engine increments its own grid
    (r3d_engine_set_volume_buffer) and `allreduce_` / `reduce_` add the grids once at the end
    of the job.  The counters are uint32 (a 10 GB grid at the 300 x 64 x 256 x 256 size of
    BASELINE config 5), and a sum over ranks of 1e8..1e9 histories could pass 2^32 in a hot
    cell.  The reduction first asks (one scalar MAX over ranks) whether any cell CAN reach 2^31:
    if not -- the usual case -- the int32 storage is all-reduced as it is, 4 bytes per cell on
    the wire; else it runs chunk by chunk in int64 and SATURATES at 2^32 - 1 instead of
    wrapping; `saturated` counts the cells that hit the ceiling, `widened` says which path ran.
    (torch has no arithmetic on uint32, so the storage is int32 holding the same bits.)"""

    def __init__(self, engine, origin, cell_size, dims, n_frames, frame_dt, device):
        self.shape = (2, int(n_frames), int(dims[2]), int(dims[1]), int(dims[0]))
        n = 1
        for d in self.shape:
            n *= d
        self.counters = torch.zeros(n, dtype=torch.int32, device=device)
        self.saturated = 0
        self.widened = None
        self.engine = engine
        if engine is not None:   # (the engine keeps a reference to the tensor: model.Engine.set_volume_buffer)
            engine.set_volume_buffer(origin, cell_size, dims, n_frames, frame_dt, self.counters)

    def detach(self):
        """Stop the engine from adding into this grid (before the tensor is dropped or re-used)."""
        if self.engine is not None:
            self.engine.detach_volume()
            self.engine = None

    def zero_(self):
        self.counters.zero_()
        self.saturated = 0

    def _headroom(self, world):
        """True when no cell's sum over `world` ranks can reach 2^31: the counters can then be added
        as they are stored (int32), with no widening.  One scalar all-reduce (MAX) decides it for
        all ranks alike."""
        lo, hi = int(self.counters.min().item()), int(self.counters.max().item())
        worst = torch.tensor([(1 << 32) if lo < 0 else hi], dtype=torch.int64, device=self.counters.device)
        dist.all_reduce(worst, op=dist.ReduceOp.MAX)
        return int(worst.item()) * world < (1 << 31)

    def _reduce_chunks(self, collective, chunk_elems, keep):
        """Add the grids over ranks.  Usual case (every rank's largest counter x ranks < 2^31):
        the collective runs on the int32 storage itself, in place, 4 bytes per cell on the wire.
        Otherwise: widen, add, saturate, store back -- chunk by chunk so that the int64 scratch
        stays small next to a multi-GB grid (8 bytes per cell on the wire)."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return self
        if self._headroom(dist.get_world_size()):
            self.widened = False
            for lo in range(0, self.counters.numel(), chunk_elems):
                collective(self.counters[lo:lo + chunk_elems])
            return self
        self.widened = True
        sat = 0
        for lo in range(0, self.counters.numel(), chunk_elems):
            part = self.counters[lo:lo + chunk_elems]
            wide = _as_unsigned(part)
            collective(wide)
            if keep:
                sat += int((wide > U32_MAX).sum().item())
                part.copy_(_to_storage(wide.clamp_(max=U32_MAX)))
        self.saturated += sat
        return self

    def allreduce_(self, chunk_elems=1 << 28):
        """Every rank ends with the job's grid (one all-reduce(SUM) per chunk of 1 GiB)."""
        return self._reduce_chunks(lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM), chunk_elems, True)

    def reduce_(self, dst=0, chunk_elems=1 << 28):
        """Rank `dst` ends with the job's grid; the other ranks' buffers are then undefined
        (a reduce may use them as scratch)."""
        return self._reduce_chunks(lambda t: dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM), chunk_elems,
                                   dist.is_initialized() and dist.get_rank() == dst)

    def total(self, chunk_elems=1 << 26):
        """Sum of all counters (events binned), computed chunk by chunk."""
        t = 0
        for lo in range(0, self.counters.numel(), chunk_elems):
            t += int(_as_unsigned(self.counters[lo:lo + chunk_elems]).sum().item())
        return t

    def to_numpy(self):
        import numpy as np
        return self.counters.cpu().numpy().view(np.uint32).reshape(self.shape)
