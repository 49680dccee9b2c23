"""Sharding of a history range over ranks and the one reduction at the end.

Histories are independent and keyed by id (Philox counter = id), the model is
read-only and every output is a sum, so the job shards with no data-path
exchange: rank r runs a contiguous id range, and the per-receiver bins and
counters are summed once at the end -- the same semantics as the reference's
process-level replicas + `combine` (scripts/do-parallel.sh:23-29,
vis/seisplot/combine.m:26-33).  On GPUs the reduction is the PRODUCT's: the
ranks form an RCCL communicator through the engine library (`Comm`:
r3d_comm_create, include/r3d.h) and `DeviceResult.allreduce_` calls
r3d_comm_reduce -- the code r3d_node_run (one process, N shards: what
`./main --devices` runs) reduces with; torch.distributed only carries the
128-byte communicator id to the ranks and the bench's barriers.  Host tensors
(the gloo tests of the sharding) are summed by torch.distributed's all-reduce.
The event grid of BASELINE config 5 (10 GB per rank) is reduced BY FRAME and
sparsely instead (DeviceVolume.reduce_scatter_frames_, torch.distributed point to point).
"""
import ctypes as C

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


class Comm:
    """This rank's handle of an RCCL communicator made by the engine library (include/r3d.h r3d_comm_*).

    form(device): rank 0 of the torch.distributed group makes the id (r3d_comm_unique_id), the group broadcasts
    its 128 bytes, every rank calls r3d_comm_create with its own device.  A collective: every rank calls it."""

    def __init__(self, handle, lib):
        self._c, self._lib = handle, lib

    @classmethod
    def form(cls, device, lib=None):
        from . import _ffi
        L = _ffi.hip_lib(path=lib)
        rank, world = dist.get_rank(), dist.get_world_size()
        ident = C.create_string_buffer(_ffi.R3D_COMM_ID_BYTES)
        on = torch.device(device) if dist.get_backend() == "nccl" else torch.device("cpu")
        # Forming the communicator is a collective INSIDE the library: a rank that cannot take part (no librccl to bind)
        # must say so BEFORE the others enter it and wait for it for ever -- every rank probes the library first (an id
        # of its own, thrown away) and the ranks agree on the outcome.
        probe = C.create_string_buffer(_ffi.R3D_COMM_ID_BYTES)
        can = torch.tensor([0 if L.r3d_comm_unique_id(probe) else 1], dtype=torch.int32, device=on)
        dist.all_reduce(can, op=dist.ReduceOp.MIN)
        if not int(can.item()):
            raise RuntimeError("librccl cannot be bound on some rank: " + L.r3d_last_error().decode())
        status = 0
        if rank == 0 and L.r3d_comm_unique_id(ident):
            status = 1
        # (the id and rank 0's status travel together, so that every rank gives up when rank 0 could not make one)
        box = torch.tensor(list(ident.raw) + [status], dtype=torch.uint8, device=on)
        dist.broadcast(box, src=0)
        box = box.cpu()
        if int(box[-1]):
            raise RuntimeError("r3d_comm_unique_id failed on rank 0" + (": " + L.r3d_last_error().decode() if rank == 0 else ""))
        ident = C.create_string_buffer(bytes(box[:-1].tolist()), _ffi.R3D_COMM_ID_BYTES)
        handle = L.r3d_comm_create(ident, rank, world, torch.device(device).index or 0)
        # every rank learns whether ALL of them have a communicator: a rank without one must not be waited for
        ok = torch.tensor([1 if handle else 0], dtype=torch.int32, device=on)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if not int(ok.item()):
            err = "" if handle else L.r3d_last_error().decode()
            if handle:
                L.r3d_comm_destroy(handle)
            raise RuntimeError("r3d_comm_create failed on some rank" + (": " + err if err else ""))
        return cls(handle, L)

    def describe(self):
        from . import _ffi
        info = _ffi.CommInfo()
        if self._lib.r3d_comm_describe(self._c, C.byref(info)):
            raise RuntimeError("r3d_comm_describe failed: " + self._lib.r3d_last_error().decode())
        return {"n_ranks": info.n_ranks, "rank": info.rank, "device": info.device, "rccl_version": info.rccl_version,
                "device_uuid": info.device_uuid.decode(), "library": info.library.decode()}

    def reduce_(self, energy, counts, scalars, root=-1, stream=None):
        """Sum the three device buffers over the ranks in place (root < 0: every rank ends with the sums)."""
        s = stream if stream is not None else torch.cuda.current_stream(energy.device).cuda_stream
        if self._lib.r3d_comm_reduce(self._c, energy.data_ptr(), energy.numel(), counts.data_ptr(), counts.numel(),
                                     scalars.data_ptr(), scalars.numel(), root, s):
            raise RuntimeError("r3d_comm_reduce failed: " + self._lib.r3d_last_error().decode())

    def close(self):
        if self._c:
            self._lib.r3d_comm_destroy(self._c)
            self._c = None


class DeviceResult:
    """Result block resident in HBM (torch tensors are only the allocator; the engine writes through raw
    pointers and the library's communicator -- `comm`, a Comm -- reduces them).  Without a communicator
    the block is summed by torch.distributed (host tensors in the gloo tests)."""

    def __init__(self, model, device, comm=None):
        from . import _ffi
        n = model.n_seismometers * model.n_bins
        self.model = model
        self.comm = comm
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
        if self.comm is not None:     # the product's reduce (r3d_comm_reduce): one grouped all-reduce of the three buffers
            self.comm.reduce_(self.energy, self.counts, self.scalars)
        elif dist.is_available() and dist.is_initialized():
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
    (r3d_engine_set_volume_buffer) and the grids are added ONCE, at the end of the job:

    reduce_scatter_frames_()   the job's reduction ("keep sharded by frame"): rank r
        ends up with the job's counts for ITS range of frames, [r F / N, (r + 1) F / N) of both wave
        types -- a video frame is rendered from one frame's cells, so nobody needs all of them.
        Sparse form (the usual case: ~10 events per history touch < 5 % of config 5's 2.5e9 cells):
        every rank compacts, per owner, the non-zero counters of the owner's frames into 8-byte
        (index, count) pairs (r3d_volume_compact), the pairs travel point to point -- every
        xGMI link of the node's full mesh at once --, the owner adds what it receives
        (r3d_volume_scatter_add, saturating at 2^32 - 1).  Dense form (a grid too full for its
        pair buffer): one reduce(SUM) per owner and wave type on the int32 storage itself,
        (N - 1) / N of the grid per rank on the wire; widened to int64 and saturating when a
        cell's sum could reach 2^31.
    allgather_frames_()        afterwards and outside any timed region, for a caller that does want
        the whole grid everywhere: every owner broadcasts its frames.
    allreduce_() / reduce_()   the whole grid on every rank / on one rank in one go (round 3's
        form: 2 (N - 1) / N x 10 GB per rank on the wire; kept for small grids and as the
        reference point of the tests).

    `owned` says which frames of this rank's buffer hold JOB totals after a reduction: None = all
    of them (or none was run: the rank's own counts), (lo, hi) after reduce_scatter_frames_, () on
    the ranks a reduce_ left with scratch.  Reads outside `owned` raise.
    (torch has no arithmetic on uint32, so the storage is int32 holding the same bits.)"""

    def __init__(self, engine, origin, cell_size, dims, n_frames, frame_dt, device):
        self.shape = (2, int(n_frames), int(dims[2]), int(dims[1]), int(dims[0]))
        self.n_frames = int(n_frames)
        self.frame_cells = int(dims[2]) * int(dims[1]) * int(dims[0])
        n = 2 * self.n_frames * self.frame_cells
        self.counters = torch.zeros(n, dtype=torch.int32, device=device)
        self.saturated = 0
        self.widened = None
        self.owned = None
        self.timing = {}
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
        self.owned = None
        self.widened = None

    # ---- who owns what ---------------------------------------------------------------------------
    def frame_range(self, rank, world):
        """Frames [lo, hi) whose job totals rank `rank` of `world` holds after reduce_scatter_frames_."""
        return shard_range(self.n_frames, rank, world)

    def _segments(self, lo, hi):
        """The two contiguous runs of counters (one per wave type) that hold frames [lo, hi)."""
        fc = self.frame_cells
        return [((t * self.n_frames + lo) * fc, (t * self.n_frames + hi) * fc) for t in (0, 1) if hi > lo]

    def _group(self):
        return dist.is_available() and dist.is_initialized()

    # ---- dense reductions --------------------------------------------------------------------------
    def _headroom(self, world):
        """True when no cell's sum over `world` ranks can reach 2^31: the counters can then be added
        as they are stored (int32), with no widening.  One scalar all-reduce (MAX) decides it for
        all ranks alike."""
        lo, hi = int(self.counters.min().item()), int(self.counters.max().item())
        worst = torch.tensor([(1 << 32) if lo < 0 else hi], dtype=torch.int64, device=self.counters.device)
        dist.all_reduce(worst, op=dist.ReduceOp.MAX)
        return int(worst.item()) * world < (1 << 31)

    def _reduce_run(self, collective, b, e, chunk_elems, keep, headroom):
        """Add counters[b:e] over ranks with `collective`.  Usual case (every rank's largest counter
        x ranks < 2^31): on the int32 storage itself, in place, 4 bytes per cell on the wire.
        Otherwise: widen, add, saturate, store back -- chunk by chunk so that the int64 scratch stays
        small next to a multi-GB grid (8 bytes per cell on the wire)."""
        for lo in range(b, e, chunk_elems):
            part = self.counters[lo:min(lo + chunk_elems, e)]
            if headroom:
                collective(part)
                continue
            wide = _as_unsigned(part)
            collective(wide)
            if keep:
                self.saturated += int((wide > U32_MAX).sum().item())
                part.copy_(_to_storage(wide.clamp_(max=U32_MAX)))

    def _reduce_whole(self, collective, chunk_elems, keep):
        if not self._group():
            return self
        # (also at world size 1: a job of one rank takes the same path through the collective library)
        headroom = self._headroom(dist.get_world_size())
        self.widened = not headroom
        self._reduce_run(collective, 0, self.counters.numel(), chunk_elems if headroom else min(chunk_elems, 1 << 26),
                         keep, headroom)
        return self

    def allreduce_(self, chunk_elems=1 << 28):
        """Every rank ends with the job's grid (one all-reduce(SUM) per chunk of 1 GiB)."""
        self._reduce_whole(lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM), chunk_elems, True)
        self.owned = None
        return self

    def reduce_(self, dst=0, chunk_elems=1 << 28):
        """Rank `dst` ends with the job's grid; the other ranks' buffers are scratch afterwards (a
        reduce may use them so) and refuse to be read."""
        if not self._group():
            return self
        mine = dist.get_rank() == dst
        self._reduce_whole(lambda t: dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM), chunk_elems, mine)
        self.owned = None if mine else ()
        return self

    # ---- the job's reduction: by frame ------------------------------------------------------------
    def reduce_scatter_frames_(self, mode="auto", pair_capacity=None, chunk_elems=1 << 28):
        """Rank r ends with the job's counts for its frame_range(r, N) (both wave types); the rest of
        its buffer is stale.  mode: "sparse" (pairs, point to point), "dense" (one reduce per owner
        and wave type), "auto" = sparse unless some rank's pairs do not fit `pair_capacity`
        (default: a sixteenth of the cells, i.e. half the grid's bytes) -- decided for all ranks
        alike from the exchanged counts.  `timing` records the phases."""
        if mode not in ("auto", "sparse", "dense"):
            raise ValueError(mode)
        if not self._group():
            return self
        rank, world = dist.get_rank(), dist.get_world_size()
        self.timing = {"mode": None}
        t0 = self._clock()
        if mode != "dense" and self.counters.numel() < (1 << 32):
            cap = int(pair_capacity if pair_capacity is not None else max(1024, self.counters.numel() // 16))
            if self._reduce_scatter_sparse(rank, world, cap, t0):
                return self
            if mode == "sparse":   # (the same on every rank: they all saw the same counts)
                raise RuntimeError("sparse reduction of the grid refused: " + self.timing["sparse_refused"])
        elif mode == "sparse":
            raise RuntimeError("sparse reduction of the grid needs indices below 2^32")
        headroom = self._headroom(world)
        self.widened = not headroom
        for owner in range(world):
            for b, e in self._segments(*self.frame_range(owner, world)):
                self._reduce_run(lambda t: dist.reduce(t, dst=owner, op=dist.ReduceOp.SUM), b, e,
                                 chunk_elems if headroom else min(chunk_elems, 1 << 26), rank == owner, headroom)
        self.owned = self.frame_range(rank, world)
        self.timing.update(mode="dense int64, saturating" if self.widened else "dense int32",
                           total_s=self._clock() - t0,
                           bytes_sent=(self.counters.numel() - sum(e - b for b, e in self._segments(*self.owned)))
                           * (8 if self.widened else 4))
        return self

    def _clock(self):
        import time
        if self.counters.is_cuda:
            torch.cuda.synchronize(self.counters.device)
        return time.perf_counter()

    def _compact(self, b, e, pairs, n_dev, cap):
        """Append the non-zero counters of [b, e) to `pairs` (int32 [cap, 2]) at *n_dev."""
        if self.counters.is_cuda:
            from . import _ffi
            lib = _ffi.hip_lib()
            if lib.r3d_volume_compact(self.counters.device.index, self.counters.data_ptr(), b, e, pairs.data_ptr(), cap,
                                      n_dev.data_ptr(), torch.cuda.current_stream(self.counters.device).cuda_stream):
                raise RuntimeError("r3d_volume_compact failed: " + lib.r3d_last_error().decode())
            return
        # host tensors (the gloo tests of the exchange): the same result with torch operations
        seg = self.counters[b:e]
        idx = seg.nonzero().flatten()
        at = int(n_dev.item())
        fit = max(0, min(idx.numel(), cap - at))
        pairs[at:at + fit, 0] = _to_storage(idx[:fit] + b)
        pairs[at:at + fit, 1] = seg[idx[:fit]]
        n_dev += idx.numel()

    def _scatter_add(self, pairs):
        """counters[index] += count for the received pairs, saturating."""
        if pairs.shape[0] == 0:
            return
        if self.counters.is_cuda:
            from . import _ffi
            lib = _ffi.hip_lib()
            flags = torch.zeros(2, dtype=torch.int64, device=self.counters.device)
            if lib.r3d_volume_scatter_add(self.counters.device.index, self.counters.data_ptr(), self.counters.numel(),
                                          pairs.data_ptr(), pairs.shape[0], flags.data_ptr(),
                                          torch.cuda.current_stream(self.counters.device).cuda_stream):
                raise RuntimeError("r3d_volume_scatter_add failed: " + lib.r3d_last_error().decode())
            sat, stray = (int(v) for v in flags.tolist())
            if stray:
                raise RuntimeError(f"{stray} received pairs lie outside the grid")
            self.saturated += sat
            return
        idx, val = _as_unsigned(pairs[:, 0]), _as_unsigned(pairs[:, 1])
        b, e = int(idx.min().item()), int(idx.max().item()) + 1
        wide = _as_unsigned(self.counters[b:e])
        wide.index_add_(0, idx - b, val)
        self.saturated += int((wide > U32_MAX).sum().item())
        self.counters[b:e] = _to_storage(wide.clamp_(max=U32_MAX))

    def _reduce_scatter_sparse(self, rank, world, cap, t0):
        """The sparse form; returns False (nothing changed) when some rank's pairs do not fit."""
        dev = self.counters.device
        pairs = torch.empty((cap, 2), dtype=torch.int32, device=dev)
        n_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        # ends[o] = pairs written (or merely counted) once owner o's frames are compacted.  Kept ON THE DEVICE until the
        # ranks' counts have been exchanged: one read-back for the whole reduction, not one per owner (N host
        # synchronisations inside the timed region, serial in N)
        ends_dev = torch.zeros(world, dtype=torch.int64, device=dev)
        for owner in range(world):
            if owner != rank:   # (this rank's own frames stay where they are)
                for b, e in self._segments(*self.frame_range(owner, world)):
                    self._compact(b, e, pairs, n_dev, cap)
            ends_dev[owner:owner + 1] = n_dev
        counts = ends_dev - torch.cat([ends_dev.new_zeros(1), ends_dev[:-1]])
        matrix = [torch.zeros_like(counts) for _ in range(world)]
        dist.all_gather(matrix, counts)                       # matrix[src][dst] = pairs src has for dst
        host = torch.cat([torch.stack(matrix).reshape(-1), ends_dev]).cpu()   # (the one read-back)
        matrix, ends = host[:world * world].reshape(world, world), [int(v) for v in host[world * world:].tolist()]
        t1 = self._clock()
        if int(matrix.sum(1).max().item()) > cap:             # (every rank sees the same matrix: one decision)
            self.timing["sparse_refused"] = f"{int(matrix.sum(1).max().item())} pairs on some rank, capacity {cap}"
            return False
        incoming = [int(matrix[src][rank].item()) for src in range(world)]
        recv = torch.empty((sum(incoming), 2), dtype=torch.int32, device=dev)
        ops, at = [], 0
        for peer in range(world):
            if peer == rank:
                continue
            lo, hi = (ends[peer - 1] if peer else 0), ends[peer]
            if hi > lo:
                ops.append(dist.P2POp(dist.isend, pairs[lo:hi], peer))
            if incoming[peer]:
                ops.append(dist.P2POp(dist.irecv, recv[at:at + incoming[peer]], peer))
                at += incoming[peer]
        if ops:
            for work in dist.batch_isend_irecv(ops):
                work.wait()
        t2 = self._clock()
        self._scatter_add(recv)
        self.owned = self.frame_range(rank, world)
        t3 = self._clock()
        self.widened = None
        self.timing.update(mode="sparse pairs", compact_s=t1 - t0, exchange_s=t2 - t1, add_s=t3 - t2, total_s=t3 - t0,
                           pairs_sent=ends[-1], pairs_received=sum(incoming), bytes_sent=8 * ends[-1])
        return True

    def allgather_frames_(self, chunk_elems=1 << 28):
        """After reduce_scatter_frames_: every owner broadcasts its frames, every rank ends with the
        job's whole grid (for callers that want it everywhere; not part of the job's reduction)."""
        if not self._group():
            return self
        # Every rank must enter the broadcasts or none: after reduce_(dst) the destination holds everything
        # (owned None) while the others hold scratch (owned ()) -- the ranks would part ways and the ones that
        # went on would wait for ever.  So the ranks first agree on what state they are in, with one small
        # all-reduce, and a job that is not in the by-frame state everywhere gets an error on EVERY rank.
        state = 0 if self.owned is None else (2 if self.owned == () else 1)
        seen = torch.zeros(3, dtype=torch.int64, device=self.counters.device)
        seen[state] = 1
        dist.all_reduce(seen, op=dist.ReduceOp.SUM)
        n_all, n_frames, n_scratch = (int(v) for v in seen.tolist())
        if n_scratch or (n_all and n_frames):
            raise RuntimeError("allgather_frames_ needs every rank's grid reduced by frame (reduce_scatter_frames_): "
                               f"{n_frames} ranks hold frames, {n_all} hold a whole grid, {n_scratch} hold scratch of a reduce_")
        if n_frames == 0:
            return self          # nothing was reduced by frame anywhere: every rank's own counts, as they stand
        world = dist.get_world_size()
        for owner in range(world):
            for b, e in self._segments(*self.frame_range(owner, world)):
                for lo in range(b, e, chunk_elems):
                    dist.broadcast(self.counters[lo:min(lo + chunk_elems, e)], src=owner)
        self.owned = None
        return self

    # ---- reading -----------------------------------------------------------------------------------
    def _valid_runs(self):
        if self.owned is None:
            return [(0, self.counters.numel())]
        if self.owned == ():
            raise RuntimeError("this rank's grid was scratch of a reduce_ to another rank: it holds nothing to read")
        return self._segments(*self.owned)

    def total(self, chunk_elems=1 << 26):
        """Sum of the counters this rank holds job totals for (all of them unless a reduction by frame
        has run), computed chunk by chunk."""
        t = 0
        for b, e in self._valid_runs():
            for lo in range(b, e, chunk_elems):
                t += int(_as_unsigned(self.counters[lo:min(lo + chunk_elems, e)]).sum().item())
        return t

    def job_total(self):
        """Events binned by the whole job: total() summed over the owners after reduce_scatter_frames_."""
        t = self.total()
        if self._group() and self.owned is not None:
            v = torch.tensor([t], dtype=torch.int64, device=self.counters.device)
            dist.all_reduce(v, op=dist.ReduceOp.SUM)
            t = int(v.item())
        return t

    def frames_numpy(self):
        """(lo, hi, counts[2][hi - lo][z][y][x]) of the frames this rank holds job totals for."""
        import numpy as np
        if self.owned == ():
            self._valid_runs()
        lo, hi = (0, self.n_frames) if self.owned is None else self.owned
        full = self.counters.view(self.shape)[:, lo:hi]
        return lo, hi, full.cpu().numpy().view(np.uint32)

    def to_numpy(self):
        """The whole grid; refuses while only a range of frames holds job totals."""
        import numpy as np
        if self.owned is not None:
            self._valid_runs() if self.owned == () else None
            raise RuntimeError(f"only frames {self.owned} of this rank's grid hold job totals "
                               "(reduce_scatter_frames_): read frames_numpy(), or allgather_frames_() first")
        return self.counters.cpu().numpy().view(np.uint32).reshape(self.shape)
