"""Host-side Python mirror of the reference's run interface for the hot path.

``Model(args)`` takes the reference's own command-line tokens (the ones a
do-*.sh script assembles, scripts/do-fundamentals.sh:396-419) and builds the
flat tables through the C++ builder (libr3d_host.so).  ``Engine(model)`` puts
them in HBM and ``Engine.run(n, first_id, seed)`` is the drop-in for the body
of ``Model::RunSimulation()`` (reference model.cpp:602-633): N histories of
GenerateEventPhonon + Propagate, returning filled seismometer bins and the
loss counters.

There is no CPU fallback here: without libr3d_hip.so (or without a GPU)
``Engine`` raises.
"""
import ctypes as C

import numpy as np

from . import _ffi


class Result:
    """Seismometer bins + counters of a run (reference BinRecord,
    dataout.hpp:77-93, and DataReporter counters, dataout.cpp:591-617)."""

    def __init__(self, n_seis, n_bins):
        self.n_seis, self.n_bins = n_seis, n_bins
        self.energy = np.zeros((n_seis, n_bins, _ffi.R3D_N_ENERGY), dtype=np.float64)
        self.counts = np.zeros((n_seis, n_bins, _ffi.R3D_N_COUNT), dtype=np.uint64)
        self.n_lost = self.n_timeout = self.n_invalid = 0
        self.invalid_reasons = np.zeros(_ffi.R3D_INV_NUM, dtype=np.uint64)
        self.events = dict.fromkeys(_ffi.R3D_EV_NAMES, 0)

    # -- C view -------------------------------------------------------------
    def _as_c(self):
        r = _ffi.Result()
        r.energy = self.energy.ctypes.data_as(C.POINTER(C.c_double))
        r.counts = self.counts.ctypes.data_as(C.POINTER(C.c_uint64))
        r.n_lost, r.n_timeout, r.n_invalid = self.n_lost, self.n_timeout, self.n_invalid
        for i in range(_ffi.R3D_INV_NUM):
            r.invalid_reasons[i] = int(self.invalid_reasons[i])
        for i, k in enumerate(_ffi.R3D_EV_NAMES):
            r.events[i] = self.events[k]
        return r

    def _from_c(self, r):
        self.n_lost, self.n_timeout, self.n_invalid = int(r.n_lost), int(r.n_timeout), int(r.n_invalid)
        for i in range(_ffi.R3D_INV_NUM):
            self.invalid_reasons[i] = r.invalid_reasons[i]
        for i, k in enumerate(_ffi.R3D_EV_NAMES):
            self.events[k] = int(r.events[i])

    @property
    def diag_invalid(self):
        """7-bit OR of the invalid reasons (DataReporter::mDiagInvalid)."""
        return sum(1 << i for i in range(_ffi.R3D_INV_NUM) if self.invalid_reasons[i])

    def scalars(self):
        return np.array([self.n_lost, self.n_timeout, self.n_invalid, *self.invalid_reasons,
                         *[self.events[k] for k in _ffi.R3D_EV_NAMES]], dtype=np.uint64)

    def set_scalars(self, v):
        v = [int(x) for x in v]
        self.n_lost, self.n_timeout, self.n_invalid = v[0:3]
        self.invalid_reasons[:] = v[3:3 + _ffi.R3D_INV_NUM]
        for i, k in enumerate(_ffi.R3D_EV_NAMES):
            self.events[k] = v[3 + _ffi.R3D_INV_NUM + i]


class Model:
    """A built, immutable Earth model + source + seismometers."""

    def __init__(self, args):
        if isinstance(args, str):
            args = args.split()
        self._lib = _ffi.host_lib()
        self.args = list(args)
        argv = (C.c_char_p * len(args))(*[a.encode() for a in args])
        self._h = self._lib.r3dh_model_from_args(len(args), argv)
        if not self._h:
            raise RuntimeError("model build failed: " + self._lib.r3dh_last_error().decode())
        self.desc_p = self._lib.r3dh_model_desc(self._h)
        self.desc = self.desc_p.contents

    def close(self):
        if getattr(self, "_h", None):
            self._lib.r3dh_model_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- facts about the model ------------------------------------------------
    @property
    def n_cells(self):
        return self.desc.n_cells

    @property
    def n_scatterers(self):
        return self.desc.n_scatterers

    @property
    def n_seismometers(self):
        return self.desc.n_seismometers

    @property
    def n_bins(self):
        return self.desc.params.n_bins

    @property
    def n_toa(self):
        return self.desc.n_toa

    @property
    def num_phonons(self):
        return int(self._lib.r3dh_num_phonons(self._h))

    @property
    def seed(self):
        return int(self._lib.r3dh_seed(self._h))

    @property
    def log(self):
        return self._lib.r3dh_model_log(self._h).decode()

    def grid_dump(self):
        return self._lib.r3dh_grid_dump(self._h).decode()

    def scatterer_info(self, i):
        out = (C.c_double * 10)()
        if self._lib.r3dh_scatterer_info(self._h, i, out):
            raise IndexError(i)
        keys = ("nu", "eps", "a", "kappa", "el", "gam0", "mfp_p", "mfp_s", "dipole_p", "dipole_s")
        return dict(zip(keys, out))

    def scatterer_dump(self):
        return self._lib.r3dh_scatterer_dump(self._h).decode()

    def params_echo(self):
        return self._lib.r3dh_params_echo(self._h).decode()

    def write_outputs(self, result, outdir, trace_path=None, mparams_path=None):
        """Write seis_NNN.octv (+ ASCII traces, + parameter file) in the reference's
        formats; returns the post-sim console summary."""
        import os
        trace_path = trace_path or os.path.join(outdir or ".", "seis_traces_asc.dat")
        c = result._as_c()
        txt = self._lib.r3dh_write_outputs(self._h, C.byref(c), (outdir or "").encode(), trace_path.encode(),
                                           mparams_path.encode() if mparams_path else None)
        if txt is None:
            raise RuntimeError("write_outputs failed: " + self._lib.r3dh_last_error().decode())
        return txt.decode()

    @property
    def coordinates(self):
        """(map code, Earth radius, flattened) of the coordinate system the model was built in
        (reference ecs.hpp:242-257: 0 ENU_ORTHO, 1 RAE_ORTHO, 2 RAE_CURVED, 3 RAE_SPHERICAL)."""
        code, flat, rad = C.c_int(), C.c_int(), C.c_double()
        self._lib.r3dh_model_coordinates(self._h, C.byref(code), C.byref(rad), C.byref(flat))
        return code.value, rad.value, bool(flat.value)

    def grid_nodes(self):
        """((ni, nj, nk), array of _ffi.GridNode): the grid as the cell builders see it.  Call right
        after building the model (the coordinate system is process-global, as in the reference)."""
        dims = (C.c_int * 3)()
        self._lib.r3dh_grid_size(self._h, dims)
        n = dims[0] * dims[1] * dims[2]
        nodes = (_ffi.GridNode * n)()
        if self._lib.r3dh_grid_nodes(self._h, nodes, n):
            raise RuntimeError("r3dh_grid_nodes failed: " + self._lib.r3dh_last_error().decode())
        return tuple(dims), nodes

    def grid_nodes_raw(self):
        """array of _ffi.GridNodeRaw in the grid's own order: the nodes before the coordinate system's conversion."""
        dims = (C.c_int * 3)()
        self._lib.r3dh_grid_size(self._h, dims)
        n = dims[0] * dims[1] * dims[2]
        nodes = (_ffi.GridNodeRaw * n)()
        if self._lib.r3dh_grid_nodes_raw(self._h, nodes, n):
            raise RuntimeError("r3dh_grid_nodes_raw failed: " + self._lib.r3dh_last_error().decode())
        return nodes

    def seismometer_axes(self, i):
        """0 ENZ, 1 RTZ (model.cpp:486-491)."""
        return int(self._lib.r3dh_seismometer_axes(self._h, i))

    @property
    def device_tables(self):
        """True if built with --device-tables: the scattering tables are made by the engine."""
        return bool(self._lib.r3dh_model_device_tables(self._h))

    @property
    def report_mask(self):
        """R3D_RPT_* mask asked for by --reports in the model's arguments."""
        return int(self._lib.r3dh_model_report_mask(self._h))

    def format_reports(self, events, path=None):
        """Event records (Engine.read_event_log) as the reference's report lines
        (dataout.cpp:484-520); written to `path`, or returned as text."""
        import numpy as np
        ev = np.ascontiguousarray(events, dtype=_ffi.event_dtype())
        txt = self._lib.r3dh_write_reports(self._h, ev.ctypes.data, len(ev), path.encode() if path else None)
        if txt is None:
            raise RuntimeError("write_reports failed: " + self._lib.r3dh_last_error().decode())
        return txt.decode()

    def new_result(self):
        return Result(self.n_seismometers, self.n_bins)


def run_model(model, n, first_id=0, seed=0x5EED, n_gpus=1, devices=None):
    """r3d_run_model: the whole seam in one call, sharded over devices 0 .. n_gpus-1 -- or, with
    `devices`, r3d_run_model_on: shard g on devices[g] (a device may be named more than once)."""
    lib = _ffi.hip_lib()
    res = model.new_result()
    c = res._as_c()
    if devices is not None:
        devs = (C.c_int * len(devices))(*devices)
        rc = lib.r3d_run_model_on(model.desc_p, n, first_id, seed, devs, len(devices), C.byref(c))
    else:
        rc = lib.r3d_run_model(model.desc_p, n, first_id, seed, n_gpus, C.byref(c))
    if rc:
        raise RuntimeError("r3d_run_model failed: " + lib.r3d_last_error().decode())
    res._from_c(c)
    return res


class Node:
    """r3d_node_*: one engine per entry of `devices`, kept across runs; a run shards the id range over them
    and sums the shards' blocks on the devices (RCCL ncclReduce to devices[0]; on the host for a node of one
    shard, when two shards share a device or when RCCL is not to be had -- `reduction` says which,
    `reduction_note` why)."""

    def __init__(self, model, devices, lib=None):
        self.model = model
        self._lib = _ffi.hip_lib(path=lib)
        devs = (C.c_int * len(devices))(*devices)
        self._n = self._lib.r3d_node_create(model.desc_p, devs, len(devices))
        if not self._n:
            raise RuntimeError("r3d_node_create failed: " + self._lib.r3d_last_error().decode())

    @property
    def reduction(self):
        return self._lib.r3d_node_reduction(self._n).decode()

    @property
    def reduction_note(self):
        """Why the host adds the shards' blocks, when it does (r3d_node_reduction_note)."""
        return self._lib.r3d_node_reduction_note(self._n).decode()

    def __len__(self):
        return self._lib.r3d_node_size(self._n)

    def run(self, n, first_id=0, seed=0x5EED, result=None):
        res = result if result is not None else self.model.new_result()
        c = res._as_c()
        if self._lib.r3d_node_run(self._n, n, first_id, seed, C.byref(c)):
            raise RuntimeError("r3d_node_run failed: " + self._lib.r3d_last_error().decode())
        res._from_c(c)
        return res

    def close(self):
        if self._n:
            self._lib.r3d_node_destroy(self._n)
            self._n = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def reduce_volumes_by_frame(engines):
    """r3d_volume_reduce_by_frame: the grids of several engines of ONE process (one per shard of a job, each
    with its own grid of the same shape, on any devices) added by frame; engine g ends with the job's counts
    for frames [frames[g], frames[g + 1]).  Returns (frames, saturated cells)."""
    lib = engines[0]._lib
    n = len(engines)
    handles = (C.c_void_p * n)(*[e._e for e in engines])
    frames = (C.c_uint32 * (n + 1))()
    sat = C.c_uint64(0)
    if lib.r3d_volume_reduce_by_frame(handles, n, frames, C.byref(sat)):
        raise RuntimeError("r3d_volume_reduce_by_frame failed: " + lib.r3d_last_error().decode())
    return list(frames), int(sat.value)


def volume_desc(origin, cell_size, dims, n_frames, frame_dt):
    v = _ffi.VolumeDesc()
    for k in range(3):
        v.origin[k], v.cell_size[k], v.dims[k] = origin[k], cell_size[k], dims[k]
    v.n_frames, v.frame_dt = n_frames, frame_dt
    return v


class Engine:
    """The model resident in HBM + the HIP traversal kernels (libr3d_hip.so)."""

    def __init__(self, model, device=0, reproducible=False, lib=None, residency=None, pool_slots=None,
                 accumulator_bits=None, lds_reserve=None):
        """reproducible: the build without wave-voted series choices (see _ffi.hip_lib); lib: the path
        of another build of the engine (tools/).  residency, pool_slots, accumulator_bits, lds_reserve:
        the kernel's LDS carve-up in the caller's hands (include/r3d.h r3d_engine_opts) -- how the
        tests reach every compiled kernel variant on small models; None = automatic, and with all
        four None the engine is made by plain r3d_engine_create."""
        self._lib = _ffi.hip_lib(reproducible, lib)
        self.model = model
        self._volume_keepalive = None
        if residency is None and pool_slots is None and accumulator_bits is None and lds_reserve is None:
            self._e = self._lib.r3d_engine_create(model.desc_p, device)
        else:
            o = _ffi.EngineOpts(C.sizeof(_ffi.EngineOpts), -1 if residency is None else residency,
                                pool_slots or 0, -1 if accumulator_bits is None else accumulator_bits,
                                lds_reserve or 0)
            self._e = self._lib.r3d_engine_create_ex(model.desc_p, device, C.byref(o))
        if not self._e:
            raise RuntimeError("r3d_engine_create failed: " + self._lib.r3d_last_error().decode())
        if model.device_tables:   # mean free paths / dipoles are the engine's output
            for s in range(model.n_scatterers):
                st = self.scatterer_stats(s)
                model._lib.r3dh_model_set_scatterer_stats(model._h, s, (C.c_double * 2)(*st[0:2]),
                                                          (C.c_double * 2)(*st[2:4]))

    def close(self, discard_carried=False):
        """Release the engine.  Refuses (RuntimeError) while histories carried over by
        run_device(carry="carry") await their flush -- their tallies and bins would be lost --
        unless discard_carried is set."""
        if getattr(self, "_e", None):
            if discard_carried or self._lib.r3d_engine_close(self._e):
                if not discard_carried:
                    raise RuntimeError("r3d_engine_close failed: " + self._lib.r3d_last_error().decode())
                self._lib.r3d_engine_destroy(self._e)
            self._e = None
            self._volume_keepalive = None   # (only now may a caller-owned grid be freed)

    def __del__(self):
        try:
            self.close(discard_carried=True)
        except Exception:
            pass

    @property
    def carry_pending(self):
        return bool(self._lib.r3d_engine_carry_pending(self._e))

    def run(self, n, first_id=0, seed=0x5EED, result=None, trace=False):
        """Run histories [first_id, first_id+n); accumulate into `result`."""
        res = result if result is not None else self.model.new_result()
        c = res._as_c()
        finals = None
        if trace:
            finals = (_ffi.Final * n)()
            rc = self._lib.r3d_run_traced(self._e, n, first_id, seed, C.byref(c), finals)
        else:
            rc = self._lib.r3d_run(self._e, n, first_id, seed, C.byref(c))
        if rc:
            raise RuntimeError("r3d_run failed: " + self._lib.r3d_last_error().decode())
        res._from_c(c)
        return (res, finals) if trace else res

    def run_device(self, n, first_id, seed, d_energy, d_counts, d_scalars, stream=None, carry=None):
        """Asynchronous, device-resident accumulate (pointers are raw device
        addresses, e.g. tensor.data_ptr()).  carry: None (a self-contained launch),
        "carry" (one of a chain: unfinished histories stay in the engine for its next
        launch) or "final" (resume and finish everything carried); see r3d_run_device_carry."""
        if carry not in (None, "carry", "final"):
            raise ValueError(f"carry must be None, 'carry' or 'final', not {carry!r}")
        if carry is not None:
            rc = self._lib.r3d_run_device_carry(self._e, n, first_id, seed, d_energy, d_counts, d_scalars,
                                                stream, 1 if carry == "final" else 0)
        else:
            rc = self._lib.r3d_run_device(self._e, n, first_id, seed, d_energy, d_counts, d_scalars,
                                          None, stream)
        if rc:
            raise RuntimeError("r3d_run_device failed: " + self._lib.r3d_last_error().decode())

    # -- volumetric scatter-event grid (config 5 of BASELINE.json) -----------
    def set_volume(self, origin, cell_size, dims, n_frames, frame_dt):
        """Attach count[type][frame][z][y][x] (uint32, HBM) filled at SCT / REF events."""
        v = volume_desc(origin, cell_size, dims, n_frames, frame_dt)
        if self._lib.r3d_engine_set_volume(self._e, C.byref(v)):
            raise RuntimeError("r3d_engine_set_volume failed: " + self._lib.r3d_last_error().decode())
        self._vol_shape = (2, int(n_frames), int(dims[2]), int(dims[1]), int(dims[0]))

    def set_volume_buffer(self, origin, cell_size, dims, n_frames, frame_dt, counters):
        """The same grid in caller-owned device memory: `counters` is a contiguous 32-bit integer
        tensor of 2*n_frames*nz*ny*nx zeroed elements (e.g. a torch tensor that is reduced over
        ranks afterwards).  The engine writes through its raw address, so this object keeps a
        reference to the tensor until the grid is detached or the engine closed."""
        shape = (2, int(n_frames), int(dims[2]), int(dims[1]), int(dims[0]))
        want = 1
        for d in shape:
            want *= d
        if not hasattr(counters, "data_ptr"):
            raise TypeError("set_volume_buffer takes the tensor itself, not an address: the engine "
                            "must keep it alive while it adds into it")
        if counters.numel() != want or counters.element_size() != 4 or not counters.is_contiguous():
            raise ValueError(f"volume buffer must be {want} contiguous 32-bit counters, got "
                             f"{counters.numel()} x {counters.element_size()} bytes")
        v = volume_desc(origin, cell_size, dims, n_frames, frame_dt)
        if self._lib.r3d_engine_set_volume_buffer(self._e, C.byref(v), counters.data_ptr()):
            raise RuntimeError("r3d_engine_set_volume_buffer failed: " + self._lib.r3d_last_error().decode())
        assert self._lib.r3d_volume_len(self._e) == want
        self._volume_keepalive = counters
        self._vol_shape = shape

    def detach_volume(self):
        """Stop binning events (r3d_engine_set_volume_buffer(NULL)); releases the caller's tensor."""
        if getattr(self, "_e", None):
            if self._lib.r3d_engine_set_volume_buffer(self._e, None, None):
                raise RuntimeError("detaching the volume failed: " + self._lib.r3d_last_error().decode())
        self._volume_keepalive = None

    def set_production_finals(self, base_id, capacity):
        """Final records out of the production kernels for ids [base_id, base_id + capacity) (capacity 0: off)."""
        if self._lib.r3d_engine_set_production_finals(self._e, base_id, capacity):
            raise RuntimeError("r3d_engine_set_production_finals failed: " + self._lib.r3d_last_error().decode())

    def production_finals(self, first, count):
        out = (_ffi.Final * count)()
        if self._lib.r3d_production_finals_read(self._e, out, first, count):
            raise RuntimeError("r3d_production_finals_read failed: " + self._lib.r3d_last_error().decode())
        return out

    def read_volume(self, reset=False):
        out = np.zeros(self._vol_shape, dtype=np.uint32)
        assert out.size == self._lib.r3d_volume_len(self._e)
        if self._lib.r3d_volume_read(self._e, out.ctypes.data_as(C.POINTER(C.c_uint32)), int(reset)):
            raise RuntimeError("r3d_volume_read failed: " + self._lib.r3d_last_error().decode())
        return out

    def volume_device_ptr(self):
        return self._lib.r3d_volume_device_ptr(self._e)

    # -- scattering tables as the engine holds them ------------------------------
    def scatterer_stats(self, s):
        """[mfp_p, mfp_s, dipole_p, dipole_s, total_pp, total_ps, total_sp, total_ss]"""
        out = (C.c_double * 8)()
        if self._lib.r3d_engine_scatterer_stats(self._e, s, out):
            raise IndexError(s)
        return list(out)

    def download_scatterer(self, s):
        """(cdf[4, n_toa], spol[n_toa]) copied from HBM."""
        n = self.model.n_toa
        cdf, spol = np.zeros((4, n)), np.zeros(n)
        ptrs = (_ffi._dp * 4)(*[cdf[k].ctypes.data_as(_ffi._dp) for k in range(4)])
        if self._lib.r3d_engine_download_scatterer(self._e, s, ptrs, spol.ctypes.data_as(_ffi._dp)):
            raise RuntimeError("download failed: " + self._lib.r3d_last_error().decode())
        return cdf, spol

    def download_source(self):
        """(cdf[3, n_toa], whole[3]): the source's cumulative P / SH / SV tables as the engine holds them."""
        n = self.model.n_toa
        cdf, whole = np.zeros((3, n)), np.zeros(3)
        ptrs = (_ffi._dp * 3)(*[cdf[k].ctypes.data_as(_ffi._dp) for k in range(3)])
        if self._lib.r3d_engine_download_source(self._e, ptrs, whole.ctypes.data_as(_ffi._dp)):
            raise RuntimeError("download failed: " + self._lib.r3d_last_error().decode())
        return cdf, whole

    def download_toa(self):
        """toa[n_toa, 2]: the take-off set (theta, phi) as the engine holds it."""
        toa = np.zeros((self.model.n_toa, 2))
        if self._lib.r3d_engine_download_toa(self._e, toa.ctypes.data_as(_ffi._dp)):
            raise RuntimeError("download failed: " + self._lib.r3d_last_error().decode())
        return toa

    # -- per-event report stream (the reference's --reports) -------------------
    def set_event_log(self, mask=_ffi.R3D_RPT_ALL, capacity=1 << 20):
        """Attach an HBM buffer of `capacity` r3d_event records for the tags in `mask`."""
        if self._lib.r3d_engine_set_event_log(self._e, int(mask), int(capacity)):
            raise RuntimeError("r3d_engine_set_event_log failed: " + self._lib.r3d_last_error().decode())
        self._ev_cap = int(capacity) if mask else 0

    def event_log_count(self):
        return int(self._lib.r3d_event_log_count(self._e))

    def read_event_log(self, reset=False):
        """Stored records as a numpy structured array (_ffi.event_dtype)."""
        n = min(self.event_log_count(), self._ev_cap)
        out = np.zeros(n, dtype=_ffi.event_dtype())
        got = self._lib.r3d_event_log_read(self._e, out.ctypes.data, n, int(reset))
        if got == (1 << 64) - 1:
            raise RuntimeError("r3d_event_log_read failed: " + self._lib.r3d_last_error().decode())
        return out[:got]

    @property
    def variant(self):
        """(cell kind, table residency) of the compiled kernel this engine launches:
        kind 0 cylinder / 1 tetra / 2 sphere shell; residency 0 cells + scatterer heads in LDS,
        1 heads only, 2 neither."""
        v = int(self._lib.r3d_engine_variant(self._e))
        return v // 4, v % 4

    @property
    def pool_slots(self):
        return int(self._lib.r3d_engine_pool_slots(self._e))

    @property
    def accumulators(self):
        """entries of a workgroup's LDS table of bin accumulators"""
        return int(self._lib.r3d_engine_accumulators(self._e))

    def last_kernel_ms(self):
        return float(self._lib.r3d_last_kernel_ms(self._e))

    def launch_count(self):
        """Launches enqueued so far; launch ids run from 1."""
        return int(self._lib.r3d_launch_count(self._e))

    def kernel_ms(self, launch):
        """Kernel time of launch id `launch` (one of the 64 most recent), -1 if not on record."""
        return float(self._lib.r3d_kernel_ms(self._e, int(launch)))
