"""ctypes mirror of include/r3d.h and include/r3d_host.h.

Only plain C crosses the boundary; these Structure classes restate the POD
layouts field for field.  tests/test_abi.py checks sizes against the compiled
libraries.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBDIR = os.path.join(_HERE, "lib")
REPO = os.path.dirname(_HERE)

R3D_CELL_CYLINDER, R3D_CELL_TETRA, R3D_CELL_SPHERESHELL = 0, 1, 2
R3D_FACE_COLLECT, R3D_FACE_REFLECT, R3D_FACE_ADJOIN, R3D_FACE_DISCON = 1, 2, 4, 8
R3D_INV_NUM = 7
R3D_EV_NAMES = ("generated", "iterations", "scatter", "collect", "catch", "reflect",
                "transfer", "rtsolve", "volume_out")
R3D_EV_NUM = len(R3D_EV_NAMES)
R3D_N_ENERGY, R3D_N_COUNT = 5, 2
R3D_N_SCALARS = 3 + R3D_INV_NUM + R3D_EV_NUM

_dp = C.POINTER(C.c_double)


class Face(C.Structure):
    _fields_ = [("normal", C.c_double * 3), ("point", C.c_double * 3), ("radius", C.c_double),
                ("neighbor", C.c_int32), ("flags", C.c_uint32)]


class Cell(C.Structure):
    _fields_ = [("vel_c", C.c_double * 2), ("vel_a", C.c_double * 2),
                ("vel_grad", (C.c_double * 3) * 2), ("rho_c", C.c_double), ("rho_a", C.c_double),
                ("rho_grad", C.c_double * 3), ("q", C.c_double * 2), ("zero_rad2", C.c_double * 2),
                ("scatterer", C.c_int32), ("n_faces", C.c_int32), ("faces", Face * 4)]


class Scatterer(C.Structure):
    _fields_ = [("mfp", C.c_double * 2), ("whole_cdf", (C.c_double * 4) * 2), ("cdf", _dp * 4),
                ("spol", _dp), ("het", C.c_double * 6), ("psdf_numer", C.c_double),
                ("mfp_fixed", C.c_uint32), ("pad_", C.c_uint32)]


class Source(C.Structure):
    _fields_ = [("loc", C.c_double * 3), ("cell", C.c_int32), ("pad_", C.c_int32),
                ("whole_cdf", C.c_double * 3), ("cdf", _dp * 3), ("moment", C.c_double * 6)]


class Seismometer(C.Structure):
    _fields_ = [("loc", C.c_double * 3), ("axes", (C.c_double * 3) * 3), ("r_in", C.c_double * 2),
                ("r_out", C.c_double * 2), ("area", C.c_double * 2)]


class Params(C.Structure):
    _fields_ = [("ttl", C.c_double), ("frequency", C.c_double), ("time_per_bin", C.c_double),
                ("n_bins", C.c_uint32), ("no_deflect", C.c_uint32), ("min_theta", C.c_double),
                ("max_theta", C.c_double), ("slow_concern", C.c_double),
                ("loop_concern", C.c_uint64), ("earth_center", C.c_double * 3)]


class ModelDesc(C.Structure):
    _fields_ = [("cell_kind", C.c_int32), ("n_cells", C.c_int32), ("cells", C.POINTER(Cell)),
                ("n_scatterers", C.c_int32), ("n_seismometers", C.c_int32),
                ("scatterers", C.POINTER(Scatterer)), ("seismometers", C.POINTER(Seismometer)),
                ("n_toa", C.c_uint64), ("toa", _dp), ("source", Source), ("params", Params),
                ("toa_degree", C.c_int32), ("pad_", C.c_int32)]


class GridNode(C.Structure):
    """r3dh_grid_node (include/r3d_host.h): one grid node as the cell builders see it."""
    _fields_ = [("loc", C.c_double * 3), ("radius", C.c_double), ("side", (C.c_double * 9) * 2),
                ("n_sets", C.c_int32), ("pad_", C.c_int32)]


class GridNodeRaw(C.Structure):
    """r3dh_grid_node_raw (include/r3d_host.h): one grid node as the model definition wrote it."""
    _fields_ = [("x", C.c_double * 3), ("set", (C.c_double * 11) * 2), ("n_sets", C.c_int32), ("pad_", C.c_int32)]


class Result(C.Structure):
    _fields_ = [("energy", _dp), ("counts", C.POINTER(C.c_uint64)), ("n_lost", C.c_uint64),
                ("n_timeout", C.c_uint64), ("n_invalid", C.c_uint64),
                ("invalid_reasons", C.c_uint64 * R3D_INV_NUM), ("events", C.c_uint64 * R3D_EV_NUM)]


class VolumeDesc(C.Structure):
    _fields_ = [("origin", C.c_double * 3), ("cell_size", C.c_double * 3), ("dims", C.c_uint32 * 3),
                ("n_frames", C.c_uint32), ("frame_dt", C.c_double)]


class Final(C.Structure):
    _fields_ = [("time", C.c_double), ("path", C.c_double), ("amp", C.c_double),
                ("loc", C.c_double * 3), ("dir", C.c_double * 3), ("moves", C.c_uint32),
                ("fate", C.c_uint8), ("type", C.c_uint8), ("n_catch", C.c_uint16)]


class Event(C.Structure):
    """r3d_event (include/r3d.h): one report-stream record."""
    _fields_ = [("id", C.c_uint64), ("time", C.c_double), ("path", C.c_double), ("amp", C.c_double),
                ("loc", C.c_double * 3), ("dir", C.c_double * 3), ("cell", C.c_uint32),
                ("moves", C.c_uint32), ("tag", C.c_uint8), ("type", C.c_uint8), ("pad_", C.c_uint8 * 6)]


R3D_RPT_TAGS = ("GEN", "SCT", "REF", "COL", "CEL", "LST", "TMO", "INV")
R3D_RPT_ALL = 255


def event_dtype():
    """numpy view of an r3d_event array."""
    import numpy as np
    return np.dtype([("id", "<u8"), ("time", "<f8"), ("path", "<f8"), ("amp", "<f8"), ("loc", "<f8", 3),
                     ("dir", "<f8", 3), ("cell", "<u4"), ("moves", "<u4"), ("tag", "u1"), ("type", "u1"),
                     ("pad_", "u1", 6)])


def _load(path):
    if not os.path.exists(path):
        raise RuntimeError(
            f"native library missing: {path} -- build it with `make` (or "
            f"`python -c 'import __graft_entry__ as g; g.build()'`); there is no fallback path")
    return C.CDLL(path)


_host = None
_hip = {}


def host_lib():
    """libr3d_host.so: the C++ model builder."""
    global _host
    if _host is None:
        L = _load(os.path.join(LIBDIR, "libr3d_host.so"))
        L.r3dh_model_from_args.restype = C.c_void_p
        L.r3dh_model_from_args.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
        L.r3dh_model_free.argtypes = [C.c_void_p]
        L.r3dh_model_desc.restype = C.POINTER(ModelDesc)
        L.r3dh_model_desc.argtypes = [C.c_void_p]
        L.r3dh_num_phonons.restype = C.c_uint64
        L.r3dh_num_phonons.argtypes = [C.c_void_p]
        L.r3dh_seed.restype = C.c_uint64
        L.r3dh_seed.argtypes = [C.c_void_p]
        L.r3dh_model_log.restype = C.c_char_p
        L.r3dh_model_log.argtypes = [C.c_void_p]
        L.r3dh_grid_dump.restype = C.c_char_p
        L.r3dh_grid_dump.argtypes = [C.c_void_p]
        L.r3dh_scatterer_info.restype = C.c_int
        L.r3dh_scatterer_info.argtypes = [C.c_void_p, C.c_int, _dp]
        L.r3dh_last_error.restype = C.c_char_p
        L.r3dh_scatterer_dump.restype = C.c_char_p
        L.r3dh_scatterer_dump.argtypes = [C.c_void_p]
        L.r3dh_params_echo.restype = C.c_char_p
        L.r3dh_params_echo.argtypes = [C.c_void_p]
        L.r3dh_model_set_scatterer_stats.restype = C.c_int
        L.r3dh_model_set_scatterer_stats.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
        L.r3dh_model_device_tables.restype = C.c_int
        L.r3dh_model_device_tables.argtypes = [C.c_void_p]
        L.r3dh_report_mask.restype = C.c_uint32
        L.r3dh_report_mask.argtypes = [C.c_char_p]
        L.r3dh_model_report_mask.restype = C.c_uint32
        L.r3dh_model_report_mask.argtypes = [C.c_void_p]
        L.r3dh_write_reports.restype = C.c_char_p
        L.r3dh_write_reports.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_char_p]
        L.r3dh_write_outputs.restype = C.c_char_p
        L.r3dh_write_outputs.argtypes = [C.c_void_p, C.POINTER(Result), C.c_char_p, C.c_char_p, C.c_char_p]
        L.r3dh_model_coordinates.restype = C.c_int
        L.r3dh_model_coordinates.argtypes = [C.c_void_p, C.POINTER(C.c_int), _dp, C.POINTER(C.c_int)]
        L.r3dh_grid_size.restype = C.c_int
        L.r3dh_grid_size.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.r3dh_grid_nodes.restype = C.c_int
        L.r3dh_grid_nodes.argtypes = [C.c_void_p, C.POINTER(GridNode), C.c_size_t]
        L.r3dh_grid_nodes_raw.restype = C.c_int
        L.r3dh_grid_nodes_raw.argtypes = [C.c_void_p, C.POINTER(GridNodeRaw), C.c_size_t]
        L.r3dh_seismometer_axes.restype = C.c_int
        L.r3dh_seismometer_axes.argtypes = [C.c_void_p, C.c_int]
        _host = L
    return _host


R3D_COMM_ID_BYTES = 128


class CommInfo(C.Structure):   # include/r3d.h r3d_comm_info
    _fields_ = [("n_ranks", C.c_int32), ("rank", C.c_int32), ("device", C.c_int32), ("rccl_version", C.c_int32),
                ("device_uuid", C.c_char * 40), ("library", C.c_char * 256)]


class EngineOpts(C.Structure):
    """include/r3d.h r3d_engine_opts"""
    _fields_ = [("size", C.c_uint32), ("residency", C.c_int32), ("pool_slots", C.c_uint32),
                ("accumulator_bits", C.c_int32), ("lds_reserve", C.c_uint32)]


def hip_lib(reproducible=False, path=None):
    """libr3d_hip.so: the HIP engine.  Fails loudly when it was not built.

    reproducible: libr3d_hip_repro.so, the same engine built with -DR3D_REPRODUCIBLE -- no
    wave-voted series choices (csrc/r3d_math.h all_lanes), so a history's result is bit-defined by
    (model, seed, id).  path: another build of the engine (a `make variant` library, tools/); nothing
    here or in the library reads the environment."""
    key = os.path.abspath(path) if path else bool(reproducible)
    if key not in _hip:
        # One HIP runtime per process: torch ships its own libamdhip64, and whichever copy is loaded first
        # serves everyone (same SONAME).  The harness allocates result blocks and grids as torch tensors, so
        # torch's copy has to be that one -- loaded the other way round, torch finds "no HIP GPUs".
        import torch  # noqa: F401
        L = _load(path or os.path.join(LIBDIR, "libr3d_hip_repro.so" if reproducible else "libr3d_hip.so"))
        L.r3d_engine_create.restype = C.c_void_p
        L.r3d_engine_create.argtypes = [C.POINTER(ModelDesc), C.c_int]
        L.r3d_engine_create_ex.restype = C.c_void_p
        L.r3d_engine_create_ex.argtypes = [C.POINTER(ModelDesc), C.c_int, C.POINTER(EngineOpts)]
        L.r3d_engine_destroy.argtypes = [C.c_void_p]
        L.r3d_energy_len.restype = C.c_size_t
        L.r3d_energy_len.argtypes = [C.c_void_p]
        L.r3d_counts_len.restype = C.c_size_t
        L.r3d_counts_len.argtypes = [C.c_void_p]
        L.r3d_run.restype = C.c_int
        L.r3d_run.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(Result)]
        L.r3d_run_traced.restype = C.c_int
        L.r3d_run_traced.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64,
                                     C.POINTER(Result), C.POINTER(Final)]
        L.r3d_run_device.restype = C.c_int
        L.r3d_run_device.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.r3d_engine_set_volume.restype = C.c_int
        L.r3d_engine_set_volume.argtypes = [C.c_void_p, C.POINTER(VolumeDesc)]
        L.r3d_volume_len.restype = C.c_size_t
        L.r3d_volume_len.argtypes = [C.c_void_p]
        L.r3d_volume_read.restype = C.c_int
        L.r3d_volume_read.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_int]
        L.r3d_volume_device_ptr.restype = C.c_void_p
        L.r3d_volume_device_ptr.argtypes = [C.c_void_p]
        L.r3d_run_device_carry.restype = C.c_int
        L.r3d_run_device_carry.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.r3d_run_model.restype = C.c_int
        L.r3d_run_model.argtypes = [C.POINTER(ModelDesc), C.c_uint64, C.c_uint64, C.c_uint64, C.c_int,
                                    C.POINTER(Result)]
        L.r3d_engine_scatterer_stats.restype = C.c_int
        L.r3d_engine_scatterer_stats.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
        L.r3d_engine_download_scatterer.restype = C.c_int
        L.r3d_engine_download_scatterer.argtypes = [C.c_void_p, C.c_int, C.POINTER(_dp), _dp]
        L.r3d_engine_download_source.restype = C.c_int
        L.r3d_engine_download_source.argtypes = [C.c_void_p, C.POINTER(_dp), _dp]
        L.r3d_engine_download_toa.restype = C.c_int
        L.r3d_engine_download_toa.argtypes = [C.c_void_p, _dp]
        L.r3d_engine_set_event_log.restype = C.c_int
        L.r3d_engine_set_event_log.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64]
        L.r3d_event_log_count.restype = C.c_uint64
        L.r3d_event_log_count.argtypes = [C.c_void_p]
        L.r3d_event_log_read.restype = C.c_uint64
        L.r3d_event_log_read.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]
        L.r3d_last_kernel_ms.restype = C.c_double
        L.r3d_last_kernel_ms.argtypes = [C.c_void_p]
        L.r3d_launch_count.restype = C.c_uint64
        L.r3d_launch_count.argtypes = [C.c_void_p]
        L.r3d_selftest_math.restype = C.c_int
        L.r3d_selftest_math.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                        C.POINTER(C.c_double), C.c_uint64]
        L.r3d_kernel_ms.restype = C.c_double
        L.r3d_kernel_ms.argtypes = [C.c_void_p, C.c_uint64]
        L.r3d_engine_close.restype = C.c_int
        L.r3d_engine_close.argtypes = [C.c_void_p]
        L.r3d_engine_carry_pending.restype = C.c_int
        L.r3d_engine_carry_pending.argtypes = [C.c_void_p]
        L.r3d_engine_set_volume_buffer.restype = C.c_int
        L.r3d_engine_set_volume_buffer.argtypes = [C.c_void_p, C.POINTER(VolumeDesc), C.c_void_p]
        L.r3d_engine_set_production_finals.restype = C.c_int
        L.r3d_engine_set_production_finals.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        L.r3d_production_finals_read.restype = C.c_int
        L.r3d_production_finals_read.argtypes = [C.c_void_p, C.POINTER(Final), C.c_uint64, C.c_uint64]
        L.r3d_node_create.restype = C.c_void_p
        L.r3d_node_create.argtypes = [C.POINTER(ModelDesc), C.POINTER(C.c_int), C.c_int]
        L.r3d_node_run.restype = C.c_int
        L.r3d_node_run.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(Result)]
        L.r3d_node_size.restype = C.c_int
        L.r3d_node_size.argtypes = [C.c_void_p]
        L.r3d_node_engine.restype = C.c_void_p
        L.r3d_node_engine.argtypes = [C.c_void_p, C.c_int]
        L.r3d_node_reduction.restype = C.c_char_p
        L.r3d_node_reduction.argtypes = [C.c_void_p]
        L.r3d_node_reduction_note.restype = C.c_char_p
        L.r3d_node_reduction_note.argtypes = [C.c_void_p]
        L.r3d_node_destroy.argtypes = [C.c_void_p]
        L.r3d_comm_unique_id.restype = C.c_int
        L.r3d_comm_unique_id.argtypes = [C.c_char_p]
        L.r3d_comm_create.restype = C.c_void_p
        L.r3d_comm_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]
        L.r3d_comm_reduce.restype = C.c_int
        L.r3d_comm_reduce.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                      C.c_int, C.c_void_p]
        L.r3d_comm_describe.restype = C.c_int
        L.r3d_comm_describe.argtypes = [C.c_void_p, C.POINTER(CommInfo)]
        L.r3d_comm_destroy.argtypes = [C.c_void_p]
        L.r3d_run_model_on.restype = C.c_int
        L.r3d_run_model_on.argtypes = [C.POINTER(ModelDesc), C.c_uint64, C.c_uint64, C.c_uint64,
                                       C.POINTER(C.c_int), C.c_int, C.POINTER(Result)]
        L.r3d_engine_variant.restype = C.c_int
        L.r3d_engine_variant.argtypes = [C.c_void_p]
        L.r3d_engine_accumulators.restype = C.c_uint32
        L.r3d_engine_accumulators.argtypes = [C.c_void_p]
        L.r3d_engine_pool_slots.restype = C.c_uint32
        L.r3d_engine_pool_slots.argtypes = [C.c_void_p]
        L.r3d_volume_compact.restype = C.c_int
        L.r3d_volume_compact.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64,
                                         C.c_void_p, C.c_void_p]
        L.r3d_volume_scatter_add.restype = C.c_int
        L.r3d_volume_scatter_add.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p,
                                             C.c_void_p]
        L.r3d_volume_read_range.restype = C.c_int
        L.r3d_volume_read_range.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32)]
        L.r3d_volume_reduce_by_frame.restype = C.c_int
        L.r3d_volume_reduce_by_frame.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
        L.r3d_last_error.restype = C.c_char_p
        L.r3d_version.restype = C.c_char_p
        _hip[key] = L
    return _hip[key]
