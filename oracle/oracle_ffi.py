"""ctypes access to the CPU oracle (oracle/libr3d_oracle.so).

TEST INFRASTRUCTURE: import this only from tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg.  The product (radiative3d_amd/) never does.
"""
import ctypes as C
import os

from radiative3d_amd import _ffi

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libr3d_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"oracle not built: {path} (run `make oracle`)")
        L = C.CDLL(path)
        L.r3d_oracle_run.restype = C.c_int
        L.r3d_oracle_run.argtypes = [C.POINTER(_ffi.ModelDesc), C.c_uint64, C.c_uint64, C.c_uint64,
                                     C.POINTER(_ffi.Result), C.POINTER(_ffi.Final)]
        L.r3d_oracle_rt_probs.argtypes = [C.c_double] * 7 + [C.c_int, C.POINTER(C.c_double)]
        L.r3d_oracle_philox.argtypes = [C.POINTER(C.c_uint32)] * 3
        L.r3d_oracle_draw.restype = C.c_double
        L.r3d_oracle_draw.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
        L.r3d_oracle_advance.argtypes = [C.POINTER(_ffi.ModelDesc), C.c_int, C.c_int, C.POINTER(C.c_double),
                                         C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double)]
        L.r3d_oracle_boundary.argtypes = [C.POINTER(_ffi.ModelDesc), C.c_int, C.c_int, C.POINTER(C.c_double),
                                          C.c_double, C.c_double, C.POINTER(C.c_double)]
        L.r3d_oracle_set_volume.argtypes = [C.POINTER(_ffi.VolumeDesc), C.POINTER(C.c_uint32)]
        L.r3d_oracle_set_event_log.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32]
        L.r3d_oracle_event_count.restype = C.c_uint64
        _lib = L
    return _lib


def run_with_events(model, n, mask=_ffi.R3D_RPT_ALL, capacity=1 << 20, first_id=0, seed=0x5EED):
    """Oracle run that also records the report stream; -> (result, events[structured], n_reported)."""
    import numpy as np
    ev = np.zeros(capacity, dtype=_ffi.event_dtype())
    lib().r3d_oracle_set_event_log(ev.ctypes.data, capacity, mask)
    try:
        res = run(model, n, first_id, seed)
        total = int(lib().r3d_oracle_event_count())
    finally:
        lib().r3d_oracle_set_event_log(None, 0, 0)
    return res, ev[:min(total, capacity)], total


def run_with_volume(model, n, vdesc, first_id=0, seed=0x5EED):
    """Oracle run that also fills the volumetric scatter-event grid; -> (result, counts[2,F,z,y,x])."""
    import numpy as np
    shape = (2, int(vdesc.n_frames), int(vdesc.dims[2]), int(vdesc.dims[1]), int(vdesc.dims[0]))
    vol = np.zeros(shape, dtype=np.uint32)
    lib().r3d_oracle_set_volume(C.byref(vdesc), vol.ctypes.data_as(C.POINTER(C.c_uint32)))
    try:
        res = run(model, n, first_id, seed)
    finally:
        lib().r3d_oracle_set_volume(None, None)
    return res, vol


def advance(model, cell, rtype, loc, theta, phi, length):
    """-> dict(loc, theta, phi, time, atten) after moving `length` along the ray."""
    out = (C.c_double * 8)()
    lib().r3d_oracle_advance(model.desc_p, cell, rtype, (C.c_double * 3)(*loc), theta, phi, length, out)
    return dict(loc=list(out[0:3]), theta=out[3], phi=out[4], time=out[5], atten=out[6])


def boundary(model, cell, rtype, loc, theta, phi):
    """-> dict(loc, theta, phi, time, length, face) at the cell boundary."""
    out = (C.c_double * 8)()
    lib().r3d_oracle_boundary(model.desc_p, cell, rtype, (C.c_double * 3)(*loc), theta, phi, out)
    return dict(loc=list(out[0:3]), theta=out[3], phi=out[4], time=out[5], length=out[6], face=int(out[7]))


def run(model, n, first_id=0, seed=0x5EED, result=None, trace=False):
    """Oracle counterpart of Engine.run (same contract)."""
    res = result if result is not None else model.new_result()
    c = res._as_c()
    finals = (_ffi.Final * n)() if trace else None
    rc = lib().r3d_oracle_run(model.desc_p, n, first_id, seed, C.byref(c), finals)
    if rc:
        raise RuntimeError("oracle run failed")
    res._from_c(c)
    return (res, finals) if trace else res


def rt_probs(rho1, a1, b1, rho2, a2, b2, sini, intype):
    out = (C.c_double * 6)()
    lib().r3d_oracle_rt_probs(rho1, a1, b1, rho2, a2, b2, sini, intype, out)
    return list(out)


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().r3d_oracle_philox(c, k, o)
    return list(o)
