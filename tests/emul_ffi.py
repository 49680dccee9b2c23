"""ctypes access to the TEST-ONLY host build of the kernel's per-lane code
(tests/emul/libr3d_emul.so).  See tests/emul/emul.cpp."""
import ctypes as C
import os
import subprocess

from radiative3d_amd import _ffi

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(_HERE, "emul", "libr3d_emul.so")
        src = os.path.join(_HERE, "emul", "emul.cpp")
        deps = [src] + [os.path.join(_ffi.REPO, "radiative3d_amd", "csrc", f)
                        for f in os.listdir(os.path.join(_ffi.REPO, "radiative3d_amd", "csrc"))
                        if f.endswith(".h")]
        if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
            subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-o", so, src])
        L = C.CDLL(so)
        L.r3d_emul_run.restype = C.c_int
        L.r3d_emul_run.argtypes = [C.POINTER(_ffi.ModelDesc), C.c_uint64, C.c_uint64, C.c_uint64,
                                   C.POINTER(_ffi.Result), C.POINTER(_ffi.Final)]
        L.r3d_emul_set_volume.argtypes = [C.POINTER(_ffi.VolumeDesc), C.POINTER(C.c_uint32)]
        L.r3d_emul_face_class.restype = C.c_uint32
        L.r3d_emul_face_class.argtypes = [C.POINTER(_ffi.ModelDesc), C.c_int, C.c_int]
        L.r3d_emul_class_from_corners.restype = C.c_uint32
        L.r3d_emul_class_from_corners.argtypes = [C.POINTER(C.c_double), C.c_int]
        L.r3d_emul_sample_cdf.restype = None
        L.r3d_emul_sample_cdf.argtypes = [C.POINTER(C.c_double), C.c_uint64, C.c_uint32, C.POINTER(C.c_double), C.c_uint64,
                                          C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.r3d_emul_slow_moves.restype = C.c_ulonglong
        L.r3d_emul_slow_moves.argtypes = [C.c_int]
        L.r3d_emul_math.restype = C.c_double
        L.r3d_emul_math.argtypes = [C.c_int, C.c_double, C.c_double]
        _lib = L
    return _lib


def run_with_volume(model, n, vdesc, first_id=0, seed=0x5EED):
    import numpy as np
    shape = (2, int(vdesc.n_frames), int(vdesc.dims[2]), int(vdesc.dims[1]), int(vdesc.dims[0]))
    vol = np.zeros(shape, dtype=np.uint32)
    lib().r3d_emul_set_volume(C.byref(vdesc), vol.ctypes.data_as(C.POINTER(C.c_uint32)))
    try:
        res = run(model, n, first_id, seed)
    finally:
        lib().r3d_emul_set_volume(None, None)
    return res, vol


def run(model, n, first_id=0, seed=0x5EED, result=None, trace=False):
    res = result if result is not None else model.new_result()
    c = res._as_c()
    finals = (_ffi.Final * n)() if trace else None
    if lib().r3d_emul_run(model.desc_p, n, first_id, seed, C.byref(c), finals):
        raise RuntimeError("emul run failed")
    res._from_c(c)
    return (res, finals) if trace else res


F_SMOOTH, F_STEP = 16, 32


def class_from_corners(steps):
    """Pack-time class of a face from the signed velocity steps [(P, S), ...] at its corners."""
    flat = [x for pair in steps for x in pair]
    return int(lib().r3d_emul_class_from_corners((C.c_double * len(flat))(*flat), len(steps)))


def face_class(model, cell, face):
    return int(lib().r3d_emul_face_class(model.desc_p, cell, face))


def math_fn(which, x, y=0.0):
    """One of the kernel's lean elementary functions (tests/emul/emul.cpp r3d_emul_math)."""
    return float(lib().r3d_emul_math(which, float(x), float(y)))


def sample_cdf_both_ways(cdf, u, bits=0):
    """Indices drawn from the cumulative table `cdf` for the uniforms `u`: through the search guide's
    cells and by bisecting the whole table; and the longest bracket of the guide."""
    import numpy as np
    cdf = np.ascontiguousarray(cdf, dtype=np.float64)
    u = np.ascontiguousarray(u, dtype=np.float64)
    a = np.empty(u.size, dtype=np.uint64)
    b = np.empty(u.size, dtype=np.uint64)
    longest = C.c_uint64(0)
    dp, up = C.POINTER(C.c_double), C.POINTER(C.c_uint64)
    lib().r3d_emul_sample_cdf(cdf.ctypes.data_as(dp), cdf.size, bits, u.ctypes.data_as(dp), u.size,
                              a.ctypes.data_as(up), b.ctypes.data_as(up), C.byref(longest))
    return a, b, int(longest.value)
