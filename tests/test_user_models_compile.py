"""The reference's own model-definition files (user.cpp + user_*_inc.cpp) must
work unchanged against this library's grid.hpp (north_star).
Runs only where the reference checkout is present (not on the GPU box).  The
user files are compiled from where they lie into pytest's temp dir; nothing is
copied into the repo."""
import ctypes as C
import glob
import os
import subprocess

import pytest

from radiative3d_amd import _ffi
from tests.configs import CONFIGS

REF = "/root/reference"
HOST = os.path.join(_ffi.REPO, "radiative3d_amd", "host")

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "user.cpp")),
                                reason="reference checkout not present")


@pytest.fixture(scope="module")
def user_lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("userlib") / "libr3d_host_user.so")
    srcs = [f for f in glob.glob(os.path.join(HOST, "*.cpp")) if not f.endswith("main.cpp")]
    # user.cpp says #include "grid.hpp": feed it on stdin so that the quote-include
    # resolves through -I to THIS repo's grid.hpp, and the user_*_inc.cpp files
    # through the second -I to the reference directory.
    cmd = ["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-pthread", "-I", HOST, "-I", REF,
           "-o", out, "-x", "c++", "-", "-x", "none"] + srcs
    with open(os.path.join(REF, "user.cpp")) as f:
        subprocess.check_call(cmd, stdin=f, cwd=str(tmp_path_factory.getbasetemp()))
    L = C.CDLL(out)
    L.r3dh_model_from_args.restype = C.c_void_p
    L.r3dh_model_from_args.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
    L.r3dh_model_desc.restype = C.POINTER(_ffi.ModelDesc)
    L.r3dh_model_desc.argtypes = [C.c_void_p]
    L.r3dh_grid_dump.restype = C.c_char_p
    L.r3dh_grid_dump.argtypes = [C.c_void_p]
    L.r3dh_model_free.argtypes = [C.c_void_p]
    L.r3dh_last_error.restype = C.c_char_p
    return L


def build(lib, args):
    argv = (C.c_char_p * len(args))(*[a.encode() for a in args])
    h = lib.r3dh_model_from_args(len(args), argv)
    assert h, lib.r3dh_last_error().decode()
    return h


MOHO20 = ("0.8,0.06,0.25,0.5,250,0.8,0.05,0.5,0.5,1000,0.8,0.04,1.0,0.5,2000,"
          "0.7,0.03,0.4,0.3,1500")
NSCP25 = ("0.8,0.01,0.20,0.2,200,0.8,0.01,0.20,0.3,1500,0.8,0.02,0.30,0.3,1200,"
          "0.8,0.01,0.20,0.4,1500,0.8,0.01,0.20,0.5,900")
EXTRA = {
    "lopnor-moho": ["--grid-compiled=1", "--model-args=" + MOHO20, "--toa-degree=2", "--range=1200", "--flatten"],
    "lopnor-moho-alt2": ["--grid-compiled=21", "--model-args=" + MOHO20, "--toa-degree=2", "--range=1200"],
    "lopnor-moho-alt2-defaults": ["--grid-compiled=21", "--toa-degree=2", "--range=1200"],
    "lopnor-moho-alt2-15": ["--grid-compiled=21", "--toa-degree=2", "--range=1200",
                            "--model-args=" + ",".join(MOHO20.split(",")[:15])],
    "toysphere": ["--grid-compiled=30", "--toa-degree=2", "--source-loc=0,0,-500"],
    "upthrust-defaults": ["--grid-compiled=8", "--toa-degree=2"],
    "upthrust-32": ["--grid-compiled=8", "--toa-degree=2", "--model-args=" + NSCP25 + ",2.5,28,9,-12,-3,1.5,4"],
    "upthrust-reversed": ["--grid-compiled=8", "--toa-degree=2", "--model-args=" + NSCP25 + ",2,30,10,-2,-9,0.7,3"],
    "upthrust-level": ["--grid-compiled=8", "--toa-degree=2", "--model-args=" + NSCP25 + ",2,30,10,-5,-5,1,2"],
    "upthrust-28": ["--grid-compiled=8", "--toa-degree=2", "--model-args=" + NSCP25 + ",3,25,8"],
    "scat-params-study": ["--grid-compiled=128", "--toa-degree=2", "--range=300", "--source-loc=0,0,-5"],
}


@pytest.mark.parametrize("name", ["halfspace", "crustpinch", "lopnor", "sphere"] + sorted(EXTRA))
def test_builtin_models_equal_the_users_definitions(user_lib, name):
    """Same do-script arguments through the reference's user_*_inc.cpp and through
    the built-in table-driven definitions: identical grids and identical cell tables."""
    args = EXTRA[name] if name in EXTRA else CONFIGS[name](2)
    builtin = _ffi.host_lib()
    hu, hb = build(user_lib, args), build(builtin, args)
    assert user_lib.r3dh_grid_dump(hu) == builtin.r3dh_grid_dump(hb)
    du, db = user_lib.r3dh_model_desc(hu).contents, builtin.r3dh_model_desc(hb).contents
    assert (du.n_cells, du.n_scatterers, du.cell_kind) == (db.n_cells, db.n_scatterers, db.cell_kind)
    size = C.sizeof(_ffi.Cell) * du.n_cells
    assert C.string_at(du.cells, size) == C.string_at(db.cells, size)
    user_lib.r3dh_model_free(hu), builtin.r3dh_model_free(hb)


def test_every_selector_of_the_users_dispatcher_is_built_in(user_lib):
    """user.cpp:69-125: 1-4, 5-7, 8, 16, 21, 30, 40, 128 (32764 is a deliberate trap)."""
    builtin = _ffi.host_lib()
    for sel in (1, 2, 3, 4, 5, 6, 7, 8, 16, 21, 30, 40, 128):
        args = [f"--grid-compiled={sel}", "--toa-degree=1", "--range=600", "--source-loc=0,0,-5"]
        hu, hb = build(user_lib, args), build(builtin, args)
        assert user_lib.r3dh_grid_dump(hu) == builtin.r3dh_grid_dump(hb), sel
        user_lib.r3dh_model_free(hu), builtin.r3dh_model_free(hb)
