"""The C-ABI libraries load and export every entry point include/*.h declares;
the ctypes mirrors have the layouts the C compiler gives the structs.  No
compute calls here: these run without a GPU."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

from radiative3d_amd import _ffi

INCLUDE = os.path.join(_ffi.REPO, "include")


def declared_functions(header):
    text = open(os.path.join(INCLUDE, header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"#define[^\n]*", "", text)
    return sorted(set(re.findall(r"\b(r3dh?_[a-z0-9_]+)\s*\(", text)))


def test_engine_library_exports_every_declared_symbol():
    names = declared_functions("r3d.h")
    assert {"r3d_engine_create", "r3d_run", "r3d_run_device", "r3d_run_traced",
            "r3d_engine_destroy", "r3d_last_error"} <= set(names)
    lib = C.CDLL(os.path.join(_ffi.LIBDIR, "libr3d_hip.so"))
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_host_library_exports_every_declared_symbol():
    names = declared_functions("r3d_host.h")
    lib = C.CDLL(os.path.join(_ffi.LIBDIR, "libr3d_host.so"))
    missing = [n for n in names if not hasattr(lib, n)]
    assert names and not missing, missing


def test_ctypes_mirror_matches_c_layout():
    prog = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "r3d.h"
    int main(void) {
      printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(r3d_face), sizeof(r3d_cell),
             sizeof(r3d_scatterer), sizeof(r3d_source), sizeof(r3d_seismometer), sizeof(r3d_params),
             sizeof(r3d_model_desc), sizeof(r3d_result), sizeof(r3d_final));
      printf("%zu %zu %zu %d\n", offsetof(r3d_cell, faces), offsetof(r3d_model_desc, source),
             offsetof(r3d_result, events), R3D_N_SCALARS);
      return 0;
    }'''
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "s.c")
        open(src, "w").write(prog)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", INCLUDE, "-o", exe, src])
        out = subprocess.check_output([exe]).decode().split()
    sizes = [int(x) for x in out]
    want = [C.sizeof(t) for t in (_ffi.Face, _ffi.Cell, _ffi.Scatterer, _ffi.Source, _ffi.Seismometer,
                                  _ffi.Params, _ffi.ModelDesc, _ffi.Result, _ffi.Final)]
    assert sizes[:9] == want
    assert sizes[9:12] == [_ffi.Cell.faces.offset, _ffi.ModelDesc.source.offset, _ffi.Result.events.offset]
    assert sizes[12] == _ffi.R3D_N_SCALARS


def test_engine_fails_loudly_without_a_gpu(models):
    """No silent CPU fallback: on a box without a HIP device the product path
    raises (on the GPU box this test is a no-op)."""
    import torch
    if torch.cuda.is_available() or torch.cuda.device_count() > 0:
        pytest.skip("GPU present")
    from radiative3d_amd import Engine
    with pytest.raises(RuntimeError, match="no HIP device|no CPU path"):
        Engine(models("halfspace", 3))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(_ffi.REPO, "radiative3d_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")):
                text = open(os.path.join(root, f), errors="ignore").read()
                assert "oracle_ffi" not in text and "r3d_oracle" not in text and "libr3d_oracle" not in text, f


def test_headers_are_strict_c99_and_the_c_example_links(tmp_path):
    """include/*.h from plain C (gcc -std=c99 -pedantic -Werror): no C++ leaks into the ABI."""
    import subprocess
    exe = str(tmp_path / "run_model")
    lib = os.path.join(_ffi.REPO, "radiative3d_amd", "lib")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror",
                           "-I", os.path.join(_ffi.REPO, "include"),
                           os.path.join(_ffi.REPO, "examples", "run_model.c"),
                           "-L", lib, "-lr3d_host", "-lr3d_hip", "-L/opt/rocm/lib",
                           "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    # no GPU here: the model builds, the run is refused with the engine's message
    out = subprocess.run([exe, "1", str(tmp_path)] + [a for a in __import__("tests.configs", fromlist=["x"]).halfspace(3)]
                         + ["--num-phonons=1000"], capture_output=True, text=True)
    if out.returncode != 0:
        assert "no HIP device" in out.stderr or "no CPU path" in out.stderr, out.stderr


def test_the_shipped_engine_reads_no_environment_and_holds_no_developer_switch():
    """The library a maintainer links takes its knobs through r3d_engine_create_ex (include/r3d.h
    r3d_engine_opts), never from the environment; the timing-only R3D_ABLATE_* / R3D_PHASE_TIMING blocks
    exist only under -DR3D_DEV_BUILD, which `make all` never says."""
    csrc = os.path.join(_ffi.REPO, "radiative3d_amd", "csrc")
    switches = set()
    for f in sorted(os.listdir(csrc)):
        text = open(os.path.join(csrc, f)).read()
        code = re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", text, flags=re.S))
        assert "getenv" not in code, f
        switches |= set(re.findall(r"#\s*if\w*\s+!?(?:defined\()?(R3D_ABLATE_\w+|R3D_PHASE_TIMING|R3D_STEP_FINALS)", code))
    # every switch used anywhere is named in the one guard of r3d_tables.h
    guard = re.search(r"#if !defined\(R3D_DEV_BUILD\) && \((.*?)\)\n#error", open(os.path.join(csrc, "r3d_tables.h")).read(), re.S)
    assert guard, "the R3D_DEV_BUILD guard is gone from r3d_tables.h"
    assert switches and switches <= set(re.findall(r"defined\((\w+)\)", guard.group(1))), switches
    # `make all` never defines it: only the variant target's VDEFS does
    mk = open(os.path.join(_ffi.REPO, "Makefile")).read()
    assert [l for l in mk.splitlines() if "R3D_DEV_BUILD" in l and not l.startswith("#")] == ["VDEFS = -DR3D_DEV_BUILD $(DEFS)"]
    # and the built libraries carry no R3D_* names but the header's own constants (which error messages quote)
    header = open(os.path.join(INCLUDE, "r3d.h")).read()
    for so in ("libr3d_hip.so", "libr3d_hip_repro.so"):
        blob = open(os.path.join(_ffi.LIBDIR, so), "rb").read()
        names = {n.decode() for n in set(re.findall(rb"R3D_[A-Z_]{3,}", blob))}
        assert not {n for n in names if not re.search(r"\b" + n + r"\b", header)}, (so, names)
    for py in ("_ffi.py", "model.py", "parallel.py", "launch.py"):
        assert "environ.get(\"R3D_" not in open(os.path.join(_ffi.REPO, "radiative3d_amd", py)).read(), py


def test_engine_opts_mirror_matches_c_layout(tmp_path):
    prog = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "r3d.h"
    int main(void) {
      printf("%zu %zu %zu %zu %zu %zu\n", sizeof(r3d_engine_opts), offsetof(r3d_engine_opts, residency),
             offsetof(r3d_engine_opts, pool_slots), offsetof(r3d_engine_opts, accumulator_bits),
             offsetof(r3d_engine_opts, lds_reserve), offsetof(r3d_engine_opts, size));
      return 0;
    }'''
    src = tmp_path / "s.c"
    src.write_text(prog)
    subprocess.check_call(["gcc", "-std=c99", "-I", INCLUDE, "-o", str(tmp_path / "s"), str(src)])
    got = [int(x) for x in subprocess.check_output([str(tmp_path / "s")]).split()]
    E = _ffi.EngineOpts
    assert got == [C.sizeof(E), E.residency.offset, E.pool_slots.offset, E.accumulator_bits.offset,
                   E.lds_reserve.offset, E.size.offset]


def test_engine_opts_are_validated_before_anything_touches_a_device(models):
    """r3d_engine_create_ex refuses a malformed r3d_engine_opts with a message of its own (no GPU needed: the
    check comes before the device is looked for)."""
    lib = _ffi.hip_lib()
    m = models("halfspace", 3)

    def refused(**kw):
        o = _ffi.EngineOpts(C.sizeof(_ffi.EngineOpts), -1, 0, -1, 0)
        for k, v in kw.items():
            setattr(o, k, v)
        assert not lib.r3d_engine_create_ex(m.desc_p, 0, C.byref(o))
        return lib.r3d_last_error().decode()

    assert "size" in refused(size=8)
    assert "residency" in refused(residency=3) and "residency" in refused(residency=-2)
    assert "accumulator_bits" in refused(accumulator_bits=3) and "accumulator_bits" in refused(accumulator_bits=9)
    assert "pool_slots" in refused(pool_slots=4096)
    assert "lds_reserve" in refused(lds_reserve=1 << 20)
