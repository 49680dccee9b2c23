"""Registers, spills, scratch and LDS of every kernel in libr3d_hip.so, from the code object's
metadata notes:  python tools/kernel_resources.py [lib.so] [name filter]"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
DEFAULT_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                           "radiative3d_amd", "lib", "libr3d_hip.so")
KEYS = ("name", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
        "private_segment_fixed_size", "group_segment_fixed_size", "agpr_count")


def kernel_rows(lib=DEFAULT_LIB):
    """One dict per kernel of the gfx950 code object in `lib`: the KEYS above plus `demangled`."""
    tmp = tempfile.mkdtemp()
    # the fat binaries sit in .hip_fatbin, one offload bundle per translation unit: pull the section
    # out, cut it at the bundle magic, unbundle each
    subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, tmp + "/fat.bin"])
    blob = open(tmp + "/fat.bin", "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    notes = ""
    for n, at in enumerate(starts):
        end = starts[n + 1] if n + 1 < len(starts) else len(blob)
        open(f"{tmp}/fat{n}.bin", "wb").write(blob[at:end])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--type=o", "--unbundle", f"--input={tmp}/fat{n}.bin",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={tmp}/dev{n}.co"])
        notes += subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", f"{tmp}/dev{n}.co"], text=True)
    cur, rows = {}, []
    for line in notes.split("\n"):
        m = re.match(r"\s+- \.(agpr_count|args):", line)
        if m and m.group(1) == "agpr_count" and cur:
            rows.append(cur)
            cur = {}
        m = re.match(r"\s+-?\s*\.(\w+):\s+(.*)$", line)
        if m and m.group(1) in KEYS:
            if m.group(1) == "name" and "name" in cur and not m.group(2).startswith("_Z") and "kernel" not in m.group(2):
                continue
            cur[m.group(1)] = m.group(2).strip("'")
    if cur:
        rows.append(cur)
    seen, out = set(), []
    for r in rows:
        n = r.get("name", "?")
        if n in seen:
            continue
        seen.add(n)
        r["demangled"] = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip() \
            .replace("r3d::", "").replace("(KArgs)", "")
        out.append(r)
    return out


def traversal_variants(lib=DEFAULT_LIB):
    """{(cell kind, table residency, role)} of the traversal kernels compiled into `lib`; role is
    "trace" (pool_kernel<..., true>: final records / report stream), "production" (pool_kernel<..., false>:
    a step launch of a carry chain), "job" (pool_job_kernel: a self-contained production launch) or
    "drain" (pool_drain_kernel: a chain's flush).  Residency: 0 cells + scatterer heads in LDS,
    1 heads only, 2 neither."""
    res_of = {("true", "true"): 0, ("false", "true"): 1, ("false", "false"): 2}
    found = set()
    for r in kernel_rows(lib):
        m = re.match(r"void pool_kernel<(\d), (true|false), (true|false), (true|false)>", r["demangled"])
        if m:
            found.add((int(m.group(1)), res_of[(m.group(2), m.group(3))], "trace" if m.group(4) == "true" else "production"))
            continue
        m = re.match(r"void pool_(drain|job)_kernel<(\d), (true|false), (true|false)>", r["demangled"])
        if m:
            found.add((int(m.group(2)), res_of[(m.group(3), m.group(4))], m.group(1)))
    return found


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else DEFAULT_LIB
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for r in kernel_rows(lib):
        if flt not in r.get("name", "?"):
            continue
        print(f"{r['demangled']:58s} vgpr {r.get('vgpr_count','?'):>4} sgpr {r.get('sgpr_count','?'):>4} "
              f"vspill {r.get('vgpr_spill_count','?'):>4} sspill {r.get('sgpr_spill_count','?'):>4} "
              f"scratch {r.get('private_segment_fixed_size','?'):>5} lds {r.get('group_segment_fixed_size','?'):>6}")


if __name__ == "__main__":
    main()
